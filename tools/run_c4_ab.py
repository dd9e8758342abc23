"""In-process A/B of the time-sliced mode on configs[3]'s own graph (11 M Higgs-shaped points, GLOBAL exact kNN, k 6 -> 8-D, hubness
weighting, node ids permuted): the graph is built ONCE, every variant = a fresh handle under its debug knobs, one warm-up batch + `steps`
timed ones from the same random start.
usage: python tools/run_c4_ab.py "name:KNOB=V,KNOB=V;name2:..." [steps] [shape c4|c5|c3]     (a variant without knobs: "name:")"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
variants = sys.argv[1] if len(sys.argv) > 1 else "default:"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
shape = sys.argv[3] if len(sys.argv) > 3 else "c4"
sys.argv = ["bench.py"]
import bench  # noqa: E402
import annembed_amd as A  # noqa: E402
from annembed_amd import _lib as L  # noqa: E402

if shape == "c3":
    gr = bench.exact_knn_graph(A, bench.higgs_shaped_points(1_650_000), 6, "Higgs-shaped points")
    n, k, d, hubw = 1_650_000, 6, 2, True
elif shape in ("c5", "c5full"):
    gr = bench.config_graphs(A, "c5_full" if shape == "c5full" else "c5")
    n, k, d, hubw = gr["n"], gr["k"], 16, False
else:
    gr = bench.config_graphs(A, "c4")
    n, k, d, hubw = gr["n"], gr["k"], 8, True
g = A.KGraph(gr["indptr"], gr["nbr"], gr["dist"], k)
hub = g.hubness() if hubw else None
npar = A.to_proba_edges(g, 1.0, 1.0)
y0 = A.set_data_box(np.random.default_rng(1).normal(size=(n, d)).astype(np.float32), 10.0)
for v in variants.split(";"):
    name, _, knobs = v.partition(":")
    env = dict(kv.split("=") for kv in knobs.split(",") if kv)
    if env:
        env["AE_DEBUG_KNOBS"] = "1"
    os.environ.update(env)
    try:
        eo = A.EntropyOptim(g, npar, A.EmbedderParams(asked_dim=d, nb_grad_batch=25, ce_mode=A.AE_CE_SLICED, grad_step=1.0, hubness_weighting=hubw), y0, hub_counts=hub)
        S = 10 * eo.get_nb_edges()
        ts = []
        for it in range(1, steps + 2):
            L.check(L.load().ae_synchronize())
            t0 = time.perf_counter()
            eo.gradient_iteration_threaded(S, 1.0 * (1 - it / 25), it)
            L.check(L.load().ae_synchronize())
            ts.append(time.perf_counter() - t0)
        cl, ov, _, slices = eo.slice_info()
        drawn = eo.samples_drawn()[0]
        print("AB", json.dumps({"variant": name, "events_vs_expected_in_sigma": round((drawn - (steps + 1) * S) / np.sqrt((steps + 1) * S), 2), "knobs": env, "ms_per_batch": [round(t * 1e3, 1) for t in ts[1:]], "mean_ms": round(float(np.mean(ts[1:])) * 1e3, 2),
                                "ce_after": eo.ce_compute_threaded(), "classes": cl, "overflow": ov, "slices": slices}), flush=True)
        del eo
    except A.AnnembedError as e:
        print("AB", json.dumps({"variant": name, "knobs": env, "error": str(e)[:300]}), flush=True)
    finally:
        for kk in env:
            os.environ.pop(kk, None)
