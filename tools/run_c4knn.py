"""configs[3] size on a graph with REAL in-degree skew: the exact kNN graph (k = 6) of 11 M Higgs-shaped points, asked_dim 8, hubness-weighted
negatives -- time-sliced mode (what AE_CE_AUTO picks at 660 M samples per batch) against the ordered dataflow asked for by name.
usage: python tools/run_c4knn.py [n]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 11_000_000
sys.argv = ["bench.py"]
import bench  # noqa: E402
import annembed_amd as A  # noqa: E402
from annembed_amd import _lib as L  # noqa: E402

t0 = time.perf_counter()
x = bench.higgs_shaped_points(n)
print("points generated in %.1f s" % (time.perf_counter() - t0), flush=True)
t0 = time.perf_counter()
kg = A.KGraph.bruteforce_l2(x, 6)
print("exact kNN graph in %.1f s" % (time.perf_counter() - t0), flush=True)
del x
hub = kg.hubness()
print("max in-degree %d" % int(hub.max()), flush=True)
d = 8
y0 = A.set_data_box(np.random.default_rng(1).normal(size=(n, d)).astype(np.float32), 10.0)
npar = A.to_proba_edges(kg, 1.0, 1.0)
auto = A.EntropyOptim(kg, npar, A.EmbedderParams(asked_dim=d, hubness_weighting=True), y0, hub_counts=hub)
print("AE_CE_AUTO resolves to mode %d, slice info %s" % (auto.get_ce_mode(), auto.slice_info()), flush=True)
del auto
for tail in [t for t in os.environ.get("TAILS", "").split(",") if t]:  # A/B of the class cut (debug knob AE_SL_TAIL: the tail of classes with at most this share of the mass joins the overflow class)
    os.environ["AE_DEBUG_KNOBS"] = "1"
    os.environ["AE_SL_TAIL"] = tail
    r = bench.time_mode(A, L, kg, npar, y0, d, A.AE_CE_SLICED, 2, 1, hub=hub)
    print("sliced, tail %s: ms/step %.1f ce_after %.0f info %s" % (tail, r["ms_per_step"], r["ce_after"], r["eo"].slice_info()), flush=True)
    del r
    os.environ.pop("AE_SL_TAIL")
for copt in [t for t in os.environ.get("COPTS", "").split(",") if t]:  # A/B of the cost model's price of an optimistic event (ns)
    os.environ["AE_DEBUG_KNOBS"] = "1"
    os.environ["AE_SL_COPT"] = copt
    r = bench.time_mode(A, L, kg, npar, y0, d, A.AE_CE_SLICED, 2, 1, hub=hub)
    print("sliced, optimistic event priced at %s ns: ms/step %.1f ce_after %.0f info %s" % (copt, r["ms_per_step"], r["ce_after"], r["eo"].slice_info()), flush=True)
    del r
    os.environ.pop("AE_SL_COPT")
modes = os.environ.get("MODES", "sliced,ordered,rounds").split(",")
if "embed" in modes:  # configs[3] end to end on ONE GPU: Embedder.embed() in the default mode (diffusion-map initialisation + 40 batches), then the quality estimate
    par = A.EmbedderParams(asked_dim=d, nb_grad_batch=40, scale_rho=0.75, beta=1.0, grad_step=1.0, nb_sampling_by_edge=10, dmap_init=True, hubness_weighting=True)
    emb = A.Embedder(kg, par)
    L.check(L.load().ae_synchronize())
    t0 = time.perf_counter()
    rc = emb.embed()
    t_embed = time.perf_counter() - t0
    y = emb.get_embedded()
    print("embed() rc %d in %.2f s, finite %s, cross entropy before / after %s" % (rc, t_embed, bool(np.isfinite(y).all()), emb.get_cross_entropy()), flush=True)
    t0 = time.perf_counter()
    q = emb.get_quality_estimate_from_edge_length(50)
    print("quality estimate in %.2f s" % (time.perf_counter() - t0), flush=True)
    del emb
for name, mode in (("sliced", A.AE_CE_SLICED), ("ordered", A.AE_CE_ORDERED), ("rounds", A.AE_CE_HOGWILD)):
    if name not in modes:
        continue
    r = bench.time_mode(A, L, kg, npar, y0, d, mode, 2, 1, hub=hub)
    print("%s ms/step %.1f ce_after %.0f" % (name, r["ms_per_step"], r["ce_after"]), flush=True)
    del r
