"""Time of ONE rank's share of a sharded configs[3] batch, alone on this GPU (no communicator: the other shards' rows stay as they are) -- what a
rank of an N-GPU run computes between two exchanges.  usage: python tools/run_shard_time.py [worlds, e.g. 1,2,4,8]   -> JSON lines"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
worlds = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1,2,4,8").split(",")]
sys.argv = ["bench.py"]
import bench  # noqa: E402
import annembed_amd as A  # noqa: E402
from annembed_amd import _lib as L  # noqa: E402
from annembed_amd.dist import shard_range  # noqa: E402

for world in worlds:
    gr = bench.config_graphs(A, "c4", permute_seed=None, shuffle_within_shards=world)
    n, k, d = gr["n"], gr["k"], 8
    g = A.KGraph(gr["indptr"], gr["nbr"], gr["dist"], k)
    hub = g.hubness()
    npar = A.to_proba_edges(g, 1.0, 1.0)
    y0 = A.set_data_box(np.random.default_rng(1).normal(size=(n, d)).astype(np.float32), 10.0)
    lo, hi = shard_range(n, world, 0)
    eo = A.EntropyOptim(g, npar, A.EmbedderParams(asked_dim=d, nb_grad_batch=25, ce_mode=A.AE_CE_AUTO, grad_step=1.0, hubness_weighting=True), y0,
                        node_lo=lo, node_hi=hi, hub_counts=hub)
    S = 10 * eo.get_nb_edges()
    ts = []
    for it in range(1, 6):
        L.check(L.load().ae_synchronize())
        t0 = time.perf_counter()
        eo.gradient_iteration_threaded(S, 1.0 * (1 - it / 25), it)
        L.check(L.load().ae_synchronize())
        ts.append(time.perf_counter() - t0)
    cl, ov, _, slices = eo.slice_info()
    print(json.dumps({"world": world, "rank": 0, "nodes_owned": hi - lo, "samples_per_batch": int(S), "ms_per_batch": [round(t * 1e3, 1) for t in ts[1:]],
                      "classes": cl, "overflow_mass_fraction": ov, "slices": slices, "events_per_step": S / (slices * max(cl, 1)),
                      "implied_speedup_over_one_gpu_without_exchange": None}), flush=True)
    del eo, g, npar
