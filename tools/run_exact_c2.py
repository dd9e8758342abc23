import sys, time, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import annembed_amd as A
from annembed_amd import _lib as L
from oracle import oracle as O
import torch
sys.argv = ["bench.py"]
import bench
n, k = 60000, 12
x = bench.synth_points(n, 784, seed=1)
nbr, dist = bench.knn_rows(x, 0, n, k)
indptr = np.arange(n + 1, dtype=np.uint64) * np.uint64(k)
nbr = nbr.cpu().numpy().astype(np.uint32).reshape(-1); dist = dist.cpu().numpy().reshape(-1)
g = A.KGraph(indptr, nbr, dist, k)
rc, p0, s0 = O.to_proba_edges(indptr, nbr, dist, 1.0, 1.0)
y0 = O.set_data_box(np.random.default_rng(0).normal(size=(n, 2)).astype(np.float32), 10.0)
eo = A.EntropyOptim(g, A.NodeParams.from_host(g, p0, s0), A.EmbedderParams(ce_mode=1), y0)
oo = O.EntropyOptim(indptr, nbr, p0, s0, y0)
S = 10 * len(nbr)
for frac in (0.05, 1.0):
    ns = int(S * frac)
    t0 = time.perf_counter(); eo.gradient_iteration_threaded(ns, 1.0, 1); L.check(L.load().ae_synchronize()); t1 = time.perf_counter()
    oo.gradient_iteration(ns, 1.0, 1); t2 = time.perf_counter()
    print("samples", ns, "gpu sequential s", t1 - t0, "oracle s", t2 - t1, "bit exact", np.array_equal(eo.get_embedded(), oo.y), flush=True)
