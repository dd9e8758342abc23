import sys, time, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import annembed_amd as A
from annembed_amd import _lib as L
import torch
sys.argv = ["bench.py"]
import bench
x = bench.synth_points(60000, 784, seed=1)
nbr, dist = bench.knn_rows(x, 0, 60000, 12)
indptr = np.arange(60001, dtype=np.uint64) * np.uint64(12)
kg = A.KGraph(indptr, nbr.cpu().numpy().astype(np.uint32).reshape(-1), dist.cpu().numpy().reshape(-1), 12)
lap = A.DiffusionMaps(A.DiffusionParams(2, 5.0, 12)).laplacian_from_kgraph(kg)
lap.do_svd(want_u=False)
L.check(L.load().ae_synchronize())
t0 = time.perf_counter()
for _ in range(20): lap.do_svd(want_u=False)
L.check(L.load().ae_synchronize())
print("do_svd ms", (time.perf_counter() - t0) / 20 * 1e3)
