"""SVD of the graph laplacian (do_approx_svd: rank 20, 5 iterations) timed at the C2 shape and at a lattice of N nodes, on one GPU.
usage: python tools/run_svd_init.py [N ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import annembed_amd as A  # noqa: E402
from annembed_amd import _lib as L  # noqa: E402

sizes = [int(a) for a in sys.argv[1:]] or [1650000]
sys.argv = ["bench.py"]
import bench  # noqa: E402


def timed(kg, tag):
    lap = A.DiffusionMaps(A.DiffusionParams(2, 5.0, 12)).laplacian_from_kgraph(kg)
    s = lap.do_svd(want_u=False)
    L.check(L.load().ae_synchronize())
    reps = 10
    t0 = time.perf_counter()
    for _ in range(reps):
        s = lap.do_svd(want_u=False)
    L.check(L.load().ae_synchronize())
    print("%s do_svd %.3f ms  sigma[0..4] %s" % (tag, (time.perf_counter() - t0) / reps * 1e3, np.array2string(np.asarray(s.s)[:5], precision=6)), flush=True)


x = bench.synth_points(60000, 784, seed=1)
nbr, dist = bench.knn_rows(x, 0, 60000, 12)
indptr = np.arange(60001, dtype=np.uint64) * np.uint64(12)
timed(A.KGraph(indptr, nbr.cpu().numpy().astype(np.uint32).reshape(-1), dist.cpu().numpy().reshape(-1), 12), "C2 60000 k=12")
del x, nbr, dist
for n in sizes:
    ip, nb, ds = bench.lattice_graph(n, 6, seed=7, permute=True)
    timed(A.KGraph(ip, nb, ds, 6), "lattice %d k=6" % n)
