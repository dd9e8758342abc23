"""configs[3]'s graph as HNSW would give it: the GLOBAL exact kNN graph of the 11 M Higgs-shaped points (kgraph.rs:440-579: neighbours are
global), through the grouped producer.  usage: python tools/run_knn_grouped.py [n]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv, argv = ["bench.py"], sys.argv
import bench  # noqa: E402
import annembed_amd as A  # noqa: E402

n = int(argv[1]) if len(argv) > 1 else 11_000_000
t0 = time.perf_counter()
x, bounds = bench.mixture_points_gpu(n, 28, 64, seed=3, mean_sigma=2.0, higgs_like=True)
t1 = time.perf_counter()
g = A.KGraph.bruteforce_l2_grouped(x, 6, bounds.astype(np.uint64))
t2 = time.perf_counter()
fell, pb, pa = g.knn_stats
print("points %.1f s, global exact kNN (k = 6) of %d points in %d groups: %.1f s; pairs inside groups %.3g, pruned phase %.3g = %.2f %% of the other pairs, fallback rows %d"
      % (t1 - t0, n, len(bounds) - 1, t2 - t1, pa, pb, 100.0 * pb / (float(n) * n - pa), fell))
ip, nb, ds = g.get_neighbours()
lab = np.searchsorted(bounds, np.arange(n), side="right") - 1
cross = lab[np.repeat(np.arange(n), 6)] != lab[nb]
print("edges that leave their cluster: %.3f %%" % (100.0 * cross.mean()))
indeg = np.bincount(nb, minlength=n)
print("max in-degree %d, q999 %d" % (indeg.max(), np.quantile(indeg, 0.999)))
t3 = time.perf_counter()
order, ranges, rep = g.partition(8)
print("partition into 8 without coordinates: %.2f s" % (time.perf_counter() - t3), rep)
