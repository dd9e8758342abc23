"""times the brute-force kNN producer at the C2 shape and checks it against torch's exact kNN (GPU box)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ["bench.py"]
import bench
import annembed_amd as A
from annembed_amd import _lib as L
n, dim, k = int(os.environ.get("N", 60000)), int(os.environ.get("DIM", 784)), 12
x = bench.synth_points(n, dim, seed=1)
nbr, dist = bench.knn_rows(x, 0, n, k)
xh = x.cpu().numpy()
for rep in range(2):
    t0 = time.perf_counter()
    g = A.KGraph.bruteforce_l2(xh, k)
    L.check(L.load().ae_synchronize())
    dt = time.perf_counter() - t0
    print("bruteforce_l2 n=%d dim=%d k=%d: %.1f ms (%.2f TFLOP/s incl. upload)" % (n, dim, k, dt * 1e3, 2.0 * n * n * dim / dt / 1e12), flush=True)
ip, nb, ds = g.get_neighbours()
nb = nb.reshape(n, k); ds = ds.reshape(n, k)
print("index agreement with torch topk:", float((nb == nbr.cpu().numpy()).mean()), "max |d - d_torch| rel:",
      float(np.max(np.abs(ds - dist.cpu().numpy()) / np.maximum(ds, 1e-6))))
