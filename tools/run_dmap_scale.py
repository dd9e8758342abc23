"""The diffusion-map initialisation REPLICATED on one GPU at the node count of a multi-GPU config (configs[4]: 50 M nodes -> 16-D): time and
device memory of the laplacian build, the rank-20 randomized SVD and the whole initial embedding -- what every rank of a sharded run pays
(the initialisation is not sharded).  usage: python tools/run_dmap_scale.py [n] [k] [asked_dim]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 6
d = int(sys.argv[3]) if len(sys.argv) > 3 else 16
sys.argv = ["bench.py"]
import bench  # noqa: E402
import torch  # noqa: E402
import annembed_amd as A  # noqa: E402
from annembed_amd import _lib as L  # noqa: E402


def used_gb():
    free, total = torch.cuda.mem_get_info()
    return (total - free) / 2**30


def sync():
    L.check(L.load().ae_synchronize())


t0 = time.perf_counter()
ip, nb, ds = bench.lattice_graph(n, k, seed=7, permute=True)
print("lattice %d x k %d built on the host in %.1f s" % (n, k, time.perf_counter() - t0), flush=True)
kg = A.KGraph(ip, nb, ds, k)
del ip, nb, ds
sync()
print("graph on the device: %.1f GB in use" % used_gb(), flush=True)
dm = A.DiffusionMaps(A.DiffusionParams(d, 5.0, 12))
t0 = time.perf_counter()
lap = dm.laplacian_from_kgraph(kg)
sync()
print("laplacian: %.2f s, %.1f GB in use" % (time.perf_counter() - t0, used_gb()), flush=True)
t0 = time.perf_counter()
s = lap.do_svd(want_u=False)
sync()
print("do_svd (rank 20, 5 iterations): %.2f s, %.1f GB in use, sigma[0..3] %s" % (time.perf_counter() - t0, used_gb(), np.asarray(s.s)[:4]), flush=True)
del lap, s
t0 = time.perf_counter()
y0 = dm.embed_from_kgraph(kg)
sync()
print("embed_from_kgraph -> %s: %.2f s, %.1f GB in use, finite %s" % (y0.shape, time.perf_counter() - t0, used_gb(), bool(np.isfinite(y0).all())), flush=True)
