"""writes the 16/64-component graph of tools/run_comp_bias.py to /tmp/comp1m.npz (A/B against another build of the library)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = ["bench.py"]
import bench
import annembed_amd as A
n, k, d = int(os.environ.get("N", "1000000")), 6, 2
x, bounds = bench.mixture_points_gpu(n, 28, 16 if n <= 200000 else 64, seed=5, mean_sigma=10.0)
indptr, nbr, dist = bench.component_knn_graph(A, x, bounds, k, permute_seed=int(os.environ["PERMUTE"]) if "PERMUTE" in os.environ else None)
y0 = A.set_data_box(np.random.default_rng(2).normal(size=(n, d)).astype(np.float32), 10.0)
np.savez("/tmp/comp1m.npz", indptr=indptr, nbr=nbr, dist=dist, y0=y0)
