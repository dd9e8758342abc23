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
par = A.EmbedderParams(asked_dim=2, nb_grad_batch=25, nb_sampling_by_edge=10, grad_step=1.0, scale_rho=1.0, beta=1.0)  # examples/mnist_fashion.rs:92-110
for rep in range(3):
    e = A.Embedder(kg, par)
    L.check(L.load().ae_synchronize())
    t0 = time.perf_counter(); rc = e.embed(); dt = time.perf_counter() - t0
    print("embed() rc", rc, "wall ms", dt * 1e3, "ce", e.get_cross_entropy())
t0 = time.perf_counter(); rep_ = e.get_quality_estimate_from_edge_length(50); dt = time.perf_counter() - t0
print("quality estimate wall ms", dt * 1e3)
print(rep_)
