"""one rank's share of a sharded batch on a ring lattice (rank 0 of `world`, no communicator), time-sliced mode: the body of
tools/run_persist_ab.py as a program of its own (profiling).  usage: python tools/run_rank_share.py <n> <world> [batches]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv, argv = ["bench.py"], sys.argv
import bench  # noqa: E402
import annembed_amd as A  # noqa: E402
from annembed_amd import _lib as L  # noqa: E402

n, world, d = int(argv[1]), int(argv[2]), 8
batches = int(argv[3]) if len(argv) > 3 else 4
indptr, nbr, dst = bench.lattice_graph(n, 6, seed=7, permute=False)
kg = A.KGraph(indptr, nbr, dst, 6)
npar = A.to_proba_edges(kg, 1.0, 1.0)
y0 = A.set_data_box(np.random.default_rng(1).normal(size=(n, d)).astype(np.float32), 10.0)
eo = A.EntropyOptim(kg, npar, A.EmbedderParams(asked_dim=d, nb_grad_batch=25, ce_mode=A.AE_CE_SLICED, grad_step=1.0), y0, node_lo=0, node_hi=n // world)
S = 10 * eo.get_nb_edges()
for it in (1, 2):
    eo.gradient_iteration_threaded(S, 1.0 - it / 25, it)
L.check(L.load().ae_synchronize())
t0 = time.perf_counter()
for it in range(3, 3 + batches):
    eo.gradient_iteration_threaded(S, 1.0 - it / 25, it)
L.check(L.load().ae_synchronize())
print("RESULT %.1f ms per batch and rank" % ((time.perf_counter() - t0) / batches * 1e3))
