"""Rounds / time-sliced mode at a scale shape WITH hubness-weighted negative sampling (examples/higgs.rs sets it): the alias-table
sampler keeps the gathered negatives (the LDS tile serves the uniform sampler only).  usage: python tools/run_hub_scale.py [n k d]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import annembed_amd as A  # noqa: E402
from annembed_amd import _lib as L  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 11_000_000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 6
d = int(sys.argv[3]) if len(sys.argv) > 3 else 8
indptr, nbr, dst = bench.lattice_graph(n, k, seed=7, permute=True)
kg = A.KGraph(indptr, nbr, dst, k)
y0 = A.set_data_box(np.random.default_rng(1).normal(size=(n, d)).astype(np.float32), 10.0)
npar = A.to_proba_edges(kg, 1.0, 1.0)
hub = kg.hubness()
for name, mode in (("rounds", A.AE_CE_HOGWILD), ("sliced", A.AE_CE_SLICED)):
    for hw in (False, True):
        par = A.EmbedderParams(asked_dim=d, nb_grad_batch=25, ce_mode=mode, hubness_weighting=hw)
        eo = A.EntropyOptim(kg, npar, par, y0, hub_counts=hub if hw else None)
        S = 10 * eo.get_nb_edges()
        eo.gradient_iteration_threaded(S, 0.9, 1)
        L.check(L.load().ae_synchronize())
        t0 = time.perf_counter()
        for it in (2, 3, 4):
            eo.gradient_iteration_threaded(S, 0.9, it)
        L.check(L.load().ae_synchronize())
        print("%-7s hubness %-5s ms/batch %.2f  ce after 4 batches %.0f" % (name, hw, (time.perf_counter() - t0) / 3 * 1e3, eo.ce_compute_threaded()), flush=True)
        del eo
