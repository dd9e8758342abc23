"""time per batch of AE_CE_ORDERED on the C2 graph (MNIST-shaped 60 k x k 12) + final CE ratio to sequential: python tools/run_ordered_time.py [batches]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 25
sys.argv = ["bench.py"]
import bench  # noqa: E402
import annembed_amd as A  # noqa: E402
from annembed_amd import _lib as L  # noqa: E402

n = 60000
kg = A.KGraph.bruteforce_l2(bench.synth_points(n, 784, seed=1).cpu().numpy(), 12)
npar = A.to_proba_edges(kg, 1.0, 1.0)
y0 = A.set_data_box(A.DiffusionMaps(A.DiffusionParams(2, 5.0, 12)).embed_from_kgraph(kg), 10.0)
out = {}
for name, mode in (("sequential", A.AE_CE_SEQUENTIAL), ("ordered", A.AE_CE_ORDERED)):
    eo = A.EntropyOptim(kg, npar, A.EmbedderParams(nb_grad_batch=nb, ce_mode=mode), y0)
    S = 10 * eo.get_nb_edges()
    ts = []
    for it in range(1, nb + 1):
        L.check(L.load().ae_synchronize())
        t0 = time.perf_counter()
        eo.gradient_iteration_threaded(S, 1.0 - it / nb, it)
        L.check(L.load().ae_synchronize())
        ts.append((time.perf_counter() - t0) * 1e3)
    out[name] = eo.ce_compute_threaded()
    df = eo.dataflow_time()
    print("%-11s per batch: first %.2f, median %.2f, min %.2f ms; dataflow kernel %.2f ms avg; CE %.0f (ratio %.4f)" % (
        name, ts[0], float(np.median(ts)), min(ts), df[0], out[name], out[name] / out["sequential"]), flush=True)
