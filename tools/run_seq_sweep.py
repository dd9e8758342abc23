"""Sequential (dataflow) mode on the C2 workload: time per batch and bit-exactness against the chip-wide variant under the debug
knobs of launch_dataflow.  usage: AE_DEBUG_KNOBS=1 [AE_DF_ONE_XCD_MAX=.. AE_DF_LANE_STRIDE=.. AE_DF_GRID=..] python tools/run_seq_sweep.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import annembed_amd as A  # noqa: E402
sys.argv = ["bench.py"]
import bench  # noqa: E402

n, k = 60000, int(os.environ.get("K", "12"))
x = bench.synth_points(n, 784, seed=1)
nb_t, ds_t = bench.knn_rows(x, 0, n, k)
indptr = np.arange(n + 1, dtype=np.uint64) * np.uint64(k)
g = A.KGraph(indptr, nb_t.cpu().numpy().astype(np.uint32).reshape(-1), ds_t.cpu().numpy().reshape(-1), k)
npar = A.to_proba_edges(g, 1.0, 1.0)
y0 = (np.random.default_rng(5).random(size=(n, 2)).astype(np.float32) - 0.5) * 10
nb = 8
eo = A.EntropyOptim(g, npar, A.EmbedderParams(nb_grad_batch=nb, ce_mode=A.AE_CE_SEQUENTIAL), y0)
S = 10 * eo.get_nb_edges()
eo.gradient_iteration_threaded(S, 1.0 - 1 / nb, 1)
t0 = time.perf_counter()
for it in range(2, nb + 1):
    eo.gradient_iteration_threaded(S, 1.0 - it / nb, it)
y = eo.get_embedded()
dt = (time.perf_counter() - t0) / (nb - 1) * 1e3
ms, cnt = eo.dataflow_time()
import hashlib
print("ms/batch %.3f  dataflow kernel %.3f ms  ce %.6f  sha %s" % (dt, ms, eo.ce_compute_threaded(), hashlib.sha1(y.tobytes()).hexdigest()[:12]), flush=True)
