"""configs[2]-size default mode (AE_CE_ORDERED, 99 M samples per batch) with and without the look-ahead preparation of the next batch
(debug knob AE_DF_AHEAD; CU shares AE_DF_PREP_CUS): python tools/run_c3_ahead.py"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, time, numpy as np
sys.path.insert(0, %r)
sys.argv = ["bench.py"]
import bench, annembed_amd as A
from annembed_amd import _lib as L
n = 1650000
indptr, nbr, dst = bench.lattice_graph(n, 6, seed=7, permute=True)
kg = A.KGraph(indptr, nbr, dst, 6)
npar = A.to_proba_edges(kg, 1.0, 1.0)
y0 = A.set_data_box(np.random.default_rng(1).normal(size=(n, 2)).astype(np.float32), 10.0)
eo = A.EntropyOptim(kg, npar, A.EmbedderParams(nb_grad_batch=25, ce_mode=A.AE_CE_ORDERED), y0)
S = 10 * eo.get_nb_edges()
ts = []
for it in range(1, 10):
    L.check(L.load().ae_synchronize())
    t0 = time.perf_counter()
    eo.gradient_iteration_threaded(S, 1.0 - it / 25, it)
    L.check(L.load().ae_synchronize())
    ts.append((time.perf_counter() - t0) * 1e3)
df = eo.dataflow_time()
print("RESULT per batch: first %%.1f, median %%.1f, min %%.1f ms; dataflow kernel %%.1f ms; CE %%.4e" %% (ts[0], float(np.median(ts[2:])), min(ts), df[0], eo.ce_compute_threaded()))
''' % ROOT
variants = [("default", {})] + [("look-ahead, %s CUs prepare" % c, {"AE_DEBUG_KNOBS": "1", "AE_DF_AHEAD": "1", "AE_DF_PREP_CUS": c}) for c in (sys.argv[1:] or ["64", "128"])]
for name, env in variants:
    r = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT")]
    print(name, ":", line[0] if line else "FAILED " + r.stderr[-600:], flush=True)
