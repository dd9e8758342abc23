"""A/B of the persistent run of steps (sl_persist_kernel) against one launch per step: a rank's share of a configs[3] batch (rank 0 of N,
no communicator: tools/run_shard_time.py's arrangement) and a mid-size graph on one device.  usage: python tools/run_persist_ab.py [n] [world]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CHILD = r'''
import sys, time, numpy as np
sys.path.insert(0, %r)
sys.argv, argv = ["bench.py"], sys.argv
import bench, annembed_amd as A
from annembed_amd import _lib as L
n, world, d = int(argv[1]), int(argv[2]), 8
indptr, nbr, dst = bench.lattice_graph(n, 6, seed=7, permute=False)   # (ring order: contiguous ranges cut next to nothing)
kg = A.KGraph(indptr, nbr, dst, 6)
npar = A.to_proba_edges(kg, 1.0, 1.0)
y0 = A.set_data_box(np.random.default_rng(1).normal(size=(n, d)).astype(np.float32), 10.0)
lo, hi = 0, n // world
import os
par = A.EmbedderParams(asked_dim=d, nb_grad_batch=25, ce_mode=A.AE_CE_SLICED, grad_step=1.0, ce_precision=int(os.environ.get("AE_AB_PRECISION", "0")))
eo = A.EntropyOptim(kg, npar, par, y0, node_lo=lo, node_hi=hi)
S = 10 * eo.get_nb_edges()
for it in (1, 2):
    eo.gradient_iteration_threaded(S, 1.0 - it / 25, it)
L.check(L.load().ae_synchronize()); eo.kernel_time()
t0 = time.perf_counter()
for it in (3, 4, 5, 6):
    eo.gradient_iteration_threaded(S, 1.0 - it / 25, it)
L.check(L.load().ae_synchronize())
ms = (time.perf_counter() - t0) / 4 * 1e3
cl, ov, _, slices = eo.slice_info()
print("RESULT n %%d world %%d: %%.1f ms per batch and rank, %%d classes, overflow %%.4f, %%d slices, %%.0f events per step, ce %%.4e" %% (n, world, ms, cl, ov, slices, S / max(1, cl * slices), eo.ce_compute_threaded()))
''' % ROOT

n = sys.argv[1] if len(sys.argv) > 1 else "11000000"
for world in (sys.argv[2:] or ["8", "4", "2", "1"]):
    variants = [("one launch per class", {"AE_DEBUG_KNOBS": "1", "AE_SL_NO_MERGE": "1"}), ("merged slices", {"AE_DEBUG_KNOBS": "1", "AE_SL_MERGE": "1"}), ("default", {})]
    if os.environ.get("AE_AB_DBG"):   # timing experiments (results are wrong): 16 no release fence, 32 no acquire fence, 64 no barrier wait
        variants = [("persistent dbg %s" % v, {"AE_DEBUG_KNOBS": "1", "AE_SL_PERSIST": "1", "AE_SL_DBG": v}) for v in os.environ["AE_AB_DBG"].split(",")]
    if os.environ.get("AE_AB_LAUNCH_DBG"):   # the launch path's debug variants: 1 no arithmetic, 2 no stores, 4 no negatives, 8 no static record
        variants = [("launches dbg %s" % v, {"AE_DEBUG_KNOBS": "1", "AE_SL_NO_PERSIST": "1", "AE_SL_DBG": v}) for v in os.environ["AE_AB_LAUNCH_DBG"].split(",")]
    for name, env in variants:
        r = subprocess.run([sys.executable, "-c", CHILD, n, world], env=dict(os.environ, **env), capture_output=True, text=True, timeout=1200)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT")]
        print(name, line[0] if line else ("FAILED: " + r.stderr[-800:]), flush=True)
        if os.environ.get("AE_CE_PROF"):
            for ln in [x for x in r.stderr.splitlines() if x.startswith("CESLICE")][-2:]:
                print("   ", ln, flush=True)
