"""merged slices (sl_slice_kernel) against one launch per class against the exact mode, several seeds each, on a clustered 1 M-node graph
(class path forced): final CE and edge-length quartiles.  usage: python tools/run_merge_fidelity.py [n] [seeds]"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, numpy as np
sys.path.insert(0, %r)
sys.argv, argv = ["bench.py"], sys.argv
import bench, annembed_amd as A
n, seed, mode = int(argv[1]), int(argv[2]), argv[3]
gr = bench.config_graphs(A, "c4", n_override=n)
kg = A.KGraph(gr["indptr"], gr["nbr"], gr["dist"], 6)
npar = A.to_proba_edges(kg, 1.0, 1.0)
y0 = A.set_data_box(np.random.default_rng(1).normal(size=(n, 8)).astype(np.float32), 10.0)
par = A.EmbedderParams(asked_dim=8, nb_grad_batch=20, grad_step=1.0, seed=seed, ce_mode=A.AE_CE_SEQUENTIAL if mode == "seq" else A.AE_CE_SLICED)
y, _, ce = A.entropy_optimize(kg, npar, par, y0)
src = np.repeat(np.arange(n), 6)
q = np.quantile(np.linalg.norm(y[src] - y[gr["nbr"]], axis=1), [0.05, 0.25, 0.5, 0.75])
print("RESULT", ce, *q)
''' % ROOT
n = sys.argv[1] if len(sys.argv) > 1 else "1000000"
seeds = [int(v) for v in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["1", "2", "3", "4"])]
rows = {}
for name, mode, env in (("sequential", "seq", {}), ("per class", "sl", {"AE_DEBUG_KNOBS": "1", "AE_SL_FORCE_CLASSES": "1", "AE_SL_NO_MERGE": "1"}),
                        ("merged", "sl", {"AE_DEBUG_KNOBS": "1", "AE_SL_FORCE_CLASSES": "1", "AE_SL_MERGE": "1"})):
    out = []
    for sd in seeds:
        r = subprocess.run([sys.executable, "-c", CHILD, n, str(sd), mode], env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT")]
        if not line:
            print(name, "FAILED", r.stderr[-500:])
            continue
        out.append([float(v) for v in line[0].split()[1:]])
    rows[name] = np.array(out)
    print(name, "mean", np.round(rows[name].mean(0), 5), "sd", np.round(rows[name].std(0, ddof=1), 5), flush=True)
for name in ("per class", "merged"):
    a, b = rows[name], rows["sequential"]
    se = np.sqrt(a.var(0, ddof=1) / len(a) + b.var(0, ddof=1) / len(b)) / b.mean(0)
    print(name, "/ sequential: ce, q05, q25, q50, q75 =", np.round(a.mean(0) / b.mean(0), 4), "2 SE", np.round(2 * se, 4))
