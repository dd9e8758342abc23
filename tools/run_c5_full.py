"""configs[4] at FULL size on one GPU, in the faithful default mode: 50 M points of the 128-D mixture (1 000 components of 50 000), k 10 -> 16-D,
exact kNN inside every component, node ids permuted; AE_CE_AUTO (-> the time-sliced mode: 5 G samples per batch in 5 segments), random
start, a warm-up batch and `steps` timed ones.  The north star shards this config over 8 GPUs; one MI355X holds it whole (the driver's
scaling run is the only place where 8 ranks meet).  usage: python tools/run_c5_full.py [n] [steps]   -> one JSON line"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
sys.argv = ["bench.py"]
import bench  # noqa: E402
import annembed_amd as A  # noqa: E402
from annembed_amd import _lib as L  # noqa: E402
import torch  # noqa: E402

t0 = time.perf_counter()
gr = bench.config_graphs(A, "c5", permute_seed=9, n_override=n)
k, d = gr["k"], 16
g = A.KGraph(gr["indptr"], gr["nbr"], gr["dist"], k)
indeg = np.bincount(gr["nbr"], minlength=n)
del gr["dist"]
npar = A.to_proba_edges(g, 1.0, 1.0)
y0 = A.set_data_box(np.random.default_rng(1).normal(size=(n, d)).astype(np.float32), 10.0)
nb = 20
eo = A.EntropyOptim(g, npar, A.EmbedderParams(asked_dim=d, nb_grad_batch=nb, ce_mode=A.AE_CE_AUTO, grad_step=1.0), y0)
S = 10 * eo.get_nb_edges()
ce0 = eo.ce_compute_threaded()
t_setup = time.perf_counter() - t0
times = []
for it in range(1, steps + 2):
    L.check(L.load().ae_synchronize())
    t1 = time.perf_counter()
    eo.gradient_iteration_threaded(S, 1.0 * (1 - it / nb), it)
    L.check(L.load().ae_synchronize())
    times.append(time.perf_counter() - t1)
ce1 = eo.ce_compute_threaded()
classes, ov, rounds, slices = eo.slice_info()
y = eo.get_embedded()
free, total = torch.cuda.mem_get_info()
print(json.dumps({"nodes": n, "k": k, "asked_dim": d, "graph": gr["desc"], "max_in_degree": int(indeg.max()), "in_degree_q999": float(np.quantile(indeg, 0.999)),
                  "samples_per_batch": int(S), "ce_mode": int(eo.get_ce_mode()), "setup_s": round(t_setup, 1), "warmup_batch_s": round(times[0], 3),
                  "batch_s": [round(t, 3) for t in times[1:]], "points_per_s": n / float(np.mean(times[1:])), "samples_per_s": S / float(np.mean(times[1:])),
                  "roofline_frac_by_algorithmic_bytes": 640.0 * S / float(np.mean(times[1:])) / 8e12,
                  "classes": classes, "overflow_mass_fraction": ov, "slices_last_batch": slices, "ce_before": ce0, "ce_after": ce1,
                  "finite": bool(np.isfinite(y).all()), "hbm_used_gb": round((total - free) / 1e9, 1)}))
