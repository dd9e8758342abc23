"""Run-to-run spread of the statistical modes on the 3000-node star graph of tests/test_gpu_parity.py::test_hub_and_ragged_rows (final CE and
median edge length against the oracle's sequential run).  usage: python tools/run_hub_spread.py [runs]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import annembed_amd as A
from oracle import oracle
from tests.util import knn_graph
oracle.lib()
rng = np.random.default_rng(4)
n = 3000
x = rng.normal(size=(n, 4)).astype(np.float32)
ip0, nb0, ds0 = knn_graph(x, 7)
rows_n, rows_d, ptr = [], [], [0]
for i in range(n):
    keep = 7 if i % 4 else 4
    nb_i = nb0[i * 7:i * 7 + keep].copy(); d_i = ds0[i * 7:i * 7 + keep].copy()
    if i != 0 and 0 not in nb_i: nb_i[-1] = 0
    rows_n.append(nb_i); rows_d.append(d_i); ptr.append(ptr[-1] + keep)
indptr, nbr, dist = np.array(ptr, np.uint64), np.concatenate(rows_n).astype(np.uint32), np.concatenate(rows_d).astype(np.float32)
g = A.KGraph(indptr, nbr, dist)
rc, p0, s0 = oracle.to_proba_edges(indptr, nbr, dist, 1.0, 1.0)
y0 = oracle.set_data_box(rng.normal(size=(n, 2)).astype(np.float32), 10.0)
yo, oce0, oce1 = oracle.entropy_optimize(indptr, nbr, p0, s0, y0, 5)
src = np.repeat(np.arange(n), np.diff(indptr.astype(np.int64)))
lo = np.linalg.norm(yo[src] - yo[nbr], axis=1)
for name, mode in (("event", A.AE_CE_EVENT), ("auto", A.AE_CE_AUTO), ("sliced", A.AE_CE_SLICED)):
    out = []
    for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
        y, ce0, ce1 = A.entropy_optimize(g, A.NodeParams.from_host(g, p0, s0), A.EmbedderParams(nb_grad_batch=5, ce_mode=mode), y0)
        lg = np.linalg.norm(y[src] - y[nbr], axis=1)
        out.append((ce1 / oce1, np.median(lg) / np.median(lo)))
    print(name, "ce", " ".join("%.3f" % a for a, _ in out), "| median", " ".join("%.3f" % b for _, b in out), flush=True)
