"""two-rank embed() on the uniform-square graph (ids in random order) against the one-device modes, several seeds: where the edge-length
quartiles of the sharded run sit.  usage: python tools/run_part_fidelity.py [n] [nb_batch] [exchanges]"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import annembed_amd as A  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
nb_batch = int(sys.argv[2]) if len(sys.argv) > 2 else 12
k = 6
x = np.random.default_rng(8).random((n, 2)).astype(np.float32)
g = A.KGraph.bruteforce_l2(x, k)
indptr, nbr, dist = g.get_neighbours()
src = np.repeat(np.arange(n), k)
npar = A.to_proba_edges(g, 1.0, 1.0)
y0 = A.set_data_box(A.DiffusionMaps(A.DiffusionParams(2, 5.0, 12)).embed_from_kgraph(g), 10.0)


def quart(y):
    return np.quantile(np.linalg.norm(y[src] - y[nbr], axis=1), [0.25, 0.5, 0.75])


res = {}
for name, mode in (("seq", A.AE_CE_SEQUENTIAL), ("sliced", A.AE_CE_SLICED), ("ordered", A.AE_CE_ORDERED)):
    rows = []
    for seed in (1, 2, 3, 4):
        par = A.EmbedderParams(nb_grad_batch=nb_batch, grad_step=1.0, ce_mode=mode, seed=seed)
        y, _, ce = A.entropy_optimize(g, npar, par, y0)
        rows.append([ce] + list(quart(y)))
    res[name] = np.array(rows)
    print(name, "mean", np.round(res[name].mean(0), 5), "sd", np.round(res[name].std(0), 5))
for name in ("sliced", "ordered"):
    print(name, "/ seq:", np.round(res[name].mean(0) / res["seq"].mean(0), 4))
# the sharded runs: order / ranges from the library, the ranks as processes over shared memory
with tempfile.TemporaryDirectory() as td:
    np.savez(os.path.join(td, "graph.npz"), indptr=indptr, nbr=nbr, dist=dist)
    for exch in (1, 4, 16):
        os.environ["AE_TEST_EXCHANGES"] = str(exch)
        rows = []
        for rep in range(3):
            name = "annembed_pf_%d_%d_%d" % (os.getpid(), rep, exch)
            procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "embedder_shm_worker.py"), td, str(r), "2", name, "faithful_dmap"],
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT) for r in range(2)]
            outs = [p.communicate(timeout=600) for p in procs]
            for p, (so, se) in zip(procs, outs):
                assert p.returncode == 0, se[-2000:]
            y = np.load(os.path.join(td, "y_faithful_dmap_rank0.npy"))
            ce = np.load(os.path.join(td, "ce_faithful_dmap_rank0.npy"))
            rows.append([ce[1]] + list(quart(y)))
        rows = np.array(rows)
        print("two ranks, %d exchanges per batch / seq:" % exch, np.round(rows / res["seq"].mean(0), 4).tolist(), "mean", np.round(rows.mean(0) / res["seq"].mean(0), 4))
