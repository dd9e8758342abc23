"""Fidelity of the FAITHFUL sharded CE loop (AE_CE_AUTO on node ranges -> the time-sliced mode, one process per shard on this box's GPU over the
shared-memory communicator) against the un-sharded sequential mode, as a function of the shards and of the exchanges per batch.
Graphs: (comp) 64 000 points in 16 well separated components, node ids in component order -- no cross-shard edge; (blobs) 60 000
Higgs-shaped points (64 overlapping components), exact global kNN, node ids in component order -- a few per cent of cross-shard edges.
usage: python tools/run_sliced_shard_fidelity.py [comp|blobs] [worlds, e.g. 2,8] [exchanges, e.g. 1,4,16,64,240] [n]   -> JSON lines
(n: number of points, default 64 000 / 60 000; `comp` at n = 1 000 000 uses 64 components)"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
which = sys.argv[1] if len(sys.argv) > 1 else "comp"
worlds = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "2,8").split(",") if v != "none"]   # ("none": the one-device runs only)
exch = [int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else "1,4,16,64,240").split(",")]
n_arg = int(sys.argv[4]) if len(sys.argv) > 4 else 0
sys.argv = ["bench.py"]
import bench  # noqa: E402
import annembed_amd as A  # noqa: E402


def edge_q(indptr, nbr, y, qs=(0.05, 0.25, 0.5, 0.75, 0.95)):
    src = np.repeat(np.arange(len(indptr) - 1), np.diff(indptr.astype(np.int64)))
    return np.quantile(np.linalg.norm(y[src] - y[nbr], axis=1), qs)


def run_ce(g, npar, y0, nb, mode, seed=4664397):
    par = A.EmbedderParams(asked_dim=y0.shape[1], nb_grad_batch=nb, ce_mode=mode, grad_step=1.0, seed=seed)
    eo = A.EntropyOptim(g, npar, par, y0)
    S = 10 * eo.get_nb_edges()
    for it in range(1, nb + 1):
        eo.gradient_iteration_threaded(S, 1.0 * (1 - it / nb), it)
    return eo.get_embedded(), eo.ce_compute_threaded()


if which == "comp":
    n, k, d, nb, rho = n_arg or 64000, 6, 2, 20, 1.0
    x, bounds = bench.mixture_points_gpu(n, 28, 16 if n <= 200000 else 64, seed=5, mean_sigma=10.0)
    indptr, nbr, dist = bench.component_knn_graph(A, x, bounds, k, permute_seed=None)
    y0 = A.set_data_box(np.random.default_rng(2).normal(size=(n, d)).astype(np.float32), 10.0)
else:
    n, k, d, nb, rho = n_arg or 60000, 6, 2, 40, 0.75
    x, lab = bench.higgs_shaped_points(n, with_labels=True)
    order = np.argsort(lab, kind="stable")
    indptr, nbr, dist = A.KGraph.bruteforce_l2(np.ascontiguousarray(x[order]), k).get_neighbours()
    y0 = (np.random.default_rng(5).random(size=(n, d)).astype(np.float32) - 0.5)
g = A.KGraph(indptr, nbr, dist, k)
npar = A.to_proba_edges(g, rho, 1.0)
yr, cer = run_ce(g, npar, y0, nb, A.AE_CE_SEQUENTIAL)
qr = edge_q(indptr, nbr, yr)
for name, mode, seed in (("sequential, another seed", A.AE_CE_SEQUENTIAL, 12345), ("sliced, one device", A.AE_CE_SLICED, 4664397)):
    y, ce = run_ce(g, npar, y0, nb, mode, seed)
    print(json.dumps({"graph": which, "n": n, "run": name, "ce_ratio": ce / cer, "quantile_ratios": np.round(edge_q(indptr, nbr, y) / qr, 4).tolist()}), flush=True)
del g, npar
src = np.repeat(np.arange(n), k)
for world in worlds:
    cross = float(((src * world // n) != (nbr.astype(np.int64) * world // n)).mean())
    for e in exch:
        with tempfile.TemporaryDirectory() as tmp:
            np.savez(os.path.join(tmp, "graph.npz"), indptr=indptr, nbr=nbr, dist=dist, k=k, y0=y0, scale_rho=rho)
            name = "annembed_fid_%d_%d_%d" % (os.getpid(), world, e)
            procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "sliced_shm_worker.py"), tmp, str(r), str(world), name, str(e), str(nb)],
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT) for r in range(world)]
            outs = [p.communicate(timeout=3000) for p in procs]
            bad = [(p.returncode, se[-800:]) for p, (so, se) in zip(procs, outs) if p.returncode]
            if bad:
                print(json.dumps({"graph": which, "n": n, "shards": world, "exchanges_per_batch": e, "error": bad[0][1]}), flush=True)
                continue
            y = np.load(os.path.join(tmp, "y_rank0.npy"))
            info = np.load(os.path.join(tmp, "info_rank0.npy"))
        print(json.dumps({"graph": which, "n": n, "shards": world, "cross_shard_edge_fraction": cross, "exchanges_per_batch": e, "ce_ratio": float(info[0]) / cer,
                          "quantile_ratios": np.round(edge_q(indptr, nbr, y) / qr, 4).tolist(), "bytes_received_per_rank_and_batch": float(info[3]) / nb}), flush=True)
