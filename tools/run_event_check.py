"""Event-ordered CE mode (AE_CE_EVENT) against the sequential mode (AE_CE_SEQUENTIAL, bit-exact vs the oracle) on one GPU:
fidelity (final CE, edge-length quantiles) and time per batch.  usage: python tools/run_event_check.py STAGE [args]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import annembed_amd as A  # noqa: E402


def blobs(n, dim=28, ncomp=64, seed=2):
    rng = np.random.default_rng(seed)
    means = rng.normal(size=(ncomp, dim)) * 2.0
    scales = 0.5 + rng.random((ncomp, dim))
    lab = rng.integers(0, ncomp, n)
    x = means[lab] + scales[lab] * rng.normal(size=(n, dim))
    x = (x - x.mean(0)) / x.std(0)
    return np.ascontiguousarray(x.astype(np.float32))


def edge_q(indptr, nbr, y):
    n = len(indptr) - 1
    src = np.repeat(np.arange(n), np.diff(indptr).astype(np.int64))
    d = np.sqrt(((y[src] - y[nbr]) ** 2).sum(1))
    return np.quantile(d, [0.05, 0.25, 0.5, 0.75, 0.95])


def run(g, np_, y0, mode, nb_batch, seed, hub=None, grad_step=1.0, time_it=False):
    par = A.EmbedderParams(nb_grad_batch=nb_batch, ce_mode=mode, seed=seed, hubness_weighting=hub is not None, grad_step=grad_step)
    eo = A.EntropyOptim(g, np_, par, y0, hub_counts=hub)
    ce0 = eo.ce_compute_threaded()
    S = 10 * eo.get_nb_edges()
    t0 = time.perf_counter()
    for it in range(1, nb_batch + 1):
        eo.gradient_iteration_threaded(S, grad_step * (1 - it / nb_batch), it)
    ms, cnt = eo.kernel_time() if mode != A.AE_CE_SEQUENTIAL else (0.0, 0)
    y = eo.get_embedded()
    dt = time.perf_counter() - t0
    drawn, rounds = eo.samples_drawn()
    return dict(y=y, ce0=ce0, ce=eo.ce_compute_threaded(), wall_ms_per_batch=dt / nb_batch * 1e3, event_ms_per_batch=ms, drawn=drawn / (S * nb_batch), rounds=rounds)


def stage_small():
    from tests.util import synthetic_graph
    indptr, nbr, dist, _, _ = synthetic_graph(n=3000, dim=8, k=8, seed=2, ncomp=6)
    g = A.KGraph(indptr, nbr, dist)
    npar = A.to_proba_edges(g, 1.0, 1.0)
    y0 = np.random.default_rng(0).normal(size=(3000, 2)).astype(np.float32) * 3
    for mode, name in ((A.AE_CE_SLICED, "sliced"), (A.AE_CE_EVENT, "event"), (A.AE_CE_SEQUENTIAL, "sequential")):
        r = run(g, npar, y0, mode, 6, 4664397)
        print(name, "ce0 %.1f ce %.1f drawn %.4f rounds %d finite %s q %s wall %.2f ms/batch" % (
            r["ce0"], r["ce"], r["drawn"], r["rounds"], np.isfinite(r["y"]).all(), np.round(edge_q(indptr, nbr, r["y"]), 4), r["wall_ms_per_batch"]), flush=True)


def stage_fidelity(kind="blobs6", n=20000, nb_batch=40, seeds=3):
    if kind == "blobs6":
        x = blobs(n)
        kg = A.KGraph.bruteforce_l2(x, 6)
        rho = 0.75
    else:
        import bench
        x = bench.synth_points(n, 784, seed=1)
        kg = A.KGraph.bruteforce_l2(x.cpu().numpy(), 12)
        rho = 1.0
    indptr, nbr, dist = kg.get_neighbours()
    npar = A.to_proba_edges(kg, rho, 1.0)
    y0 = (np.random.default_rng(5).random(size=(n, 2)).astype(np.float32) - 0.5)
    hub = kg.hubness() if os.environ.get("FID_HUB") else None
    if os.environ.get("FID_HUB") == "ones":  # the weighted sampler with equal weights: the uniform law through the alias tables
        hub = np.ones(n, np.uint32)
    out = {"kind": kind, "n": n, "nb_batch": nb_batch, "runs": []}
    modes = ((A.AE_CE_SEQUENTIAL, "sequential"), (A.AE_CE_EVENT, "event"), (A.AE_CE_SLICED, "sliced"), (A.AE_CE_HOGWILD, "rounds"))
    if os.environ.get("FID_MODES"):
        modes = tuple(m for m in modes if m[1] in os.environ["FID_MODES"].split(","))
    for mode, name in modes:
        for s in range(seeds):
            r = run(kg, npar, y0, mode, nb_batch, 1000 + s, hub=hub)
            q = edge_q(indptr, nbr, r["y"])
            out["runs"].append(dict(mode=name, seed=s, ce=r["ce"], q=q.tolist(), ms=r["wall_ms_per_batch"], drawn=r["drawn"], rounds=r["rounds"]))
            print(name, s, "ce %.0f q %s  %.2f ms/batch drawn %.4f rounds %d" % (r["ce"], np.round(q, 4), r["wall_ms_per_batch"], r["drawn"], r["rounds"]), flush=True)
    ref = [r for r in out["runs"] if r["mode"] == "sequential"]
    mce = np.mean([r["ce"] for r in ref])
    mq = np.mean([r["q"] for r in ref], axis=0)
    for name in ("sequential", "event", "sliced", "rounds"):
        rs = [r for r in out["runs"] if r["mode"] == name]
        if not rs:
            continue
        print("%-10s mean ce/seq %.4f (spread %.4f)  q/seq %s" % (name, np.mean([r["ce"] for r in rs]) / mce, np.std([r["ce"] for r in rs]) / mce,
                                                               np.round(np.mean([r["q"] for r in rs], axis=0) / mq, 3)), flush=True)
    return out


def stage_c2time(nb_batch=6):
    import bench
    n, k = 60000, 12
    x = bench.synth_points(n, 784, seed=1)
    nb_t, ds_t = bench.knn_rows(x, 0, n, k)
    indptr = np.arange(n + 1, dtype=np.uint64) * np.uint64(k)
    nbr, dist = nb_t.cpu().numpy().astype(np.uint32).reshape(-1), ds_t.cpu().numpy().reshape(-1)
    g = A.KGraph(indptr, nbr, dist, k)
    npar = A.to_proba_edges(g, 1.0, 1.0)
    y0 = (np.random.default_rng(5).random(size=(n, 2)).astype(np.float32) - 0.5) * 10
    for mode, name in ((A.AE_CE_SLICED, "sliced"), (A.AE_CE_EVENT, "event"), (A.AE_CE_HOGWILD, "rounds"), (A.AE_CE_SEQUENTIAL, "sequential")):
        r = run(g, npar, y0, mode, nb_batch, 7)
        print(name, "ce %.0f q %s  wall %.3f ms/batch  kernel %.3f ms/batch drawn %.4f rounds %d" % (
            r["ce"], np.round(edge_q(indptr, nbr, r["y"]), 4), r["wall_ms_per_batch"], r["event_ms_per_batch"], r["drawn"], r["rounds"]), flush=True)


if __name__ == "__main__":
    st = sys.argv[1]
    if st == "small":
        stage_small()
    elif st == "fidelity":
        res = stage_fidelity(sys.argv[2] if len(sys.argv) > 2 else "blobs6", int(sys.argv[3]) if len(sys.argv) > 3 else 20000,
                             int(sys.argv[4]) if len(sys.argv) > 4 else 40, int(sys.argv[5]) if len(sys.argv) > 5 else 3)
        if len(sys.argv) > 6:
            json.dump(res, open(sys.argv[6], "w"), indent=1)
    elif st == "c2time":
        stage_c2time()
