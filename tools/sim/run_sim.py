"""Research driver for tools/sim/sched_sim.c: fidelity of GPU-style round schedules against the sequential reference loop
(the oracle), on the host.  usage: python tools/sim/run_sim.py GRAPH [n] -- see main()."""
import ctypes as C
import json
import os
import subprocess
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402


class Sim(C.Structure):
    _fields_ = [("n", C.c_uint64), ("dim", C.c_uint64), ("nnz", C.c_uint64),
                ("indptr", C.c_void_p), ("nbr", C.c_void_p), ("proba", C.c_void_p), ("emb_scale", C.c_void_p),
                ("tptr", C.c_void_p), ("teid", C.c_void_p), ("tsrc", C.c_void_p), ("y", C.c_void_p),
                ("hub_odds", C.c_void_p), ("hub_alias", C.c_void_p), ("seed", C.c_uint64),
                ("per_round", C.c_double), ("push", C.c_int), ("carry", C.c_int), ("alternate", C.c_int),
                ("tile", C.c_int), ("tile_chunks", C.c_int), ("group", C.c_int), ("f32math", C.c_int), ("gs", C.c_int)]


def simlib():
    so = os.path.join(HERE, "libschedsim.so")
    src = os.path.join(HERE, "sched_sim.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-fopenmp", "-shared", "-fPIC", "-o", so, src, "-lm"])
    lib = C.CDLL(so)
    lib.sim_batch.restype = C.c_uint64
    return lib


def p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def blobs(n, dim=28, ncomp=64, seed=2):
    rng = np.random.default_rng(seed)
    means = rng.normal(size=(ncomp, dim)) * 2.0
    scales = 0.5 + rng.random((ncomp, dim))
    lab = rng.integers(0, ncomp, n)
    x = means[lab] + scales[lab] * rng.normal(size=(n, dim))
    x = (x - x.mean(0)) / x.std(0)
    return np.ascontiguousarray(x.astype(np.float32))


def mnist_like(n, dim=784, seed=1):
    rng = np.random.default_rng(seed)
    means = np.zeros((10, dim))
    for c in range(10):
        act = rng.choice(dim, 50, replace=False)
        means[c, act] = rng.uniform(0, 255, 50)
    lab = rng.integers(0, 10, n)
    x = np.clip(means[lab] + rng.normal(size=(n, dim)) * 30.0, 0, 255)
    return np.ascontiguousarray(x.astype(np.float32))


def transpose(indptr, nbr, n):
    src = np.repeat(np.arange(n, dtype=np.uint32), np.diff(indptr).astype(np.int64))
    order = np.lexsort((src, nbr))
    tptr = np.zeros(n + 1, np.uint64)
    np.add.at(tptr, nbr.astype(np.int64) + 1, 1)
    tptr = np.cumsum(tptr).astype(np.uint64)
    return tptr, order.astype(np.uint32), src[order].copy()


def edge_stats(indptr, nbr, y):
    n = len(indptr) - 1
    src = np.repeat(np.arange(n), np.diff(indptr).astype(np.int64))
    d = np.sqrt(((y[src] - y[nbr]) ** 2).sum(1))
    return np.quantile(d, [0.05, 0.25, 0.5, 0.75, 0.95])


def run_reference(g, y0, nb_batch, grad_step, hub, seed, hogwild=None):
    indptr, nbr, proba, scale = g
    y, ce0, ce1 = O.entropy_optimize(indptr, nbr, proba, scale, y0, nb_batch, 10, grad_step, 1.0, seed, 0, hub, hogwild)
    return y, ce0, ce1


def run_sim(lib, g, y0, nb_batch, grad_step, hub, seed, **kw):
    indptr, nbr, proba, scale = g
    n = len(indptr) - 1
    dim = y0.shape[1]
    es = O.embedded_scales(np.ascontiguousarray(scale, np.float32))
    tptr, teid, tsrc = transpose(indptr, nbr, n)
    y = np.ascontiguousarray(y0, np.float32).copy()
    s = Sim()
    s.n, s.dim, s.nnz = n, dim, len(nbr)
    keep = [indptr, nbr, proba, es, tptr, teid, tsrc, y]
    s.indptr, s.nbr, s.proba, s.emb_scale = p(indptr), p(nbr), p(proba), p(es)
    s.tptr, s.teid, s.tsrc, s.y = p(tptr), p(teid), p(tsrc), p(y)
    if hub is not None:
        ho, ha = O.alias_build(O.node_sampler_weights(np.asarray(hub)))
        keep += [ho, ha]
        s.hub_odds, s.hub_alias = p(ho), p(ha)
    s.seed = seed
    s.per_round = kw.get("per_round", 4.0)
    s.push = kw.get("push", 1)
    s.carry = kw.get("carry", 1)
    s.alternate = kw.get("alternate", 0)
    s.tile = kw.get("tile", 0)
    s.tile_chunks = kw.get("tile_chunks", 1)
    s.group = kw.get("group", 256)
    s.f32math = kw.get("f32math", 0)
    s.gs = kw.get("gs", 0)
    cnt = np.zeros(len(nbr), np.uint8)
    G = np.zeros(len(nbr) * dim, np.float32)
    vis = np.zeros(n * dim, np.float32)
    mid = np.zeros(n * dim, np.float32)
    eo = O.EntropyOptim(indptr, nbr, proba, scale, y, seed=seed)
    eo.y = y
    eo.c.y = p(y)
    ce0 = eo.ce()
    nb_sample = 10 * len(nbr)
    drawn = 0
    for it in range(1, nb_batch + 1):
        step = grad_step * (1.0 - it / nb_batch)
        drawn += lib.sim_batch(C.byref(s), C.c_uint64(nb_sample), C.c_double(step), C.c_uint32(it), p(cnt), p(G), p(vis), p(mid))
    return y, ce0, eo.ce(), drawn / (nb_sample * nb_batch)


def colour_classes(lib, indptr, nbr, n, seed=5, order_kind="random"):
    nnz = len(nbr)
    src = np.repeat(np.arange(n, dtype=np.uint32), np.diff(indptr).astype(np.int64)).copy()
    rng = np.random.default_rng(seed)
    order = (rng.permutation(nnz) if order_kind == "random" else np.arange(nnz)).astype(np.uint32)
    words = 64
    mask = np.zeros(n * words, np.uint64)
    colour = np.zeros(nnz, np.uint32)
    lib.sim_colour_edges.restype = C.c_uint32
    ncol = lib.sim_colour_edges(C.c_uint64(n), C.c_uint64(nnz), p(src), p(nbr), p(order), C.c_uint32(words), p(mask), p(colour))
    assert ncol != 0xFFFFFFFF
    cedge = np.argsort(colour, kind="stable").astype(np.uint32)
    cptr = np.zeros(ncol + 1, np.uint64)
    np.add.at(cptr, colour.astype(np.int64) + 1, 1)
    cptr = np.cumsum(cptr).astype(np.uint64)
    return ncol, cptr, cedge, src


def run_colour(lib, g, y0, nb_batch, grad_step, hub, seed, sweeps=15, neg_fresh=0, shuffle=0, classes=None, strat=0):
    indptr, nbr, proba, scale = g
    n = len(indptr) - 1
    dim = y0.shape[1]
    es = O.embedded_scales(np.ascontiguousarray(scale, np.float32))
    y = np.ascontiguousarray(y0, np.float32).copy()
    ncol, cptr, cedge, src = classes
    s = Sim()
    s.n, s.dim, s.nnz = n, dim, len(nbr)
    keep = [indptr, nbr, proba, es, y]
    s.indptr, s.nbr, s.proba, s.emb_scale, s.y = p(indptr), p(nbr), p(proba), p(es), p(y)
    if hub is not None:
        ho, ha = O.alias_build(O.node_sampler_weights(np.asarray(hub)))
        keep += [ho, ha]
        s.hub_odds, s.hub_alias = p(ho), p(ha)
    s.seed = seed
    s.alternate = strat
    vis = np.zeros(n * dim, np.float32)
    eo = O.EntropyOptim(indptr, nbr, proba, scale, y, seed=seed)
    eo.y = y
    eo.c.y = p(y)
    ce0 = eo.ce()
    nb_sample = 10 * len(nbr)
    drawn = 0
    lib.sim_batch_colour.restype = C.c_uint64
    for it in range(1, nb_batch + 1):
        step = grad_step * (1.0 - it / nb_batch)
        drawn += lib.sim_batch_colour(C.byref(s), C.c_uint64(nb_sample), C.c_double(step), C.c_uint32(it), C.c_uint32(sweeps), C.c_uint32(ncol),
                                      p(cptr), p(cedge), p(src), p(vis), C.c_int(neg_fresh), C.c_int(shuffle))
    return y, ce0, eo.ce(), drawn / (nb_sample * nb_batch)


def run_seq_stale(lib, g, y0, nb_batch, grad_step, hub, seed, refresh=15, f32math=0):
    indptr, nbr, proba, scale = g
    n = len(indptr) - 1
    dim = y0.shape[1]
    es = O.embedded_scales(np.ascontiguousarray(scale, np.float32))
    y = np.ascontiguousarray(y0, np.float32).copy()
    s = Sim()
    s.n, s.dim, s.nnz = n, dim, len(nbr)
    keep = [indptr, nbr, proba, es, y]
    s.indptr, s.nbr, s.proba, s.emb_scale, s.y = p(indptr), p(nbr), p(proba), p(es), p(y)
    if hub is not None:
        ho, ha = O.alias_build(O.node_sampler_weights(np.asarray(hub)))
        keep += [ho, ha]
        s.hub_odds, s.hub_alias = p(ho), p(ha)
    s.seed = seed
    s.f32math = f32math
    vis = np.zeros(n * dim, np.float32)
    eo = O.EntropyOptim(indptr, nbr, proba, scale, y, seed=seed)
    eo.y = y
    eo.c.y = p(y)
    ce0 = eo.ce()
    nb_sample = 10 * len(nbr)
    lib.sim_batch_seq_stale_neg.restype = C.c_uint64
    for it in range(1, nb_batch + 1):
        step = grad_step * (1.0 - it / nb_batch)
        lib.sim_batch_seq_stale_neg(C.byref(s), C.c_uint64(nb_sample), C.c_double(step), C.c_uint32(it), C.c_uint32(refresh), p(vis))
    return y, ce0, eo.ce(), 1.0


def make_graph(kind, n):
    if kind == "blobs6":      # (i) k = 6 blobs, no hubness weighting, scale_rho 0.75
        x = blobs(n)
        indptr, nbr, dist = O.knn_bruteforce_l2(x, 6)
        rc, proba, scale = O.to_proba_edges(indptr, nbr, dist, 0.75, 1.0)
        return (indptr, nbr, proba, scale), dist, 6
    if kind == "mnist12":     # C2-like
        x = mnist_like(n)
        indptr, nbr, dist = O.knn_bruteforce_l2(x, 12)
        rc, proba, scale = O.to_proba_edges(indptr, nbr, dist, 1.0, 1.0)
        return (indptr, nbr, proba, scale), dist, 12
    raise SystemExit("unknown graph kind")


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "blobs6"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
    nb_batch = int(os.environ.get("NB", "40" if kind == "blobs6" else "25"))
    init = os.environ.get("INIT", "random")
    hubw = os.environ.get("HUB", "0") == "1"
    g, dist, k = make_graph(kind, n)
    hub = O.hubness(g[0], g[1]) if hubw else None
    if init == "dmap":
        dp = O.DiffusionParams(2, 5.0, 12)
        rc, y0, _ = O.dmap_embed_from_kgraph(g[0], g[1], dist, k, dp)
        y0 = O.set_data_box(y0, 10.0)
    else:
        y0 = O.random_init(n, 2, 1.0, 77)
    lib = simlib()
    t0 = time.time()
    yr, ce0, cer = run_reference(g, y0, nb_batch, 1.0, hub, 4664397)
    qr = edge_stats(g[0], g[1], yr)
    print(f"reference sequential: ce {ce0:.0f} -> {cer:.0f}  q {np.round(qr, 4)}  ({time.time() - t0:.1f}s)", flush=True)
    t0 = time.time()
    yr2, _, cer2 = run_reference(g, y0, nb_batch, 1.0, hub, 99)
    print(f"reference seed 2    : ce {cer2 / cer:.4f}  q/qr {np.round(edge_stats(g[0], g[1], yr2) / qr, 3)}  ({time.time() - t0:.1f}s)", flush=True)
    variants = json.loads(os.environ.get("VARIANTS", "[]")) or [
        dict(push=0, per_round=4), dict(push=0, per_round=8), dict(push=3, per_round=4),
        dict(push=1, per_round=4, carry=1), dict(push=1, per_round=4, carry=0), dict(push=1, per_round=8, carry=1),
        dict(push=2, per_round=4, carry=1), dict(push=1, per_round=2, carry=1), dict(push=1, per_round=1, carry=1),
        dict(push=1, per_round=4, carry=1, alternate=1), dict(push=1, per_round=4, gs=1),
    ]
    classes = None
    for v in variants:
        t0 = time.time()
        if "refresh" in v:
            y, _, ce, frac = run_seq_stale(lib, g, y0, nb_batch, 1.0, hub, v.get("seed", 4664397), refresh=v["refresh"], f32math=v.get("f32math", 0))
        elif "sweeps" in v:
            if classes is None:
                classes = colour_classes(lib, g[0], g[1], n)
                sizes = np.diff(classes[1].astype(np.int64))
                print("colours", classes[0], "class sizes (first 8, last 8)", sizes[:8], sizes[-8:], flush=True)
            y, _, ce, frac = run_colour(lib, g, y0, nb_batch, 1.0, hub, 4664397, sweeps=v["sweeps"], neg_fresh=v.get("neg_fresh", 0),
                                        shuffle=v.get("shuffle", 0), classes=classes, strat=v.get("strat", 0))
        else:
            y, _, ce, frac = run_sim(lib, g, y0, nb_batch, 1.0, hub, 4664397, **v)
        q = edge_stats(g[0], g[1], y)
        print(f"{json.dumps(v):70s} ce {ce / cer:.4f}  q/qr {np.round(q / qr, 3)}  drawn {frac:.4f} ({time.time() - t0:.1f}s)", flush=True)


if __name__ == "__main__":
    main()
