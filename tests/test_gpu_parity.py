"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI
(annembed_amd.api is a ctypes veneer over include/annembed_hip.h), against the CPU oracle on the same
seeded inputs, against the committed golden vectors, and -- at BASELINE sizes -- through size-independent
properties.  Bars: bit-exact for index / integer work and for the b = 1 SGD arithmetic in sequential
mode; floating-point stages within the tolerance written next to each assert."""
import os

import numpy as np
import pytest

from tests.util import assert_means_close, gaussian_mixture, knn_graph, synthetic_graph

pytestmark = pytest.mark.gpu

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "golden_v1.npz"))
GOLD2 = np.load(os.path.join(os.path.dirname(__file__), "golden", "golden_v2.npz"))  # second restatement: tests/golden/make_golden_v2.py


@pytest.fixture(scope="module")
def A():
    import annembed_amd as A
    from annembed_amd import _lib
    lib = _lib.load()  # raises if the HIP extension is missing: no silent fallback
    import ctypes
    cnt = ctypes.c_int32()
    _lib.check(lib.ae_device_count(ctypes.byref(cnt)))
    assert cnt.value >= 1, "no GPU visible"
    return A


@pytest.fixture(scope="module")
def graph():  # connected (single blob) 2500-node graph, k = 8
    indptr, nbr, dist, x, _ = synthetic_graph(n=2500, dim=6, k=8, seed=21, ncomp=1)
    return indptr, nbr, dist, x


def _relmax(a, b):
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-30)))


# ------------------------------------------------------------------------------------------------
# a1 KGraph
# ------------------------------------------------------------------------------------------------
def test_kgraph_roundtrip_and_validation(A, graph):
    indptr, nbr, dist, _ = graph
    g = A.KGraph(indptr, nbr, dist)
    assert g.get_nb_nodes() == 2500 and g.get_max_nbng() == 8 and g.get_nb_edges() == len(nbr)
    ip, nb, ds = g.get_neighbours()
    assert np.array_equal(ip, indptr) and np.array_equal(nb, nbr) and np.array_equal(ds, dist)
    bad = dist.copy()
    bad[0], bad[1] = bad[1] + 1, bad[0]  # row 0 no longer sorted (kgraph.rs:508-509 invariant)
    with pytest.raises(A.AnnembedError) as e:
        A.KGraph(indptr, nbr, bad)
    assert e.value.code == 1
    ip2 = indptr.copy()
    ip2[1:] -= 8
    ip2[1] = 0  # node 0 has no neighbour -> kgraph.rs:520-537
    with pytest.raises(A.AnnembedError) as e:
        A.KGraph(ip2, nbr[8:], dist[8:])
    assert e.value.code == 3
    nb2 = nbr.copy()
    nb2[5] = 2500
    with pytest.raises(A.AnnembedError):
        A.KGraph(indptr, nb2, dist)
    nb3 = nbr.copy()
    nb3[int(indptr[7]) + 2] = 7  # node 7 as its own neighbour (assert of kgraph.rs:501)
    with pytest.raises(A.AnnembedError) as e:
        A.KGraph(indptr, nb3, dist)
    assert e.value.code == 1


def test_kgraph_from_ragged_bit_exact(A, oracle):
    rng = np.random.default_rng(3)
    n, nbng = 700, 6
    ids = rng.permutation(5000)[:n].astype(np.uint64)  # non contiguous DataIds
    order = rng.permutation(n)
    rows, ptr = [], [0]
    for p in order:
        m = int(rng.integers(3, 14))  # several layers concatenated, duplicates allowed
        cand = rng.choice(np.delete(np.arange(n), p), size=m, replace=True)
        rows.append(cand)
        ptr.append(ptr[-1] + m)
    nbr_id = ids[np.concatenate(rows)]
    nbr_d = rng.integers(1, 50, size=ptr[-1]).astype(np.float32) * 0.25  # many ties
    rc, (oip, onb, ods, oids) = oracle.kgraph_from_ragged(ids[order], np.array(ptr, np.uint64), nbr_id, nbr_d, nbng)
    assert rc == 0
    g = A.KGraph.from_ragged(ids[order], np.array(ptr, np.uint64), nbr_id, nbr_d, nbng)
    ip, nb, ds = g.get_neighbours()
    assert np.array_equal(g.data_ids, oids)
    assert np.array_equal(ip, oip) and np.array_equal(nb, onb) and np.array_equal(ds, ods)
    # isolated point -> Err (kgraph.rs:520-537)
    ptr2 = np.array(ptr, np.uint64)
    ptr2[1:] -= ptr2[1]
    with pytest.raises(A.AnnembedError) as e:
        A.KGraph.from_ragged(ids[order], ptr2, nbr_id[ptr[1]:], nbr_d[ptr[1]:], nbng)
    assert e.value.code == 3


def test_kgraph_file_roundtrip(A, graph, tmp_path):
    indptr, nbr, dist, _ = graph
    g = A.KGraph(indptr, nbr, dist)
    g.data_ids = np.arange(2500, dtype=np.uint64) * 3
    g.save(tmp_path / "g.kgraph")
    g2 = A.KGraph.load(tmp_path / "g.kgraph")
    ip, nb, ds = g2.get_neighbours()
    assert np.array_equal(ip, indptr) and np.array_equal(nb, nbr) and np.array_equal(ds, dist) and g2.get_max_nbng() == 8
    assert np.array_equal(g2.data_ids, g.data_ids)


def test_hubness_bit_exact(A, graph, oracle):
    indptr, nbr, dist, _ = graph
    assert np.array_equal(A.KGraph(indptr, nbr, dist).hubness(), oracle.hubness(indptr, nbr))


def test_distance_batching_and_bruteforce(A):
    x, _ = gaussian_mixture(1200, 24, 3, seed=5)
    indptr, nbr, dist = knn_graph(x, 7)
    g = A.KGraph(indptr, nbr, np.sort(np.abs(np.random.default_rng(0).normal(size=(1200, 7))).astype(np.float32), axis=1).reshape(-1))
    g.fill_l2_distances(x)  # recompute ||x_i - x_j|| on device and re-sort the rows
    ip, nb, ds = g.get_neighbours()
    assert np.array_equal(nb, nbr)
    assert np.allclose(ds, dist, rtol=2e-6, atol=1e-6)
    gb = A.KGraph.bruteforce_l2(x, 7)
    ip, nb, ds = gb.get_neighbours()
    assert np.allclose(ds, dist, rtol=1e-5, atol=1e-5)
    assert (nb == nbr).mean() > 0.999  # ties may be ordered differently


@pytest.mark.parametrize("n,dim,k", [(1200, 24, 7), (1000, 33, 12), (257, 130, 24), (130, 7, 5), (40, 5, 12), (20, 5, 5), (600, 10, 40), (300, 3, 56), (2000, 2, 50)])
def test_knn_producer_bit_exact(A, oracle, n, dim, k):
    """the matrix-core kNN producer returns exactly the rows of its definition (oracle.knn_bruteforce_l2:
    sequential f32 sums of squares, ties by index) -- index sets AND distances bit for bit; n and dim off the tile sizes,
    dim not a multiple of 4, fewer points than the candidate list."""
    x, _ = gaussian_mixture(n, dim, 3, seed=n + dim)
    x = (x * 50.0 + 120.0).astype(np.float32)  # image-like offsets: large norms, small neighbour distances
    ip, nb, ds = A.KGraph.bruteforce_l2(x, k).get_neighbours()
    oi, on, od = oracle.knn_bruteforce_l2(x, k)
    assert np.array_equal(ip, oi) and np.array_equal(nb, on) and np.array_equal(ds, od)


def test_knn_producer_duplicates_and_ties(A, oracle, monkeypatch):
    """40 copies of the same point (more than the 32 candidates a row keeps) and lattice points with many equal
    distances: the certificate must fail for such rows and the brute-force fallback must produce the defined order;
    the plain kernel (AE_KNN_LEGACY) gives the same graph."""
    rng = np.random.default_rng(3)
    x = rng.integers(0, 4, size=(900, 6)).astype(np.float32)  # lattice: massive ties
    x[100:140] = x[100]
    k = 10
    ip, nb, ds = A.KGraph.bruteforce_l2(x, k).get_neighbours()
    oi, on, od = oracle.knn_bruteforce_l2(x, k)
    assert np.array_equal(nb, on) and np.array_equal(ds, od)
    monkeypatch.setenv("AE_KNN_LEGACY", "1")
    ip2, nb2, ds2 = A.KGraph.bruteforce_l2(x, k).get_neighbours()
    assert np.array_equal(nb2, nb) and np.array_equal(ds2, ds)


@pytest.mark.parametrize("n,dim,k,ncomp,spread", [(3000, 12, 6, 5, 2.0), (40000, 28, 6, 16, 2.0), (9000, 7, 10, 9, 30.0), (5000, 128, 10, 4, 10.0)])
def test_grouped_knn_producer_is_the_global_exact_graph(A, oracle, n, dim, k, ncomp, spread):
    """ae_kgraph_bruteforce_l2_grouped: the exact GLOBAL kNN graph of points sorted into clusters -- the rows of the definition
    (oracle.knn_bruteforce_l2 at the small size, the whole-set producer, itself pinned to the oracle, at the others), index sets AND
    distances bit for bit -- with overlapping clusters (most pairs of groups must be visited), well separated ones (none), ragged group
    sizes, a group barely larger than k."""
    rng = np.random.default_rng(n + dim)
    means = rng.normal(size=(ncomp, dim)) * spread
    scales = 0.5 + rng.random((ncomp, dim))
    lab = np.sort(rng.integers(0, ncomp, n))
    lab[:k + 2] = 0                                   # (a small first group is still larger than k)
    x = (means[lab] + scales[lab] * rng.normal(size=(n, dim))).astype(np.float32)
    bounds = np.concatenate([[0], np.cumsum(np.bincount(lab, minlength=ncomp))]).astype(np.uint64)
    bounds = np.unique(bounds)                        # (an empty cluster is no group)
    g = A.KGraph.bruteforce_l2_grouped(x, k, bounds)
    ip, nb, ds = g.get_neighbours()
    if n <= 3000:
        oi, on, od = oracle.knn_bruteforce_l2(x, k)
    else:
        oi, on, od = A.KGraph.bruteforce_l2(x, k).get_neighbours()
    assert np.array_equal(ip, oi) and np.array_equal(nb, on) and np.array_equal(ds, od)
    fell, pairs_b, pairs_a = g.knn_stats
    print("grouped kNN %d x %d, %d groups: pairs inside groups %.3g, pruned phase %.3g of %.3g, fallback rows %d" % (n, dim, len(bounds) - 1, pairs_a, pairs_b,
                                                                                                                    float(n) * n - pairs_a, fell))
    if spread >= 10.0:
        assert pairs_b < 0.2 * (float(n) * n - pairs_a)   # separated clusters: the bound excludes (nearly) everything
    with pytest.raises(A.AnnembedError):
        A.KGraph.bruteforce_l2_grouped(x, k, np.array([0, 3, n], np.uint64))   # a group of 3 points cannot hold k neighbours


def test_grouped_knn_producer_ties_by_the_callers_ids(A, oracle):
    """lattice points (massive ties) and duplicated points, in three groups: the grouped producer reorders the points inside their groups
    internally (shells by distance to the centroid) -- ties must still fall by the CALLER's ids, as the definition says."""
    rng = np.random.default_rng(5)
    x = rng.integers(0, 4, size=(1500, 6)).astype(np.float32)
    x = x[np.argsort(x[:, 0], kind="stable")]
    x[200:240] = x[200]
    bounds = np.array([0, 500, 1100, 1500], np.uint64)
    k = 10
    ip, nb, ds = A.KGraph.bruteforce_l2_grouped(x, k, bounds).get_neighbours()
    oi, on, od = oracle.knn_bruteforce_l2(x, k)
    assert np.array_equal(nb, on) and np.array_equal(ds, od)


def test_embed_cli_end_to_end(A, tmp_path):
    """The `embed` command line (src/bin/embed.rs) on the library: CSV in (first record dropped, io.rs:170-186), exact
    kNN graph, embed, CSV out in the reference's `{:.5e}` format with one row per kept record."""
    from annembed_amd import embed_cli, io
    x, _ = gaussian_mixture(601, 8, 3, seed=2)
    src = tmp_path / "in.csv"
    io.write_csv_array2(str(src), x)
    out = tmp_path / "out.csv"
    assert embed_cli.main(["--csv", str(src), "--out", str(out), "--batch", "8", "--dim", "2", "hnsw", "--dist", "DistL2", "--nbconn", "16",
                           "--ef", "100", "--knbn", "8"]) == 0
    lines = out.read_text().strip().split("\n")
    assert len(lines) == 600
    import re
    assert all(re.fullmatch(r"-?\d\.\d{5}e-?\d+,-?\d\.\d{5}e-?\d+", ln) for ln in lines)
    y = np.array([[float(v) for v in ln.split(",")] for ln in lines], np.float32)
    assert np.isfinite(y).all() and np.abs(y).max() > 0.1
    with pytest.raises(SystemExit):
        embed_cli.main(["--csv", str(src), "--layer", "1"])


# ------------------------------------------------------------------------------------------------
# a2 to_proba_edges
# ------------------------------------------------------------------------------------------------
def test_to_proba_edges_golden_and_oracle(A, graph, oracle):
    g = A.KGraph(GOLD["g_indptr"], GOLD["g_nbr"], GOLD["g_dist"])
    p, s = A.to_proba_edges(g, 1.0, 1.0).get()
    assert np.array_equal(s, GOLD["scale"])  # +,/ only: bit exact
    assert _relmax(p, GOLD["proba"]) < 2e-6  # expf/powf ulps
    p, s = A.to_proba_edges(g, 0.75, 2.0).get()
    assert np.array_equal(s, GOLD["scale_rho075_beta2"]) and _relmax(p, GOLD["proba_rho075_beta2"]) < 2e-6
    indptr, nbr, dist, _ = graph
    np_ = A.to_proba_edges(A.KGraph(indptr, nbr, dist), 1.0, 1.0)
    p, s = np_.get()
    rc, p0, s0 = oracle.to_proba_edges(indptr, nbr, dist, 1.0, 1.0)
    assert rc == 0 and np.array_equal(s, s0) and _relmax(p, p0) < 2e-6
    assert np.allclose(p.reshape(-1, 8).sum(1), 1.0, atol=1e-5)
    assert _relmax(np_.get_perplexity(), oracle.perplexity(indptr, p0)) < 1e-5


# ------------------------------------------------------------------------------------------------
# a10-a13 EntropyOptim
# ------------------------------------------------------------------------------------------------
def _ce_pair(A, oracle, graph, dim, **kw):
    indptr, nbr, dist, _ = graph
    g = A.KGraph(indptr, nbr, dist)
    rc, p0, s0 = oracle.to_proba_edges(indptr, nbr, dist, 1.0, 1.0)
    y0 = oracle.set_data_box(np.random.default_rng(dim).normal(size=(len(indptr) - 1, dim)).astype(np.float32), 10.0)
    npar = A.NodeParams.from_host(g, p0, s0)
    hub = g.hubness() if kw.get("hubness_weighting") else None
    par = A.EmbedderParams(asked_dim=dim, **kw)
    eo = A.EntropyOptim(g, npar, par, y0, hub_counts=hub)
    oo = oracle.EntropyOptim(indptr, nbr, p0, s0, y0, b=par.b, seed=par.seed, sampler=par.ce_sampler, hub_counts=hub)
    return eo, oo, (g, npar)


def test_set_data_box_bit_exact(A, oracle):
    y = np.random.default_rng(1).normal(size=(4000, 3)).astype(np.float32) * 7 + 2
    assert np.array_equal(A.set_data_box(y, 10.0), oracle.set_data_box(y, 10.0))
    assert np.array_equal(A.set_data_box(GOLD["y_raw"], 10.0), GOLD["y_raw_box"])


def test_embedded_scales_and_ce_value(A, oracle, graph):
    eo, oo, _ = _ce_pair(A, oracle, graph, 2, ce_mode=1)   # (the bit-exact mode: the scales are summed in the reference's order)
    assert np.array_equal(eo.get_embedded_scales(), oo.emb_scale)  # f32 sequential mean: bit exact
    assert abs(eo.ce_compute_threaded() - oo.ce()) < 1e-11 * oo.ce()  # f64, summation order differs
    g = A.KGraph(GOLD["g_indptr"], GOLD["g_nbr"], GOLD["g_dist"])
    e2 = A.EntropyOptim(g, A.NodeParams.from_host(g, GOLD["proba"], GOLD["scale"]), A.EmbedderParams(ce_mode=A.AE_CE_SEQUENTIAL), GOLD["y_box"])
    assert np.array_equal(e2.get_embedded_scales(), GOLD["emb_scale"])   # the bit-exact mode: the reference's sequential f32 mean
    assert abs(e2.ce_compute_threaded() - float(GOLD["ce_value"])) < 1e-11 * float(GOLD["ce_value"])
    # every other mode sums the scales as an f64 tree (no single-lane chain of n additions in front of a mode that is not bit-comparable
    # anyway): the same mean to the accuracy of the reference's own f32 sum
    e3 = A.EntropyOptim(g, A.NodeParams.from_host(g, GOLD["proba"], GOLD["scale"]), A.EmbedderParams(), GOLD["y_box"])
    assert np.allclose(e3.get_embedded_scales(), GOLD["emb_scale"], rtol=2e-5, atol=0)   # (a sequential f32 sum of n terms is itself ~sqrt(n) ulps off)


@pytest.mark.parametrize("sampler,hub", [(0, False), (1, False), (0, True), (1, True)])
def test_sample_plan_bit_exact(A, oracle, graph, sampler, hub):
    """RNG + sampling (index work): nodes of every sample identical to the oracle's"""
    eo, oo, _ = _ce_pair(A, oracle, graph, 2, ce_sampler=sampler, hubness_weighting=hub)
    nodes, w = eo.plan(0, 3000, 4)
    ref = np.array([oo.plan(s, 4)[0] for s in range(3000)])
    refw = np.array([oo.plan(s, 4)[1] for s in range(3000)], np.float32)
    assert np.array_equal(nodes, ref) and np.array_equal(w, refw)


@pytest.mark.parametrize("dim", [2, 3, 4, 5, 8, 16])
def test_sequential_sgd_bit_exact(A, oracle, graph, dim):
    """ce_optim_edge_shannon arithmetic (b = 1: +,-,*,/ only) executed in the sequential order through the
    level schedule: coordinates bit-identical to the oracle after whole batches"""
    eo, oo, _ = _ce_pair(A, oracle, graph, dim, ce_mode=1)
    S = 10 * eo.get_nb_edges()
    for it, step in ((1, 1.6), (2, 0.8)):
        eo.gradient_iteration_threaded(S, step, it)
        oo.gradient_iteration(S, step, it)
        assert np.array_equal(eo.get_embedded(), oo.y), "dim %d batch %d" % (dim, it)
    assert abs(eo.ce_compute_threaded() - oo.ce()) < 1e-11 * oo.ce()


def test_sequential_look_ahead_any_call_pattern(A, oracle, graph):
    """The exact mode prepares batch (S, iter + 1) on a second stream while batch (S, iter) runs (ce.hip, round 3).  Whatever the
    caller does next -- the expected batch, another size, an index that skips, a repeated index, other calls in between, a
    handle destroyed with a prepared set nobody asked for -- the coordinates stay the oracle's, bit for bit."""
    eo, oo, _ = _ce_pair(A, oracle, graph, 2, ce_mode=1)
    S = 10 * eo.get_nb_edges()
    calls = [(S, 1.5, 1), (S, 1.4, 2), (S // 2, 1.3, 3), (S // 2, 1.2, 4), (S, 1.1, 7), (S, 1.0, 7), (S, 0.9, 8), (S + 1000, 0.8, 9)]
    for n_call, (ns, step, it) in enumerate(calls):
        eo.gradient_iteration_threaded(ns, step, it)
        oo.gradient_iteration(ns, step, it)
        if n_call % 3 == 1:  # other entry points between two batches
            assert abs(eo.ce_compute_threaded() - oo.ce()) < 1e-11 * oo.ce()
        assert np.array_equal(eo.get_embedded(), oo.y), "call %d %s" % (n_call, (ns, step, it))
    del eo  # a prepared set for (S + 1000, 10) is in flight or waiting: the destructor must drain it
    eo2, oo2, _ = _ce_pair(A, oracle, graph, 2, ce_mode=1)
    eo2.gradient_iteration_threaded(S, 1.0, 1)
    oo2.gradient_iteration(S, 1.0, 1)
    assert np.array_equal(eo2.get_embedded(), oo2.y)


@pytest.mark.parametrize("sampler,hub", [(1, False), (0, True)])
def test_sequential_sgd_bit_exact_samplers(A, oracle, graph, sampler, hub):
    eo, oo, _ = _ce_pair(A, oracle, graph, 2, ce_mode=1, ce_sampler=sampler, hubness_weighting=hub)
    S = 10 * eo.get_nb_edges()
    eo.gradient_iteration_threaded(S, 1.2, 1)
    oo.gradient_iteration(S, 1.2, 1)
    assert np.array_equal(eo.get_embedded(), oo.y)


def test_sequential_sgd_general_b(A, oracle, graph):
    """b != 1 goes through pow(): device libm vs glibc differ by ulps -> 1e-4 relative (north star tolerance)"""
    eo, oo, _ = _ce_pair(A, oracle, graph, 2, ce_mode=1, b=0.8)
    S = 2 * eo.get_nb_edges()
    eo.gradient_iteration_threaded(S, 1.0, 1)
    oo.gradient_iteration(S, 1.0, 1)
    y, r = eo.get_embedded(), oo.y
    assert np.max(np.abs(y - r)) < 1e-4 * np.max(np.abs(r))


def test_sequential_last_batch_step_zero_is_identity(A, oracle, graph):
    eo, oo, _ = _ce_pair(A, oracle, graph, 2, ce_mode=1)
    before = eo.get_embedded()
    eo.gradient_iteration_threaded(5000, 0.0, 9)  # B3: grad_step = 0 on the last batch (embedder.rs:875)
    assert np.array_equal(eo.get_embedded(), before)


def test_sharded_sequential_matches_oracle(A, oracle, graph):
    """node range [lo,hi): positive edges only from owned sources, streams offset by lo << 24"""
    indptr, nbr, dist, _ = graph
    g = A.KGraph(indptr, nbr, dist)
    rc, p0, s0 = oracle.to_proba_edges(indptr, nbr, dist, 1.0, 1.0)
    y0 = oracle.set_data_box(np.random.default_rng(9).normal(size=(2500, 2)).astype(np.float32), 10.0)
    lo, hi = 1000, 2500
    eo = A.EntropyOptim(g, A.NodeParams.from_host(g, p0, s0), A.EmbedderParams(ce_mode=1), y0, node_lo=lo, node_hi=hi)
    oo = oracle.EntropyOptim(indptr, nbr, p0, s0, y0, node_lo=lo, node_hi=hi)
    assert eo.get_nb_edges() == int(indptr[hi] - indptr[lo])
    S = 10 * eo.get_nb_edges()
    eo.gradient_iteration_threaded(S, 1.0, 2)
    oo.gradient_iteration(S, 1.0, 2, s_begin=lo << 24)
    assert np.array_equal(eo.get_embedded(), oo.y)
    assert abs(eo.ce_compute_threaded() - oo.ce()) < 1e-11 * oo.ce()


def test_sequential_bit_exact_at_full_c2_size(A, oracle):
    """BASELINE configs[1] shape -- 60 000 points, k = 12, nb_sampling_by_edge = 10: two whole CE batches of
    7.2 M samples each in AE_CE_SEQUENTIAL mode against the oracle, bit for bit (and ~7x faster than the oracle's
    sequential loop: 0.19 s vs 1.3 s per batch)."""
    import torch
    import bench
    n, k = 60000, 12
    x = bench.synth_points(n, 784, seed=1)
    nb_t, ds_t = bench.knn_rows(x, 0, n, k)
    del x
    indptr = np.arange(n + 1, dtype=np.uint64) * np.uint64(k)
    nbr, dist = nb_t.cpu().numpy().astype(np.uint32).reshape(-1), ds_t.cpu().numpy().reshape(-1)
    del nb_t, ds_t
    torch.cuda.empty_cache()
    g = A.KGraph(indptr, nbr, dist, k)
    rc, p0, s0 = oracle.to_proba_edges(indptr, nbr, dist, 1.0, 1.0)
    assert rc == 0
    y0 = oracle.set_data_box(np.random.default_rng(0).normal(size=(n, 2)).astype(np.float32), 10.0)
    eo = A.EntropyOptim(g, A.NodeParams.from_host(g, p0, s0), A.EmbedderParams(ce_mode=A.AE_CE_SEQUENTIAL), y0)
    oo = oracle.EntropyOptim(indptr, nbr, p0, s0, y0)
    S = 10 * len(nbr)
    for it in (1, 2):
        eo.gradient_iteration_threaded(S, 1.0 * (1 - it / 25), it)
        oo.gradient_iteration(S, 1.0 * (1 - it / 25), it)
        assert np.array_equal(eo.get_embedded(), oo.y), it
    assert abs(eo.ce_compute_threaded() - oo.ce()) < 1e-11 * oo.ce()


@pytest.mark.parametrize("k,nb_batch", [(6, 30), (12, 25)])
def test_full_schedule_bit_exact_vs_oracle_at_config_size(A, oracle, k, nb_batch):
    """One FULL CE schedule per small config against the oracle's sequential loop, bit for bit: configs[0] (MNIST-digits shape:
    60 000 points, k = 6, 30 batches of 3.6 M samples, examples/mnist_digits.rs:92-109) and configs[1] (MNIST-fashion shape:
    k = 12, 25 batches of 7.2 M samples, examples/mnist_fashion.rs:92-110), from the same initial embedding, through
    ae_entropy_optimize in the parity mode (AE_CE_SEQUENTIAL: the sequential-equivalent dataflow).  ~20 / 35 s of oracle time."""
    import torch
    import bench
    n = 60000
    x = bench.synth_points(n, 784, seed=1)
    nb_t, ds_t = bench.knn_rows(x, 0, n, k)
    del x
    indptr = np.arange(n + 1, dtype=np.uint64) * np.uint64(k)
    nbr, dist = nb_t.cpu().numpy().astype(np.uint32).reshape(-1), ds_t.cpu().numpy().reshape(-1)
    del nb_t, ds_t
    torch.cuda.empty_cache()
    g = A.KGraph(indptr, nbr, dist, k)
    rc, p0, s0 = oracle.to_proba_edges(indptr, nbr, dist, 1.0, 1.0)
    assert rc == 0
    y0 = oracle.set_data_box(np.random.default_rng(0).normal(size=(n, 2)).astype(np.float32), 10.0)
    par = A.EmbedderParams(nb_grad_batch=nb_batch, grad_step=1.0, ce_mode=A.AE_CE_SEQUENTIAL)
    y, ce0, ce1 = A.entropy_optimize(g, A.NodeParams.from_host(g, p0, s0), par, y0)
    yo, oce0, oce1 = oracle.entropy_optimize(indptr, nbr, p0, s0, y0, nb_batch, grad_step=1.0)
    assert np.array_equal(y, yo)
    assert abs(ce0 - oce0) < 1e-11 * oce0 and abs(ce1 - oce1) < 1e-11 * oce1


def _edge_len(indptr, nbr, y):
    src = np.repeat(np.arange(len(indptr) - 1), np.diff(indptr.astype(np.int64)))
    return np.linalg.norm(y[src] - y[nbr], axis=1)


def test_event_mode_statistics_match_oracle(A, oracle):
    """The event-ordered mode (AE_CE_EVENT) is not reproducible sample by sample (neither is the reference's rayon loop); its
    statistics are the sequential loop's, and so are the default's (AE_CE_AUTO -> the ordered dataflow at this size): FOUR seeds a side
    against four seeds of the exact mode, |mean ratio - 1| < 2 SE + 1 % on the final cross entropy and the median edge, + 3 % on the
    other quantiles (tests/util.py: assert_means_close; round 5 compared one run with one run at 3 % / 5 %).  AE_CE_SEQUENTIAL
    reproduces the oracle bit for bit.  Kept as evidence next to them: the rounds mode (AE_CE_HOGWILD, stale partner rows) and the
    literal racy per-sample transcription are NOT inside that envelope."""
    n = 20000
    indptr, nbr, dist, _, _ = synthetic_graph(n=n, dim=8, k=8, seed=2, ncomp=6)
    g = A.KGraph(indptr, nbr, dist)
    rc, p0, s0 = oracle.to_proba_edges(indptr, nbr, dist, 1.0, 1.0)
    y0 = oracle.set_data_box(np.random.default_rng(0).normal(size=(n, 2)).astype(np.float32), 10.0)
    npar = A.NodeParams.from_host(g, p0, s0)
    eo = A.EntropyOptim(g, npar, A.EmbedderParams(nb_grad_batch=6), y0)
    assert eo.get_ce_mode() == A.AE_CE_ORDERED
    yo, oce0, oce1 = oracle.entropy_optimize(indptr, nbr, p0, s0, y0, 6)
    yd, _, ced = A.entropy_optimize(g, npar, A.EmbedderParams(nb_grad_batch=6, ce_mode=A.AE_CE_SEQUENTIAL), y0)
    assert np.array_equal(yd, yo) and abs(ced - oce1) < 1e-11 * oce1  # the parity mode: the oracle's run, bit for bit
    lo = _edge_len(indptr, nbr, yo)
    seeds = (4242, 12121, 20000, 27879)
    names, floors = ["ce", "q25", "q50", "q75", "q95"], [0.01, 0.03, 0.01, 0.03, 0.03]

    def runs(mode):
        out = []
        for sd in seeds:
            y, ce0, ce1 = A.entropy_optimize(g, npar, A.EmbedderParams(nb_grad_batch=6, ce_mode=mode, seed=sd), y0)
            if mode != A.AE_CE_SEQUENTIAL:
                assert abs(ce0 - oce0) < 2e-6 * oce0   # (the embedded scales of a mode that is not the bit-exact one: mean summed as an f64 tree, an f32 ulp off the reference's)
            assert np.isfinite(y).all()
            out.append([ce1] + [float(np.quantile(_edge_len(indptr, nbr, y), q)) for q in (0.25, 0.5, 0.75, 0.95)])
        return out

    exact = runs(A.AE_CE_SEQUENTIAL)
    for mode, what in ((A.AE_CE_EVENT, "event-ordered / exact, 20 k nodes"), (A.AE_CE_AUTO, "default (ordered dataflow) / exact, 20 k nodes")):
        assert_means_close(runs(mode), exact, names, floors, what)
    # rounds mode: a throughput mode outside the envelope (documented) -- only sanity here
    yr, _, cer = A.entropy_optimize(g, npar, A.EmbedderParams(nb_grad_batch=6, ce_mode=A.AE_CE_HOGWILD), y0)
    assert np.isfinite(yr).all() and abs(cer - oce1) < 0.25 * oce1
    # the literal per-sample racy transcription is NOT equivalent on a GPU (most updates are lost): keep the evidence
    yr, _, cer = A.entropy_optimize(g, npar, A.EmbedderParams(nb_grad_batch=6, ce_mode=A.AE_CE_SAMPLE_RACY), y0)
    assert np.quantile(_edge_len(indptr, nbr, yr), 0.5) > 1.4 * np.quantile(lo, 0.5)


@pytest.mark.parametrize("dim,k,hub,b", [(5, 8, False, 1.0), (10, 20, True, 1.0), (20, 28, False, 1.0), (3, 32, False, 1.0), (7, 12, True, 1.0),
                                         (2, 10, False, 0.8), (6, 20, True, 1.3), (8, 12, False, 1.0), (16, 10, True, 1.0), (8, 30, True, 1.0),
                                         (16, 16, False, 0.9), (15, 6, False, 1.0), (1, 6, False, 1.0), (33, 9, False, 1.0), (64, 6, True, 1.0)])
def test_ce_any_dim_and_row_length(A, oracle, dim, k, hub, b):
    """Every asked_dim in [1, 64] and every row length (the reference is generic in the dimension and publishes 15-D runs,
    embedder.rs:604-618).  The sequential mode reproduces the oracle's run whatever the dimension (rows are stored zero-padded to
    2 / 3 / 4 / 8 / 16 / 32 / 64 columns; a zero column adds +0 to every distance and never moves): bit for bit at b = 1, 1e-5
    relative CE with the general exponent (whose pow() differs in the last bits).  The default (AE_CE_AUTO -> the ordered
    dataflow at this size), the time-sliced kernel (hubness-weighted negatives, exponent b != 1, rows of up to 32 neighbours)
    and, up to 16 columns, the event-ordered one land within 5 % (CE) / 8 % (edge-length median and q90) of it on these
    4000-node graphs (the oracle's own seed spread here is ~3 %)."""
    n = 4000
    indptr, nbr, dist, _, _ = synthetic_graph(n=n, dim=8, k=k, seed=11, ncomp=3)
    g = A.KGraph(indptr, nbr, dist)
    rc, p0, s0 = oracle.to_proba_edges(indptr, nbr, dist, 1.0, 1.0)
    y0 = oracle.set_data_box(np.random.default_rng(dim).normal(size=(n, dim)).astype(np.float32), 10.0)
    hubc = g.hubness() if hub else None
    par = A.EmbedderParams(asked_dim=dim, nb_grad_batch=5, hubness_weighting=hub, b=b, ce_mode=A.AE_CE_SEQUENTIAL)
    eo = A.EntropyOptim(g, A.NodeParams.from_host(g, p0, s0), par, y0, hub_counts=hubc)
    nb_sample = 10 * len(nbr)
    for it in range(1, 6):
        eo.gradient_iteration_threaded(nb_sample, 2.0 * (1.0 - it / 5), it)
    y, ce1 = eo.get_embedded(), eo.ce_compute_threaded()
    yo, _, oce1 = oracle.entropy_optimize(indptr, nbr, p0, s0, y0, 5, hub_counts=hubc, b=b)
    assert np.isfinite(y).all() and y.shape == (n, dim)
    src = np.repeat(np.arange(n), k)
    lo = np.linalg.norm(yo[src] - yo[nbr], axis=1)

    def close(y_, ce_, tol_ce, tol_q):
        assert abs(ce_ - oce1) < tol_ce * oce1, (ce_, oce1)
        lg = np.linalg.norm(y_[src] - y_[nbr], axis=1)
        for q in (0.5, 0.9):
            assert abs(np.quantile(lg, q) - np.quantile(lo, q)) < tol_q * np.quantile(lo, q), (q, np.quantile(lg, q), np.quantile(lo, q))
    if b == 1.0:
        assert np.array_equal(y, yo)
    close(y, ce1, 1e-5, 1e-4)
    # The statistical modes: their runs are not reproducible (negatives are read as the memory system has them) and on these small graphs
    # -- a one-column layout above all -- single runs scatter by several per cent (the ordered mode at 1 column: q90 +31 % once in ~25
    # suite runs).  So: three seeds of the mode against three seeds of the exact mode (the seed above is pinned to the oracle bit for
    # bit), mean against mean with the standard error of both sides (tests/util.py: assert_means_close); no second chances.
    seeds = (4664397, 12345, 777)

    def metrics(y_, ce_):
        lg = np.linalg.norm(y_[src] - y_[nbr], axis=1)
        return [ce_, np.quantile(lg, 0.5), np.quantile(lg, 0.9)]

    def run_mode(mode, seed):
        h = A.EntropyOptim(g, A.NodeParams.from_host(g, p0, s0),
                           A.EmbedderParams(asked_dim=dim, nb_grad_batch=5, hubness_weighting=hub, b=b, ce_mode=mode, seed=seed), y0, hub_counts=hubc)
        for it in range(1, 6):
            h.gradient_iteration_threaded(nb_sample, 2.0 * (1.0 - it / 5), it)
        yy = h.get_embedded()
        assert np.isfinite(yy).all() and yy.shape == (n, dim)
        return h, yy, h.ce_compute_threaded()

    exact = [metrics(*run_mode(A.AE_CE_SEQUENTIAL, sd)[1:]) for sd in seeds]

    def check_mode(mode, expect=None):
        rows = []
        for sd in seeds:
            h, yy, ce_ = run_mode(mode, sd)
            if expect is not None:
                assert h.get_ce_mode() == expect
            close(yy, ce_, 0.12, 0.35)   # smoke only: a single run of 5 batches on 4000 nodes (the claim is the mean's, below)
            rows.append(metrics(yy, ce_))
        assert_means_close(rows, exact, ("ce", "q50", "q90"), (0.01, 0.01, 0.03), "dim %d k %d mode %d" % (dim, k, mode), k_se=4.0)

    check_mode(A.AE_CE_AUTO, expect=A.AE_CE_ORDERED)   # the default mode (the ordered dataflow at this size)
    if dim <= 16:
        check_mode(A.AE_CE_EVENT)
    check_mode(A.AE_CE_SLICED)   # every dimension / row length / sampler / exponent


def test_converged_run_matches_reference_quality(A, oracle):
    """Full schedule (dmap initialisation, 20 batches): the sequential mode IS the oracle's sequential run; the default mode
    (AE_CE_AUTO -> the ordered dataflow) and the event-ordered mode against it on the reference's own yardsticks -- final cross
    entropy and get_quality_estimate_from_edge_length (embedder.rs:620-753) -- within 3-8 %."""
    n, k, nb = 10000, 10, 20
    indptr, nbr, dist, _, _ = synthetic_graph(n=n, dim=10, k=k, seed=7, ncomp=8)
    g = A.KGraph(indptr, nbr, dist)
    rc, p0, s0 = oracle.to_proba_edges(indptr, nbr, dist, 1.0, 1.0)
    rc, y0, _ = oracle.dmap_embed_from_kgraph(indptr, nbr, dist, k, oracle.DiffusionParams(2, 5.0, 12))
    y0 = oracle.set_data_box(y0, 10.0)
    yo, _, oce = oracle.entropy_optimize(indptr, nbr, p0, s0, y0, nb)
    yd, _, ced = A.entropy_optimize(g, A.NodeParams.from_host(g, p0, s0), A.EmbedderParams(nb_grad_batch=nb, ce_mode=A.AE_CE_SEQUENTIAL), y0)
    assert np.array_equal(yd, yo) and abs(ced - oce) < 1e-11 * oce  # the parity mode: the oracle's run
    qo = A.quality_estimate_from_edge_length(g, yo, 30)
    for mode in (A.AE_CE_EVENT, A.AE_CE_AUTO):
        y, _, ce = A.entropy_optimize(g, A.NodeParams.from_host(g, p0, s0), A.EmbedderParams(nb_grad_batch=nb, ce_mode=mode), y0)
        assert abs(ce - oce) < 0.03 * oce, (mode, ce, oce)
        q = A.quality_estimate_from_edge_length(g, y, 30)
        assert abs(q.nb_without_match - qo.nb_without_match) < 0.08 * qo.nb_without_match  # (a count of ~10 % of the nodes: single runs scatter by 2-5 %)
        assert abs(q.mean_nbmatch - qo.mean_nbmatch) < 0.05 * qo.mean_nbmatch
        assert abs(q.median_ratio - qo.median_ratio) < 0.08 * qo.median_ratio
        assert abs(q.radii_quantiles[2] - qo.radii_quantiles[2]) < 0.05 * qo.radii_quantiles[2]


def test_hub_and_ragged_rows(A, oracle):
    """a node that is everybody's neighbour (in-degree n - 1: in the event-ordered kernel its event list is sorted by the
    whole wave and it serves runs of target events per trip; in the rounds kernel its pushes overflow the in-edge windows)
    and rows of unequal length"""
    rng = np.random.default_rng(4)
    n = 3000
    x = rng.normal(size=(n, 4)).astype(np.float32)
    ip0, nb0, ds0 = knn_graph(x, 7)
    rows_n, rows_d, ptr = [], [], [0]
    for i in range(n):
        keep = 7 if i % 4 else 4  # ragged
        nb_i = nb0[i * 7:i * 7 + keep].copy()
        d_i = ds0[i * 7:i * 7 + keep].copy()
        if i != 0 and 0 not in nb_i:
            nb_i[-1] = 0  # the hub, as the farthest neighbour
        rows_n.append(nb_i)
        rows_d.append(d_i)
        ptr.append(ptr[-1] + keep)
    indptr, nbr, dist = np.array(ptr, np.uint64), np.concatenate(rows_n).astype(np.uint32), np.concatenate(rows_d).astype(np.float32)
    g = A.KGraph(indptr, nbr, dist)
    assert g.hubness()[0] >= n - 1
    rc, p0, s0 = oracle.to_proba_edges(indptr, nbr, dist, 1.0, 1.0)
    assert rc == 0
    y0 = oracle.set_data_box(rng.normal(size=(n, 2)).astype(np.float32), 10.0)
    yo, oce0, oce1 = oracle.entropy_optimize(indptr, nbr, p0, s0, y0, 5)
    yd, _, ced = A.entropy_optimize(g, A.NodeParams.from_host(g, p0, s0), A.EmbedderParams(nb_grad_batch=5, ce_mode=A.AE_CE_SEQUENTIAL), y0)
    assert np.array_equal(yd, yo)  # sequential mode: the oracle's run, hub or not
    src = np.repeat(np.arange(n), np.diff(indptr.astype(np.int64)))
    lo = np.linalg.norm(yo[src] - yo[nbr], axis=1)
    for mode in (A.AE_CE_EVENT, A.AE_CE_AUTO, A.AE_CE_SLICED):  # (the default resolves to the ordered dataflow here)
        y, ce0, ce1 = A.entropy_optimize(g, A.NodeParams.from_host(g, p0, s0), A.EmbedderParams(nb_grad_batch=5, ce_mode=mode), y0)
        assert np.isfinite(y).all() and abs(ce0 - oce0) < 2e-6 * oce0
        # run-to-run spread of these modes on this 3000-node star (12 runs each, tools/run_hub_spread.py, round 3): final CE / oracle
        # event 0.964-0.999, ordered 0.970-1.006, sliced 0.952-0.988 (mean 0.970); median edge length 0.98-1.06, 0.99-1.04, 1.00-1.05.
        # The bars are the spread plus a margin (0.05 / 0.06 failed about one run in ten).
        assert abs(ce1 - oce1) < 0.08 * oce1, (mode, ce1, oce1)
        lg = np.linalg.norm(y[src] - y[nbr], axis=1)
        assert abs(np.median(lg) - np.median(lo)) < 0.10 * np.median(lo), mode
    # the rounds mode on the same graph (rounds sized by the largest in-weight: 176 here): measured 0.89x CE, median +3 %
    yr, _, cer = A.entropy_optimize(g, A.NodeParams.from_host(g, p0, s0), A.EmbedderParams(nb_grad_batch=5, ce_mode=A.AE_CE_HOGWILD), y0)
    assert np.isfinite(yr).all() and abs(cer - oce1) < 0.25 * oce1


def test_unsupported_shape_fails_loudly(A, oracle):
    indptr, nbr, dist, _, _ = synthetic_graph(n=600, dim=6, k=6, seed=1, ncomp=1)
    g = A.KGraph(indptr, nbr, dist)
    rc, p0, s0 = oracle.to_proba_edges(indptr, nbr, dist, 1.0, 1.0)
    y0 = oracle.set_data_box(np.random.default_rng(0).normal(size=(600, 40)).astype(np.float32), 10.0)
    eo = A.EntropyOptim(g, A.NodeParams.from_host(g, p0, s0), A.EmbedderParams(asked_dim=40, ce_mode=A.AE_CE_HOGWILD), y0)
    with pytest.raises(A.AnnembedError) as e:  # the rounds mode has no kernel for 64 columns: no silent fall-back to the racy per-sample kernel
        eo.gradient_iteration_threaded(1000, 1.0, 1)
    assert e.value.code == 1
    # AE_CE_AUTO shards through the time-sliced mode -- which refuses a partition with most of its edge mass across shards (random order):
    # with the range's first batch (a range with a communicator: on every rank alike when it is attached)
    sh = A.EntropyOptim(g, A.NodeParams.from_host(g, p0, s0), A.EmbedderParams(asked_dim=40), y0, node_lo=0, node_hi=300)
    with pytest.raises(A.AnnembedError) as e:
        sh.gradient_iteration_threaded(10 * sh.get_nb_edges(), 1.0, 1)
    assert e.value.code == 1 and "cross-shard" in str(e.value)
    y0 = y0[:, :2].copy()
    ev = A.EntropyOptim(g, A.NodeParams.from_host(g, p0, s0), A.EmbedderParams(asked_dim=2, ce_mode=A.AE_CE_EVENT), y0, node_lo=0, node_hi=300)
    with pytest.raises(A.AnnembedError) as e:  # the event-ordered kernel does not shard: says so instead of running something else
        ev.gradient_iteration_threaded(1000, 1.0, 1)
    assert e.value.code == 1


# ------------------------------------------------------------------------------------------------
# a3-a9 diffusion maps
# ------------------------------------------------------------------------------------------------
def test_dmap_laplacian_csr_golden(A):
    g = A.KGraph(GOLD["g_indptr"], GOLD["g_nbr"], GOLD["g_dist"])
    lap = A.DiffusionMaps(A.DiffusionParams(2, 5.0, 6)).laplacian_from_kgraph(g, force_repr=2)
    is_csr, n, nnz = lap.info()
    assert is_csr and n == 300
    ip, ind, val = lap.get_sym_kernel()
    assert np.array_equal(ip, GOLD["dmap_lap_indptr"]) and np.array_equal(ind, GOLD["dmap_lap_indices"])  # structure: exact
    assert _relmax(val, GOLD["dmap_lap_values"]) < 1e-5  # f32 sums in a different order (SURVEY hard part 6)
    v = lap.get_vectors()
    assert np.array_equal(v["normed_scales"], GOLD["dmap_normed"])
    assert _relmax(v["q_density"], GOLD["dmap_q"]) < 5e-6 and _relmax(v["beta_scales"], GOLD["dmap_beta_scales"]) < 5e-6
    assert _relmax(v["normalizer"], GOLD["dmap_normalizer"]) < 5e-6
    assert v["mean_scale"] == float(GOLD["dmap_mean_scale"])


@pytest.mark.parametrize("force_repr", [1, 2])
def test_dmap_laplacian_vs_oracle(A, oracle, graph, force_repr):
    indptr, nbr, dist, _ = graph
    g = A.KGraph(indptr, nbr, dist)
    lap = A.DiffusionMaps(A.DiffusionParams(2, 5.0, 12)).laplacian_from_kgraph(g, force_repr=force_repr)
    rc, ol = oracle.dmap_laplacian(indptr, nbr, dist, 8, oracle.DiffusionParams(2, 5.0, 12), force_repr=force_repr)
    assert rc == 0
    v = lap.get_vectors()
    assert np.array_equal(v["normed_scales"], ol["normed_scales"])
    assert _relmax(v["q_density"], ol["q"]) < 1e-5 and _relmax(v["normalizer"], ol["normalizer"]) < 1e-5
    if force_repr == 2:
        ip, ind, val = lap.get_sym_kernel()
        assert np.array_equal(ip, ol["csr"].indptr) and np.array_equal(ind, ol["csr"].indices)
        assert _relmax(val, ol["csr"].values) < 1e-5
    else:
        assert np.max(np.abs(lap.get_sym_kernel() - ol["dense"])) < 2e-6
    # do_svd: dense & n <= 5000 -> full svd ; else rank-20 / 5 iteration randomized svd (graphlaplace.rs:127-134)
    sv = lap.do_svd()
    so, uo = oracle.laplacian_do_svd(ol)
    assert _relmax(sv.s[:20], so[:20]) < (2e-5 if force_repr == 1 else 1e-4)
    assert abs(sv.s[0] - 1.0) < (1e-5 if force_repr == 1 else 5e-2)


def test_dmap_embedding_parity_up_to_sign(A, oracle, graph):
    indptr, nbr, dist, _ = graph
    g = A.KGraph(indptr, nbr, dist)
    y = A.DiffusionMaps(A.DiffusionParams(2, 5.0, 12)).embed_from_kgraph(g)
    rc, yo, info = oracle.dmap_embed_from_kgraph(indptr, nbr, dist, 8, oracle.DiffusionParams(2, 5.0, 12))
    assert rc == 0 and y.shape == yo.shape
    for c in range(2):  # singular vectors are defined up to sign
        sgn = np.sign(np.dot(y[:, c], yo[:, c]))
        err = np.max(np.abs(sgn * y[:, c] - yo[:, c])) / np.max(np.abs(yo[:, c]))
        # sigma_2 - sigma_3 = 2.2e-5 on this graph: f32 roundoff (1e-7) of two different SVD algorithms / summation orders
        # rotates the pair by ~5e-3; measured 1.7e-3 .. 2.1e-3
        assert err < 6e-3, (c, err)


def test_dense_laplacian_and_y0_golden_v2(A):
    """The dense branch (n <= 5000) of the laplacian, do_svd and the diffusion-map coordinates on the golden graph with healthy
    spectral gaps (three clusters in a chain: sigma_k - sigma_k+1 = 4e-3, 1.2e-2, 2e-2), HIP path vs the numpy restatement of
    tests/golden/make_golden_v2.py: q / beta scales / normalizer 1e-5, kernel 1e-5 of its largest entry, sigma[0..20] 2e-5,
    Y0 up to the sign of each column at 1e-4 of the box -- the north star's coordinate tolerance (the 6e-3 of the test above is
    a property of THAT graph's 2.2e-5 gap)."""
    g = A.KGraph(GOLD2["gap_indptr"], GOLD2["gap_nbr"], GOLD2["gap_dist"])
    dm = A.DiffusionMaps(A.DiffusionParams(2, 5.0, 12))
    lap = dm.laplacian_from_kgraph(g, force_repr=1)
    assert not lap.is_csr()
    v = lap.get_vectors()
    assert _relmax(v["q_density"], GOLD2["dense_q"]) < 1e-5 and _relmax(v["beta_scales"], GOLD2["dense_beta_scales"]) < 1e-5
    assert _relmax(v["normalizer"], GOLD2["dense_normalizer"]) < 1e-5
    assert np.max(np.abs(lap.get_sym_kernel() - GOLD2["dense_lap"])) < 1e-5 * np.abs(GOLD2["dense_lap"]).max()
    sv = lap.do_svd()
    assert np.max(np.abs(sv.s[:20] - GOLD2["dense_sigma"])) < 2e-5
    y0 = dm.embed_from_kgraph(g)
    assert y0.shape == GOLD2["dense_y0"].shape
    for c in range(2):
        nz = np.nonzero(GOLD2["dense_y0"][:, c])[0][0]
        sgn = np.sign(y0[nz, c])
        assert np.max(np.abs(sgn * y0[:, c] - GOLD2["dense_y0"][:, c])) < 1e-4 * np.abs(GOLD2["dense_y0"]).max(), c


@pytest.mark.parametrize("b,key", [(1.0, "sgd_y_after_b1"), (0.8, "sgd_y_after_b08")])
def test_sequential_sgd_golden_v2(A, b, key):
    """1 000 sequential SGD samples (src/embedder.rs:1167-1302) from the golden start with the build's Philox stream:
    AE_CE_SEQUENTIAL against the numpy restatement's committed vectors -- the plan (7 nodes of every sample) and the
    coordinates, bit for bit, b = 1 and b = 0.8 (the oracle is checked against the same vectors on the CPU)."""
    g = A.KGraph(GOLD["g_indptr"], GOLD["g_nbr"], GOLD["g_dist"])
    npar = A.NodeParams.from_host(g, GOLD["proba"], GOLD["scale"])
    eo = A.EntropyOptim(g, npar, A.EmbedderParams(ce_mode=A.AE_CE_SEQUENTIAL, b=b), GOLD["y_box"])
    nodes, _ = eo.plan(0, 1000, int(GOLD2["sgd_iter"]))
    assert np.array_equal(nodes, GOLD2["sgd_plan"])
    eo.gradient_iteration_threaded(1000, float(GOLD2["sgd_step"]), int(GOLD2["sgd_iter"]))
    assert np.array_equal(eo.get_embedded(), GOLD2[key])


def test_projection_init_golden_v2(A):
    """h_embed's projection initialisation alone (src/embedder.rs:245-269) through ae_projection_init, against the numpy
    restatement: 2e-6 absolute (libm vs device logf / cosf / sinf on values of order 1); the projected rows stay within the
    reference's clip(., 2) of their projection's row"""
    x = np.random.default_rng(3).normal(size=(40, 3)).astype(np.float32)
    ip_s, nb_s, d_s = knn_graph(x, 5)
    small = A.KGraph(ip_s, nb_s, d_s)
    large = A.KGraph(GOLD2["gap_indptr"], GOLD2["gap_nbr"], GOLD2["gap_dist"])
    proj = A.KGraphProjection(small, large, GOLD2["proj_node"], GOLD2["proj_dist"])
    y0 = proj.projection_init(GOLD2["proj_y_small"], 4664397)
    assert np.array_equal(y0[:40], GOLD2["proj_y_small"])
    assert np.max(np.abs(y0 - GOLD2["proj_y0"])) < 2e-6
    assert np.max(np.abs(y0[40:] - GOLD2["proj_y_small"][GOLD2["proj_node"][40:]])) <= 2.0 + 1e-6


def test_dmap_errors(A, graph):
    indptr, nbr, dist, _ = graph
    g = A.KGraph(indptr, nbr, dist)
    dp = A.DiffusionParams(2, 5.0, 12)
    dp._c.beta = 0.3  # bypass the setter: "beta cannot be > 0." exit at diffmaps.rs:827-830
    with pytest.raises(A.AnnembedError) as e:
        A.DiffusionMaps(dp).laplacian_from_kgraph(g)
    assert e.value.code == 9


# ------------------------------------------------------------------------------------------------
# a7-a8 tools::svdapprox -- the reference's own known-answer tests, on the GPU
# ------------------------------------------------------------------------------------------------
def _sigma_ok(computed, exact, eps):
    for i in range(len(computed)):
        if exact[i] > 0:
            assert abs(1.0 - computed[i] / exact[i]) < eps, (i, computed[i], exact[i])
        else:
            assert abs(exact[i] - computed[i]) < eps, (i, computed[i], exact[i])


def test_gpu_svd_wiki_rank_full(A):  # svdapprox.rs:1310
    r = A.SvdApprox(A.MatRepr.from_array2(GOLD["wiki"])).direct_svd(A.RangeRank(3, 8))
    assert len(r.s) == 3
    _sigma_ok(r.s, GOLD["wiki_sigma"], 1e-5)


def test_gpu_svd_wiki_csr_rank(A):  # svdapprox.rs:1497 (rank deficient matrix, rank 4 asked)
    import scipy.sparse as sp
    m = sp.csr_matrix(GOLD["wiki"].astype(np.float32))
    mat = A.MatRepr.from_csrmat(m.indptr, m.indices, m.data, (4, 5))
    r = A.SvdApprox(mat).direct_svd(A.RangeRank(4, 5))
    assert len(r.s) == 4
    _sigma_ok(r.s, GOLD["wiki_sigma"], 1e-5)
    assert np.max(np.abs((r.u * r.s) @ r.vt - GOLD["wiki"])) < 1e-5  # A = U S Vt
    q = np.random.default_rng(0).normal(size=(4, 3)).astype(np.float32)  # check_transpose_dense_mult_csr :1575
    assert np.max(np.abs(A.transpose_dense_mult_csr(q, mat) - q.T @ GOLD["wiki"])) < 1e-5


def test_gpu_range_approx_rank(A, oracle):  # svdapprox.rs:1231 in f32: residual relative to ||A||
    rng = np.random.default_rng(4)
    u, v = rng.normal(size=(503, 20)), rng.normal(size=(20, 503))
    mat = (u @ v).astype(np.float32)
    q = A.subspace_iteration(A.MatRepr.from_array2(mat), 20, 4)
    assert np.max(np.abs(q.T @ q - np.eye(20))) < 1e-5  # orthonormal columns
    resid = np.linalg.norm(mat - q @ (q.T @ mat)) / np.linalg.norm(mat)
    assert resid < 1e-5
    # rank deficient tall matrix (svdapprox.rs:1160 pattern): rank 26 of 30, 28 asked
    data = rng.normal(size=(30, 500)).astype(np.float32)
    for r_ in (3, 5, 7, 9):
        data[r_] = data[2]
    q = A.subspace_iteration(A.MatRepr.from_array2(data), 28, 2)
    assert np.linalg.norm(data - q @ (q.T @ data)) / np.linalg.norm(data) < 1e-5


@pytest.mark.parametrize("m,ncols,rank", [(4096, 128, 20), (2048, 256, 20), (1500, 32, 8), (700, 1024, 20), (1000, 784, 20)])
def test_gpu_dense_direct_svd_column_counts(A, oracle, m, ncols, rank):
    """Dense direct_svd (svdapprox.rs:721-799 on a Array2: the streaming MFMA product) where the column count leaves NO remainder trip
    (ncols % 32 == 0: the remainder's loads used to sit one past the row, for the last row past the end of A -- ADVICE r3) and
    where it leaves one (784), l % 4 == 0: singular values vs the oracle's LAPACK path, A ~ U S Vt on a matrix of exact rank."""
    rng = np.random.default_rng(m + ncols)
    a = (rng.normal(size=(m, rank)) @ rng.normal(size=(rank, ncols))).astype(np.float32)
    r = A.SvdApprox(A.MatRepr.from_array2(a)).direct_svd(A.RangeRank(rank, 3))
    s_ref = np.linalg.svd(a.astype(np.float64), compute_uv=False)[:rank]
    assert _relmax(r.s, s_ref) < 2e-4
    assert np.linalg.norm((r.u * r.s) @ r.vt - a) / np.linalg.norm(a) < 1e-4


def _mixture_rows(m, n, seed):
    """rows of configs[4]'s kind: components of 50 000 points, means N(0, 10^2), sigma 1 -- a spectrum with a clear head"""
    rng = np.random.default_rng(seed)
    ncomp = max(2, m // 50_000)
    means = rng.normal(size=(ncomp, n)) * 10.0
    lab = np.arange(m) % ncomp
    return (means[lab] + rng.standard_normal((m, n), dtype=np.float32)).astype(np.float32)


def test_gpu_dense_direct_svd_c5_columns_vs_oracle(A, oracle):
    """The dense range finder at configs[4]'s column count (svdapprox.rs:285-333, 721-799 on a 200 000 x 128 block of the data matrix;
    the tall panel's QR is deferred to the small side, svd.hip: TallFactor): singular values against the oracle's LAPACK path (same
    Omega stream, same algorithm), U orthonormal, A^T U = V S."""
    a = _mixture_rows(200_000, 128, 4)
    r = A.SvdApprox(A.MatRepr.from_array2(a)).direct_svd(A.RangeRank(20, 5))
    so, uo, vto = oracle.direct_svd(a, 20, 5)
    assert _relmax(r.s[:4], so[:4]) < 1e-4   # the four centres
    assert _relmax(r.s, so) < 3e-4           # the sketch's view of the flat noise floor: two f32 summation orders apart
    assert np.max(np.abs(r.u.T.astype(np.float64) @ r.u - np.eye(20))) < 1e-4
    assert np.max(np.abs(r.vt.astype(np.float64) @ r.vt.T - np.eye(20))) < 1e-4
    # the leading triplets agree with the oracle's up to the sign of a column (the tail of a rank-20 sketch of a 128-column matrix
    # with a flat noise floor is not unique to 1e-4: compare where the spectrum has gaps)
    lead = 4
    for j in range(lead):
        assert abs(abs(float(r.u[:, j] @ uo[:, j])) - 1.0) < 1e-3, j
    resid = np.linalg.norm(a.T.astype(np.float64) @ r.u - r.vt.T * r.s) / np.linalg.norm(r.s)
    assert resid < 1e-4, resid
    # the explicit Q of subspace_iteration (the same deferred path, with the one Y <- Y R^-1 at its end) spans the same range
    q = A.subspace_iteration(A.MatRepr.from_array2(a), 20, 5)
    assert np.max(np.abs(q.T.astype(np.float64) @ q - np.eye(20))) < 1e-4
    assert np.linalg.norm(q @ (q.T @ r.u[:, :lead]) - r.u[:, :lead]) < 1e-3


def test_gpu_dense_direct_svd_c5_shard_size_properties(A):
    """... and at a rank's share of configs[4] (6.25 M x 128, 3.2 GB): size-independent properties -- U orthonormal, sigma = the norms of
    A^T u, sigma_0 the spectral norm of the centre structure (>= the largest column-block norm / sqrt(m) bound), descending spectrum."""
    m, n = 6_250_000, 128
    a = _mixture_rows(m, n, 4)
    r = A.SvdApprox(A.MatRepr.from_array2(a)).direct_svd(A.RangeRank(20, 5), want_vt=False)
    assert np.all(np.diff(r.s) <= 1e-6 * r.s[0]) and r.s[-1] > 0
    g = np.zeros((20, 20))
    atu = np.zeros((n, 20))
    for b in range(0, m, 500_000):   # (blocked f64 accumulation on the host)
        ub = r.u[b:b + 500_000].astype(np.float64)
        g += ub.T @ ub
        atu += a[b:b + 500_000].astype(np.float64).T @ ub
    assert np.max(np.abs(g - np.eye(20))) < 2e-4, np.max(np.abs(g - np.eye(20)))
    assert _relmax(np.linalg.norm(atu, axis=0), r.s) < 2e-4
    # Rayleigh bound: no unit vector gives more than sigma_0; the all-ones direction of the row space gives a lower bound
    ones = np.ones(n) / np.sqrt(n)
    lower = np.sqrt(sum(float(np.sum((a[b:b + 500_000].astype(np.float64) @ ones) ** 2)) for b in range(0, m, 500_000)))
    assert r.s[0] >= lower * (1 - 1e-4)


def test_gpu_svd_sparse_vs_oracle(A, oracle):
    """rank-20 / 5 iteration direct_svd of a sparse symmetric matrix: same Omega stream, same algorithm"""
    import scipy.sparse as sp
    rng = np.random.default_rng(8)
    n = 3000
    m = sp.random(n, n, density=0.004, random_state=8, dtype=np.float32, format="csr")
    m = (m + m.T + sp.diags(np.linspace(1, 30, n).astype(np.float32))).tocsr()
    m.sort_indices()
    mat = A.MatRepr.from_csrmat(m.indptr, m.indices, m.data, (n, n))
    r = A.SvdApprox(mat).direct_svd(A.RangeRank(20, 5))
    so, uo, vto = oracle.direct_svd(oracle.CsrMat(m.indptr, m.indices, m.data, (n, n)), 20, 5)
    assert _relmax(r.s, so) < 1e-4
    assert np.max(np.abs(r.u.T @ r.u - np.eye(20))) < 1e-4


@pytest.mark.parametrize("dim,nbng", [(2, 6), (2, 50), (3, 12), (2, 200)])
def test_quality_estimate_grid_equals_brute_force(A, monkeypatch, dim, nbng):
    """The radius of the quality estimate (distance to the nbng-th nearest embedded point) through the uniform grid (2 / 3
    embedded dimensions, n >= 4096) against the O(n^2) brute-force graph of the embedded points: every per-node quantity
    identical (the two paths compute the same f32 sums; the grid's stopping rule is exact).  Embedding: tight clusters of very
    different densities plus duplicates and a sparse background -- the cases a uniform grid handles worst."""
    rng = np.random.default_rng(11)
    n = 20000
    centers = rng.normal(size=(12, dim)) * 4
    lab = rng.integers(0, 12, n)
    y = (centers[lab] + rng.normal(size=(n, dim)) * (0.002 + 0.5 * rng.random(12))[lab][:, None]).astype(np.float32)
    y[:500] = (rng.random((500, dim)) * 60 - 30).astype(np.float32)   # sparse background
    y[500:900] = y[900:1300]                                           # exact duplicates
    k = 5
    nbr = rng.integers(0, n, size=(n, k)).astype(np.uint32)
    nbr[nbr == np.arange(n, dtype=np.uint32)[:, None]] = 1
    g = A.KGraph(np.arange(n + 1, dtype=np.uint64) * np.uint64(k), nbr.reshape(-1), np.sort(rng.random((n, k)).astype(np.float32), axis=1).reshape(-1), k)
    a = A.quality_estimate_from_edge_length(g, y, nbng)
    monkeypatch.setenv("AE_DEBUG_KNOBS", "1")
    monkeypatch.setenv("AE_QUALITY_BRUTE", "1")
    b = A.quality_estimate_from_edge_length(g, y, nbng)
    assert np.array_equal(a.ratio_by_node, b.ratio_by_node) and np.array_equal(a.first_dist, b.first_dist)
    assert np.array_equal(a.radii_quantiles, b.radii_quantiles) and np.array_equal(a.ratio_quantiles, b.ratio_quantiles)
    assert a.nb_without_match == b.nb_without_match and a.mean_nbmatch == b.mean_nbmatch
    assert abs(a.mean_ratio - b.mean_ratio) <= 1e-12 * abs(b.mean_ratio)  # (a sum of f64 atomics: the order varies)


# ------------------------------------------------------------------------------------------------
# Embedder
# ------------------------------------------------------------------------------------------------
def test_embedder_one_step(A, oracle, graph):
    indptr, nbr, dist, _ = graph
    g = A.KGraph(indptr, nbr, dist)
    e = A.Embedder(g, A.EmbedderParams(nb_grad_batch=8))
    with pytest.raises(A.AnnembedError) as ex:  # results before embed(): state error instead of a Rust panic
        e.get_embedded()
    assert ex.value.code == 8
    assert e.embed() == 1
    y, y0 = e.get_embedded(), e.get_initial_embedding()
    rc, ref = oracle.one_step_embed(indptr, nbr, dist, 8, oracle.EmbedderParams(nb_grad_batch=8))
    assert rc == 0 and y.shape == (2500, 2) and np.isfinite(y).all()
    assert abs(np.abs(y0).max() - 5.0) < 1e-4  # set_data_box(10): max |coord| = 5
    for c in range(2):
        sgn = np.sign(np.dot(y0[:, c], ref["y0"][:, c]))
        assert np.max(np.abs(sgn * y0[:, c] - ref["y0"][:, c])) < 2e-2 * 5.0
    b, a = e.get_cross_entropy()
    assert abs(b - ref["ce_before"]) < 2e-2 * ref["ce_before"] and abs(a - ref["ce_after"]) < 0.06 * ref["ce_after"]
    assert np.array_equal(e.get_embedded_reindexed(), y)
    perm = np.random.default_rng(0).permutation(2500).astype(np.uint64)
    assert np.array_equal(e.get_embedded_reindexed(perm)[perm], y)


def test_embedder_tree_sums_initial_embedding(A, graph):
    """The dmap initialisation under an approximate CE mode replaces the single-lane reference-order sums by f64 tree
    sums (linalg.h TreeSums): the initial embedding moves by float rounding only."""
    indptr, nbr, dist, _ = graph
    g = A.KGraph(indptr, nbr, dist)
    y0 = {}
    for mode in (A.AE_CE_SEQUENTIAL, A.AE_CE_HOGWILD):
        e = A.Embedder(g, A.EmbedderParams(nb_grad_batch=2, ce_mode=mode))
        assert e.embed() == 1
        y0[mode] = e.get_initial_embedding()
    d = np.abs(y0[A.AE_CE_SEQUENTIAL] - y0[A.AE_CE_HOGWILD]).max()
    assert d < 1e-4 * 5.0, d
    assert abs(np.abs(y0[A.AE_CE_HOGWILD]).max() - 5.0) < 1e-4


def test_embedder_random_init_and_hubness(A, oracle, graph):
    indptr, nbr, dist, _ = graph
    g = A.KGraph(indptr, nbr, dist)
    p = A.EmbedderParams(nb_grad_batch=5, dmap_init=False, hubness_weighting=True, asked_dim=3)
    e = A.Embedder(g, p)
    assert e.embed() == 1
    y0 = e.get_initial_embedding()
    assert np.array_equal(y0, oracle.random_init(2500, 3, 1.0, p.seed))  # U(-.5,.5), same Philox stream
    assert np.array_equal(e.get_hubness(), oracle.hubness(indptr, nbr))
    assert np.isfinite(e.get_embedded()).all()


def test_embedder_hierarchical(A, oracle):
    """h_embed (embedder.rs:194-295) on a KGraphProjection: small graph = first nodes, projection = nearest small node"""
    x, _ = gaussian_mixture(6000, 8, 4, seed=12, spread=3.0)
    n_small = 800
    ip_s, nb_s, d_s = knn_graph(x[:n_small], 6)
    ip_l, nb_l, d_l = knn_graph(x, 6)
    d2 = ((x[:, None, :] - x[None, :n_small, :]) ** 2).sum(-1) if False else None
    xs = x[:n_small].astype(np.float64)
    dd = (x.astype(np.float64) ** 2).sum(1)[:, None] + (xs ** 2).sum(1)[None, :] - 2 * x.astype(np.float64) @ xs.T
    proj_node = dd.argmin(1).astype(np.uint32)
    proj_dist = np.sqrt(np.maximum(dd.min(1), 0)).astype(np.float32)
    small, large = A.KGraph(ip_s, nb_s, d_s), A.KGraph(ip_l, nb_l, d_l)
    proj = A.KGraphProjection(small, large, proj_node, proj_dist)
    p = A.EmbedderParams(nb_grad_batch=4, grad_factor=2, scale_rho=0.75, hubness_weighting=True)
    e = A.Embedder.from_hkgraph(proj, p)
    assert e.embed() == 1
    y = e.get_embedded()
    rc, ref = oracle.h_embed((ip_s, nb_s, d_s, 6), (ip_l, nb_l, d_l, 6), proj_node, proj_dist,
                             oracle.EmbedderParams(nb_grad_batch=4, grad_factor=2, scale_rho=0.75, hubness_weighting=True))
    assert rc == 0 and y.shape == (6000, 2) and np.isfinite(y).all()
    b, a = e.get_cross_entropy()
    assert abs(a - ref["ce_after"]) < 0.08 * ref["ce_after"], (a, ref["ce_after"])
    # projected points start near their projection (clip(.,2) noise, embedder.rs:265)
    y0 = e.get_initial_embedding()
    assert np.max(np.abs(y0[n_small:] - y0[proj_node[n_small:]])) <= 2.0 + 1e-5


# ------------------------------------------------------------------------------------------------
# BASELINE size (configs[1]: 60k nodes, k = 12): size-independent properties
# ------------------------------------------------------------------------------------------------
def test_full_size_properties(A):
    n, k = 60000, 12
    rng = np.random.default_rng(0)
    base = np.arange(n)
    nbr = np.stack([(base + off) % n for off in (1, 2, 3, 5, 8, 13, n - 1, n - 2, n - 3, n - 5, n - 8, n - 13)], 1).astype(np.uint32)
    dist = np.sort(rng.gamma(2.0, 1.0, size=(n, k)).astype(np.float32), axis=1)
    indptr = np.arange(n + 1, dtype=np.uint64) * np.uint64(k)
    g = A.KGraph(indptr, nbr.reshape(-1), dist.reshape(-1))
    npar = A.to_proba_edges(g, 1.0, 1.0)
    p, s = npar.get()
    p = p.reshape(n, k)
    assert np.allclose(p.sum(1), 1.0, atol=2e-5) and (p > 0).all() and (np.diff(p, axis=1) <= 1e-7).all()  # sorted dists -> sorted probas
    y0 = A.set_data_box(rng.normal(size=(n, 2)).astype(np.float32), 10.0)
    par = A.EmbedderParams(nb_grad_batch=5, ce_mode=A.AE_CE_HOGWILD)
    eo = A.EntropyOptim(g, npar, par, y0)
    nodes, w = eo.plan(0, 200000, 1)
    src, cnt = np.unique(nodes[:, 0], return_counts=True)
    assert nodes.max() < n and cnt.max() < 30  # uniform sources
    assert (nodes[:, 2:] != nodes[:, :1]).all() and (nodes[:, 2:] != nodes[:, 1:2]).all()
    S = 10 * eo.get_nb_edges()
    before = eo.get_embedded()
    eo.gradient_iteration_threaded(S, 0.0, 1)
    assert np.array_equal(eo.get_embedded(), before)  # step 0 is the identity at any size
    eo.gradient_iteration_threaded(S, 1.0, 2)
    y = eo.get_embedded()
    assert np.isfinite(y).all() and np.isfinite(eo.ce_compute_threaded())
    ms, cnt = eo.kernel_time()
    assert cnt == 2 and ms > 0
    drawn, rounds = eo.samples_drawn()  # Poisson(nb_sample) total per batch
    assert abs(drawn - 2 * S) < 6 * np.sqrt(2 * S) and rounds == 15  # 120 samples per node and batch, 8 per round
    # the same properties in the event-ordered mode
    ev = A.EntropyOptim(g, npar, A.EmbedderParams(nb_grad_batch=5, ce_mode=A.AE_CE_EVENT), y0)
    assert ev.get_ce_mode() == A.AE_CE_EVENT
    ev.gradient_iteration_threaded(S, 0.0, 1)
    assert np.array_equal(ev.get_embedded(), before)
    ev.gradient_iteration_threaded(S, 1.0, 2)
    assert np.isfinite(ev.get_embedded()).all() and np.isfinite(ev.ce_compute_threaded())
    drawn, windows = ev.samples_drawn()
    assert abs(drawn - 2 * S) < 6 * np.sqrt(2 * S) and 4 <= windows <= 64


# ------------------------------------------------------------------------------------------------
# multi-GPU plumbing that can be exercised on one GPU
# ------------------------------------------------------------------------------------------------
def test_device_coords_alias_and_sharded_hogwild(A, oracle, graph):
    """bench.py --gpus N all-gathers the coordinate rows through a torch view of the library's device buffer:
    the view must alias (no copy).  Two shards run one after the other on one GPU emulate the per-batch protocol
    of annembed_amd.dist.ShardedCE: every source node is sampled by exactly one shard."""
    import torch
    from annembed_amd.dist import device_tensor, shard_range
    indptr, nbr, dist, _ = graph
    g = A.KGraph(indptr, nbr, dist)
    rc, p0, s0 = oracle.to_proba_edges(indptr, nbr, dist, 1.0, 1.0)
    npar = A.NodeParams.from_host(g, p0, s0)
    y0 = oracle.set_data_box(np.random.default_rng(4).normal(size=(2500, 2)).astype(np.float32), 10.0)
    shards = []
    for r in range(2):
        lo, hi = shard_range(2500, 2, r)
        auto = A.EntropyOptim(g, npar, A.EmbedderParams(nb_grad_batch=5), y0, node_lo=lo, node_hi=hi)
        with pytest.raises(A.AnnembedError):  # AE_CE_AUTO on a shard = the time-sliced mode, which refuses a node order with > 10 % of the edge mass across shards
            auto.gradient_iteration_threaded(10 * auto.get_nb_edges(), 1.0, 1)
        del auto
        shards.append((lo, hi, A.EntropyOptim(g, npar, A.EmbedderParams(nb_grad_batch=5, ce_mode=A.AE_CE_HOGWILD), y0, node_lo=lo, node_hi=hi)))
    views = [device_tensor(eo) for _, _, eo in shards]
    assert views[0].shape == (2500, 2) and views[0].is_cuda
    views[0][7, 1] = 123.5  # write through torch, read through the C ABI
    torch.cuda.synchronize()
    assert shards[0][2].get_embedded()[7, 1] == np.float32(123.5)
    views[0][7, 1] = float(y0[7, 1])
    torch.cuda.synchronize()
    total = 0
    for it in range(1, 4):
        for lo, hi, eo in shards:
            eo.gradient_iteration_threaded(10 * eo.get_nb_edges(), 1.0, it)
        from annembed_amd import _lib
        _lib.check(_lib.load().ae_synchronize())
        merged = torch.cat([views[r][shards[r][0]:shards[r][1]] for r in range(2)])  # the all-gather
        for v in views:
            v.copy_(merged)
        torch.cuda.synchronize()
    ys = [eo.get_embedded() for _, _, eo in shards]
    assert np.array_equal(ys[0], ys[1]) and np.isfinite(ys[0]).all()
    drawn = sum(eo.samples_drawn()[0] for _, _, eo in shards)
    assert abs(drawn - 3 * 10 * len(nbr)) < 6 * np.sqrt(3 * 10 * len(nbr))
    # the union of the two shards behaves like the single-GPU run: same statistics
    full = A.EntropyOptim(g, npar, A.EmbedderParams(nb_grad_batch=5, ce_mode=A.AE_CE_HOGWILD), y0)
    for it in range(1, 4):
        full.gradient_iteration_threaded(10 * len(nbr), 1.0, it)
    ce_full = full.ce_compute_threaded()
    ce_sh = sum(eo.ce_compute_threaded() for _, _, eo in shards)
    assert abs(ce_sh - ce_full) < 0.2 * ce_full  # replicas are refreshed once per batch only


@pytest.mark.parametrize("dim", [4, 8, 16])
def test_sharded_hogwild_wide_rows(A, oracle, graph, dim):
    """the C4 / C5 arrangement: d = 8 / 16 (partner rows gathered by lane groups, own rows moved as coalesced blocks)
    with the nodes sharded -- shard boundaries that are not multiples of 64, rows of other shards only ever read"""
    import torch
    from annembed_amd import _lib
    from annembed_amd.dist import device_tensor, shard_range
    indptr, nbr, dist, _ = graph
    n = 2500
    g = A.KGraph(indptr, nbr, dist)
    rc, p0, s0 = oracle.to_proba_edges(indptr, nbr, dist, 1.0, 1.0)
    npar = A.NodeParams.from_host(g, p0, s0)
    y0 = oracle.set_data_box(np.random.default_rng(dim).normal(size=(n, dim)).astype(np.float32), 10.0)
    par = A.EmbedderParams(asked_dim=dim, nb_grad_batch=5, ce_mode=A.AE_CE_HOGWILD)
    shards = []
    for r in range(3):
        lo, hi = shard_range(n, 3, r)
        shards.append((lo, hi, A.EntropyOptim(g, npar, par, y0, node_lo=lo, node_hi=hi)))
    views = [device_tensor(eo) for _, _, eo in shards]
    for it in range(1, 4):
        before = [v.clone() for v in views]
        for lo, hi, eo in shards:
            eo.gradient_iteration_threaded(10 * eo.get_nb_edges(), 1.0 * (1 - it / 5), it)
        _lib.check(_lib.load().ae_synchronize())
        for r, (lo, hi, _) in enumerate(shards):  # a shard writes its own rows only
            other = torch.ones(n, dtype=torch.bool, device=views[r].device)
            other[lo:hi] = False
            assert torch.equal(views[r][other], before[r][other])
        merged = torch.cat([views[r][shards[r][0]:shards[r][1]] for r in range(3)])
        for v in views:
            v.copy_(merged)
        torch.cuda.synchronize()
    ys = [eo.get_embedded() for _, _, eo in shards]
    assert np.array_equal(ys[0], ys[1]) and np.array_equal(ys[0], ys[2]) and np.isfinite(ys[0]).all()
    full = A.EntropyOptim(g, npar, A.EmbedderParams(asked_dim=dim, nb_grad_batch=5, ce_mode=A.AE_CE_HOGWILD), y0)
    for it in range(1, 4):
        full.gradient_iteration_threaded(10 * len(nbr), 1.0 * (1 - it / 5), it)
    ce_full = full.ce_compute_threaded()
    ce_sh = sum(eo.ce_compute_threaded() for _, _, eo in shards)
    # measured 0.77 (d = 4, rows per lane) / 0.75 / 0.74 of the single-shard CE after 3 batches on 3 shards: the protocol's
    # own effect (remote rows are a batch old), the same for the lane-group kernels and the row-per-lane one
    assert abs(ce_sh - ce_full) < 0.35 * ce_full, (ce_sh, ce_full)


# ------------------------------------------------------------------------------------------------
# 8f-1 quality estimate
# ------------------------------------------------------------------------------------------------
def test_quality_estimate_vs_oracle(A, oracle, graph):
    indptr, nbr, dist, _ = graph
    g = A.KGraph(indptr, nbr, dist)
    emb = A.Embedder(g, A.EmbedderParams(nb_grad_batch=4))
    assert emb.embed() == 1
    rep = emb.get_quality_estimate_from_edge_length(20)
    y = emb.get_embedded()
    o = oracle.quality_estimate(indptr, nbr, y, 20)
    assert rep.nb_nodes == 2500 and rep.kgraph_nbng == 8 and rep.nbng == 20 and rep.quality == 0.0
    assert np.array_equal(rep.first_dist, o["first_dist"])  # f32 arithmetic in the same order: bit exact
    assert np.allclose(rep.ratio_by_node, o["ratio_by_node"], rtol=1e-12)
    assert rep.nb_without_match == o["nb_without_match"] and abs(rep.mean_nbmatch - o["mean_nbmatch"]) < 1e-12
    assert np.allclose(rep.radii_quantiles, o["radii_quantiles"], rtol=1e-12)
    assert np.allclose(rep.ratio_quantiles, o["ratio_quantiles"], rtol=1e-12)
    assert abs(rep.mean_ratio - o["mean_ratio"]) < 1e-9 * o["mean_ratio"] and rep.median_ratio == rep.ratio_quantiles[2]
    # stage-level entry on an arbitrary embedding, ragged rows
    ip2 = np.concatenate([[0], np.cumsum(np.where(np.arange(2500) % 3 == 0, 5, 8))]).astype(np.uint64)
    keep = np.concatenate([np.arange(int(indptr[i]), int(indptr[i]) + (5 if i % 3 == 0 else 8)) for i in range(2500)])
    g2 = A.KGraph(ip2, nbr[keep], dist[keep])
    y2 = np.random.default_rng(5).normal(size=(2500, 3)).astype(np.float32)
    r2 = A.quality_estimate_from_edge_length(g2, y2, 7)
    o2 = oracle.quality_estimate(ip2, nbr[keep], y2, 7)
    assert r2.nb_without_match == o2["nb_without_match"] and np.allclose(r2.ratio_quantiles, o2["ratio_quantiles"], rtol=1e-12)
    assert np.array_equal(r2.first_dist, o2["first_dist"])
    assert "a guess at quality" in str(r2)
    with pytest.raises(A.AnnembedError):
        A.Embedder(g, A.EmbedderParams()).get_quality_estimate_from_edge_length(10)  # before embed(): embedder.rs:633-636



# ------------------------------------------------------------------------------------------------
# 8f-3 adaptative range finder / RangeApproxMode::EPSIL
# ------------------------------------------------------------------------------------------------
def test_svd_wiki_epsil(A):
    """svdapprox.rs:1459 test_svd_wiki_csr_epsil and :1530 test_svd_wiki_full_epsil (here f32: 1e-5)"""
    import scipy.sparse as sp
    wiki = GOLD["wiki"].astype(np.float32)
    m = sp.csr_matrix(wiki)
    for mat, max_rank in ((A.MatRepr.from_csrmat(m.indptr, m.indices, m.data, (4, 5)), 10), (A.MatRepr.from_array2(wiki), 4)):
        res = A.SvdApprox(mat).direct_svd(A.RangePrecision(0.1, 5, max_rank))
        s = res.get_sigma()
        assert 3 <= len(s) <= 4
        for i in range(len(s)):
            exact = GOLD["wiki_sigma"][i]
            assert (abs(1 - s[i] / exact) < 1e-5) if exact > 0 else (abs(s[i]) < 1e-5)
        u, vt = res.get_u(), res.get_vt()
        assert np.allclose((u * s) @ vt, wiki, atol=1e-5)  # rank 3 matrix: exact reconstruction


def test_adaptative_range_finder_vs_oracle(A, oracle):
    """same stopping rule as the oracle restatement (rank found within the probes' randomness), orthonormal basis,
    residual ||A - Q Q^T A|| at the level the oracle reaches"""
    rng = np.random.default_rng(5)
    a = (rng.standard_normal((300, 15)) @ rng.standard_normal((15, 700))).astype(np.float32)
    a += 1e-4 * rng.standard_normal(a.shape).astype(np.float32)
    q = A.RangeApprox(A.MatRepr.from_array2(a), A.RangePrecision(0.05, 6, 40)).get_approximator()
    qo = oracle.adaptative_range_finder(a, 0.05, 6, 40)
    assert q.shape[0] == 300 and abs(q.shape[1] - qo.shape[1]) <= 2 and q.shape[1] >= 15
    assert np.allclose(q.T @ q, np.eye(q.shape[1]), atol=2e-5)
    res, reso = np.linalg.norm(a - q @ (q.T @ a)), np.linalg.norm(a - qo @ (qo.T @ a))
    assert res < 3 * reso + 1e-6 * np.linalg.norm(a), (res, reso)
    import scipy.sparse as sp
    sm = sp.random(2000, 1500, density=0.01, random_state=3, dtype=np.float32, format="csr")
    qs = A.adaptative_range_finder_matrep(A.MatRepr.from_csrmat(sm.indptr, sm.indices, sm.data, sm.shape), 0.1, 5, 30)
    assert qs.shape == (2000, 30) and np.allclose(qs.T @ qs, np.eye(30), atol=2e-5)  # full-rank input: stops at max_rank


def test_range_approx_epsil_stops_at_the_rank(A, oracle):
    """svdapprox.rs:1191 test_range_approx_epsil (there 3003 x 3003 of rank 200, asked 500, f64, residual < 1e-5): the
    finder must stop at the rank of the matrix.  f32 here: 1200 x 1200 of rank 100, asked 300, relative residual 1e-4."""
    rng = np.random.default_rng(5)
    m = n = 1200
    rank, asked = 100, 300
    u, v = rng.standard_normal((m, m)), rng.standard_normal((n, n))
    p = np.zeros((m, n))
    p[np.arange(rank), np.arange(rank)] = 1.0
    a = (u @ (p @ v)).astype(np.float32)
    q = A.RangeApprox(A.MatRepr.from_array2(a), A.RangePrecision(0.05, 8, asked)).get_approximator()
    assert rank <= q.shape[1] < rank + 2 * 8, q.shape
    assert np.allclose(q.T @ q, np.eye(q.shape[1]), atol=5e-5)
    a64 = a.astype(np.float64)
    q64 = q.astype(np.float64)
    assert np.linalg.norm(a64 - q64 @ (q64.T @ a64)) < 1e-4 * np.linalg.norm(a64)


@pytest.mark.parametrize("kind", ["weak", "rounds", "faithful"])
def test_bench_sharded_path_two_ranks_on_one_gpu(kind):
    """bench.py's N > 1 path end to end -- two processes, sharded node ranges, the owned rows exchanged once per batch,
    max-over-ranks timing, one JSON line from rank 0 -- with the two ranks sharing this box's single GPU over gloo (RCCL
    refuses duplicate devices; the collective is the only difference to the 8-GPU launch of the driver).  The default arrangement
    (faithful): strong scaling of the component-ordered kNN graph of Higgs-shaped points (here: small) in the default mode -- AE_CE_AUTO on
    a node range = the time-sliced mode; --rounds / --weak: the approximate rounds mode on a lattice / on MNIST-shaped shards."""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--backend", "gloo"]
    env = dict(os.environ)
    if kind == "faithful":   # plain `python3 bench.py --gpus 2`, as the driver calls it: bench.py starts its two ranks itself
        cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--backend", "gloo"]
        env = {k2: v2 for k2, v2 in os.environ.items() if k2 not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    cmd += {"weak": ["--weak", "--points-per-gpu", "6000", "--dim", "64"], "rounds": ["--rounds", "--scale-nodes", "20001"],
            "faithful": ["--scale-nodes", "64000"]}[kind]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 3 and j["scaling"] == ("weak" if kind == "weak" else "strong")
    assert j["config"]["samples_per_step"] == {"weak": 2 * 6000 * 12 * 10, "rounds": 20001 * 6 * 10, "faithful": 64000 * 6 * 10}[kind]
    assert j["value"] > 0 and np.isfinite(j["ce_after"]) and j["roofline"]["launches_per_batch"] >= 1
    assert len(lines[0]) < 4096 and r.stdout.strip().splitlines()[-1] == lines[0]   # the compact line, LAST
    if kind == "faithful":
        assert j["config"]["partition"]["components"] >= 1 and j["config"]["cross_shard_mass"] < 0.05   # globally shuffled ids, the library's partition
        assert "AE_CE_SLICED" in j["config"]["ce_mode"] and str(j["faithful"]).startswith("statistically")
        assert j["dtype"].startswith("f32 coordinates, f64 scalars")
    else:
        assert j["faithful"] is False


@pytest.mark.parametrize("kind", ["faithful", "faithful_dmap"])
def test_embedder_multi_gpu_entry_faithful_two_ranks_one_gpu(A, tmp_path, kind):
    """Embedder::embed (embedder.rs:183-371) on two ranks in the DEFAULT mode: with a communicator attached AE_CE_AUTO resolves to the
    time-sliced mode on each rank's node range.  The graphs come in the REFERENCE's kind of node order -- ids shuffled globally (file
    order carries no locality, kgraph.rs:489,500): embed() partitions them itself (partition.hip) and returns the rows in the caller's
    order.  faithful: 20 000 points in 8 well separated components, random start broadcast from rank 0 -- the components are packed whole,
    no edge crosses.  faithful_dmap: ONE component (20 000 points uniform in a square), diffusion-map start -- the component is bisected
    along the initialisation, a few per cent of the edge mass cross (the library then exchanges 16 times per batch).  Both ranks end with
    the same embedding; it is the reference's: three seeds against three seeds of the one-device embed() in the sequential mode, mean
    against mean (tests/util.py: assert_means_close)."""
    import json
    import subprocess
    import sys
    sys_argv = sys.argv
    sys.argv = ["bench.py"]
    import bench
    sys.argv = sys_argv
    n, k = 20000, 6
    if kind == "faithful":
        x, bounds = bench.mixture_points_gpu(n, 16, 8, seed=6, mean_sigma=10.0)
        indptr, nbr, dist = bench.component_knn_graph(A, x, bounds, k, permute_seed=11)   # ids shuffled globally
    else:
        x = np.random.default_rng(8).random((n, 2)).astype(np.float32)   # (rows in random order already: ids carry no locality)
        indptr, nbr, dist = A.KGraph.bruteforce_l2(x, k).get_neighbours()
    np.savez(tmp_path / "graph.npz", indptr=indptr, nbr=nbr, dist=dist)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = np.repeat(np.arange(n), k)

    def metrics(y_, ce_):
        lg = np.linalg.norm(y_[src] - y_[nbr], axis=1)
        return [ce_] + list(np.quantile(lg, [0.25, 0.5, 0.75]))
    seeds = (4664397, 12345, 777)
    rows, ia = [], None
    for sd in seeds:
        name = "annembed_test_%d_%s_%d" % (os.getpid(), kind, sd % 1000)
        procs = [subprocess.Popen([sys.executable, os.path.join(root, "tests", "embedder_shm_worker.py"), str(tmp_path), str(r), "2", name, kind],
                                  stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=root, env=dict(os.environ, AE_TEST_SEED=str(sd))) for r in range(2)]
        outs = [p.communicate(timeout=600) for p in procs]
        for p, (so, se) in zip(procs, outs):
            assert p.returncode == 0, (so[-1500:], se[-3000:])
        ya, yb = np.load(tmp_path / ("y_%s_rank0.npy" % kind)), np.load(tmp_path / ("y_%s_rank1.npy" % kind))
        ia, ib = np.load(tmp_path / ("y0_%s_rank0.npy" % kind)), np.load(tmp_path / ("y0_%s_rank1.npy" % kind))
        ca, cb = np.load(tmp_path / ("ce_%s_rank0.npy" % kind)), np.load(tmp_path / ("ce_%s_rank1.npy" % kind))
        assert np.array_equal(ia, ib) and np.array_equal(ya, yb) and np.isfinite(ya).all() and np.array_equal(ca, cb)
        reps = [json.load(open(tmp_path / ("part_%s_rank%d.json" % (kind, r)))) for r in range(2)]
        assert reps[0] == reps[1]   # rank 0's partition, broadcast
        if kind == "faithful":
            assert reps[0]["components"] == 8 and reps[0]["cross_mass"] == 0.0 and reps[0]["splits"] == 0
        else:
            assert reps[0]["splits"] >= 1 and 0.0 < reps[0]["cross_mass"] < 0.04, reps[0]
        rows.append(metrics(ya, ca[1]))
    print("two-rank embed() partition:", reps[0])
    # the same schedule on one device in the sequential mode (the caller's node order on both sides).  The random start depends on the seed,
    # the diffusion-map start does not; the initial cross entropy of the relabelled problem is the caller's problem's
    g = A.KGraph(indptr, nbr, dist)
    npar = A.to_proba_edges(g, 1.0, 1.0)
    exact = []
    for sd in seeds:
        par = A.EmbedderParams(nb_grad_batch=12, grad_step=1.0, ce_mode=A.AE_CE_SEQUENTIAL, seed=sd, dmap_init=kind == "faithful_dmap")
        e1 = A.Embedder(g, par)
        assert e1.embed() == 1
        exact.append(metrics(e1.get_embedded(), e1.get_cross_entropy()[1]))
        if sd == seeds[-1]:
            assert abs(ca[0] - e1.get_cross_entropy()[0]) < 1e-4 * ca[0]   # same seed, same start: the same initial cross entropy
    assert_means_close(rows, exact, ("ce", "q25", "q50", "q75"), (0.015, 0.04, 0.02, 0.04), "two-rank embed() [%s]" % kind)


def test_library_communicator_world_one(A, oracle, graph):
    """The RCCL entry points of the C ABI (ae_comm_*, include/annembed_hip.h) with one rank -- all that one GPU allows: the
    communicator initialises, attaches to a rounds-mode handle whose range tiles [0, n), the in-batch exchanges are no-ops
    that leave the result equal to the same run without a communicator, the all-reduce returns its input; attaching to a
    faithful-mode handle (which cannot shard) or to a partial range is refused."""
    from annembed_amd.dist import LibraryComm
    indptr, nbr, dist, _ = graph
    g = A.KGraph(indptr, nbr, dist)
    rc, p0, s0 = oracle.to_proba_edges(indptr, nbr, dist, 1.0, 1.0)
    npar = A.NodeParams.from_host(g, p0, s0)
    y0 = oracle.set_data_box(np.random.default_rng(4).normal(size=(2500, 2)).astype(np.float32), 10.0)
    comm = LibraryComm(0, 1)
    par = A.EmbedderParams(nb_grad_batch=5, ce_mode=A.AE_CE_HOGWILD)
    a, b = A.EntropyOptim(g, npar, par, y0), A.EntropyOptim(g, npar, par, y0)
    comm.attach(a, 4)
    for it in (1, 2):
        a.gradient_iteration_threaded(10 * len(nbr), 0.8, it)
        b.gradient_iteration_threaded(10 * len(nbr), 0.8, it)
    assert np.isfinite(a.get_embedded()).all()
    assert abs(a.ce_compute_threaded() - b.ce_compute_threaded()) < 0.05 * b.ce_compute_threaded()  # same mode, same draws; float order of the rounds differs
    assert comm.all_reduce_sum(1.25) == 1.25
    with pytest.raises(A.AnnembedError):
        comm.attach(A.EntropyOptim(g, npar, A.EmbedderParams(nb_grad_batch=5), y0), 1)  # on the whole graph AE_CE_AUTO is the ordered dataflow, which does not shard
    with pytest.raises(A.AnnembedError):
        comm.attach(A.EntropyOptim(g, npar, par, y0, node_lo=0, node_hi=1000), 1)  # one rank must own [0, n)
    comm.close()


def test_library_communicator_rccl_calls_world_one(A, oracle, graph, monkeypatch):
    """With the debug knob AE_COMM_FORCE the one-rank communicator does not skip its collectives: the in-place ncclAllGather of
    the owned rows, the all-gather of the node ranges and the f64 all-reduce really go through RCCL (dlopen'ed librccl.so.1) on
    the library stream -- the calls an 8-GPU run makes, with one participant.  Results must equal the run without them."""
    from annembed_amd.dist import LibraryComm
    monkeypatch.setenv("AE_DEBUG_KNOBS", "1")
    monkeypatch.setenv("AE_COMM_FORCE", "1")
    indptr, nbr, dist, _ = graph
    g = A.KGraph(indptr, nbr, dist)
    rc, p0, s0 = oracle.to_proba_edges(indptr, nbr, dist, 1.0, 1.0)
    npar = A.NodeParams.from_host(g, p0, s0)
    y0 = oracle.set_data_box(np.random.default_rng(4).normal(size=(2500, 8)).astype(np.float32), 10.0)
    comm = LibraryComm(0, 1)
    par = A.EmbedderParams(nb_grad_batch=5, ce_mode=A.AE_CE_HOGWILD, asked_dim=8)
    a, b = A.EntropyOptim(g, npar, par, y0), A.EntropyOptim(g, npar, par, y0)
    comm.attach(a, 15)
    for it in (1, 2):
        a.gradient_iteration_threaded(10 * len(nbr), 0.8, it)
        b.gradient_iteration_threaded(10 * len(nbr), 0.8, it)
    ya = a.get_embedded()
    assert np.isfinite(ya).all() and (np.abs(ya - y0).max(1) > 0).all()
    assert abs(a.ce_compute_threaded() - b.ce_compute_threaded()) < 0.05 * b.ce_compute_threaded()
    assert comm.all_reduce_sum(1.25) == 1.25
    comm.close()


def test_sharded_protocol_lockstep_four_shards(A, oracle, graph):
    """The sharded CE protocol with several exchanges per batch, on one GPU: four rounds-mode shard handles run in lockstep
    (ae_entropy_optim_gradient_iteration_lockstep -- round for round and exchange for exchange what four processes with a
    communicator attached run).  After every batch the four replicas are identical; every row moved; the final cross entropy
    is compared with the UN-SHARDED SEQUENTIAL ORACLE.  NOT a parity test: the only mode that shards is the approximate rounds mode
    (asked for by name -- AE_CE_AUTO refuses a sharded range), and sharding adds to its distance from the reference: measured on
    this graph 0.69x / 0.85x / 0.91x at 1 / 4 / every-round exchanges (six batches from a random start).  The exact assertions here
    are the protocol's (identical replicas after every exchange, every row moved, ranges and modes refused when they must be); the
    ratio bars are a regression guard around those measurements, not a tolerance anybody should read as "matches the reference".  The 60 k-point measurements (1-8 shards, two graph families) are in
    profiles/r02/shard_fidelity_*.json (tools/run_shard_fidelity.py) and DESIGN 5."""
    from annembed_amd.dist import shard_range
    indptr, nbr, dist, _ = graph
    n = len(indptr) - 1
    g = A.KGraph(indptr, nbr, dist)
    rc, p0, s0 = oracle.to_proba_edges(indptr, nbr, dist, 1.0, 1.0)
    npar = A.NodeParams.from_host(g, p0, s0)
    y0 = oracle.set_data_box(np.random.default_rng(4).normal(size=(n, 2)).astype(np.float32), 10.0)
    nb_batch = 6
    yo, _, oce = oracle.entropy_optimize(indptr, nbr, p0, s0, y0, nb_batch, grad_step=1.0)
    par = A.EmbedderParams(nb_grad_batch=nb_batch, ce_mode=A.AE_CE_HOGWILD)
    ratios = {}
    for exch in (1, 4, 1000):
        shards = [A.EntropyOptim(g, npar, par, y0, node_lo=shard_range(n, 4, r)[0], node_hi=shard_range(n, 4, r)[1]) for r in range(4)]
        ns = [10 * sh.get_nb_edges() for sh in shards]
        assert sum(ns) == 10 * len(nbr)
        for it in range(1, nb_batch + 1):
            A.EntropyOptim.gradient_iteration_lockstep(shards, ns, 1.0 - it / nb_batch, it, exch)
            ys = [sh.get_embedded() for sh in shards]
            assert all(np.array_equal(y, ys[0]) for y in ys[1:]), (exch, it)
        assert np.isfinite(ys[0]).all() and (np.abs(ys[0] - y0).max(1) > 0).all()
        ratios[exch] = sum(sh.ce_compute_threaded() for sh in shards) / oce
        assert 0.55 < ratios[exch] < 1.25, ratios
    with pytest.raises(A.AnnembedError):  # ranges that do not tile [0, n)
        A.EntropyOptim.gradient_iteration_lockstep(shards[:3], ns[:3], 0.5, 1, 1)
    with pytest.raises(A.AnnembedError):  # a faithful-mode handle does not shard
        A.EntropyOptim.gradient_iteration_lockstep([A.EntropyOptim(g, npar, A.EmbedderParams(nb_grad_batch=5), y0)], [10 * len(nbr)], 0.5, 1, 1)
    assert ratios[1000] > 0.8 and ratios[4] > 0.7, ratios


def test_sharded_ce_hip_backend_two_ranks_one_gpu(A, oracle, graph, tmp_path):
    """Two processes (gloo; both on this box's one GPU) each run the HIP library on their shard of the source nodes and exchange
    the owned rows once per batch.  Checked against the UN-SHARDED SEQUENTIAL ORACLE, not against an emulation of the protocol:
    the replicas are identical after every exchange, every row moved, and the final cross entropy is within the distance the
    rounds mode + sharding are documented to have from the reference on this graph (measured 0.69-0.90x) -- a
    regression guard, not a parity claim: the sharded path is the approximate mode, by name (AE_CE_AUTO refuses a sharded range)."""
    import subprocess
    import sys
    import socket
    indptr, nbr, dist, _ = graph
    rc, p0, s0 = oracle.to_proba_edges(indptr, nbr, dist, 1.0, 1.0)
    y0 = oracle.set_data_box(np.random.default_rng(4).normal(size=(2500, 2)).astype(np.float32), 10.0)
    np.savez(tmp_path / "graph.npz", indptr=indptr, nbr=nbr, dist=dist, proba=p0, scale=s0, y0=y0)
    nb_batch = 6
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "tests", "sharded_hip_worker.py"), str(tmp_path), str(nb_batch)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    ya, yb = np.load(tmp_path / "y_rank0.npy"), np.load(tmp_path / "y_rank1.npy")
    assert np.array_equal(ya, yb) and np.isfinite(ya).all()
    assert (np.abs(ya - y0).max(1) > 0).all()
    yo, _, oce = oracle.entropy_optimize(indptr, nbr, p0, s0, y0, nb_batch, grad_step=1.0)
    ce = float(np.load(tmp_path / "ce.npy"))
    assert 0.6 * oce < ce < 1.15 * oce, (ce, oce)  # measured 0.69-0.90 over runs (six batches of the approximate mode from a random start)


@pytest.mark.parametrize("kind", ["flat", "hier"])
def test_embedder_multi_gpu_entry_two_ranks_one_gpu(A, graph, tmp_path, kind):
    """The multi-GPU embedding at the boundary the reference's callers use (Embedder::embed / from_hkgraph, embedder.rs:183-371;
    ae_embedder_set_comm): two processes on this box's one GPU, the library's communicator over shared memory (RCCL refuses
    two ranks on one device).  Both ranks hold bit-identical initial and final embeddings (rank 0's initialisation is broadcast;
    every batch ends with an exchange), the reported cross entropies are sums over the ranks, and the result is within the
    rounds mode's documented distance of the un-sharded run of the same mode."""
    import subprocess
    import sys
    indptr, nbr, dist, x = graph
    n = len(indptr) - 1
    extra = {}
    if kind == "hier":
        ns = n // 8
        si, sn, sd = knn_graph(x[:ns], 8)
        x64 = x.astype(np.float64)
        dd = (x64 ** 2).sum(1)[:, None] + (x64[:ns] ** 2).sum(1)[None, :] - 2 * x64 @ x64[:ns].T
        pn, pd = dd.argmin(1).astype(np.uint32), np.sqrt(np.maximum(dd.min(1), 0)).astype(np.float32)
        pn[:ns] = np.arange(ns)
        pd[:ns] = 0
        extra = dict(s_indptr=si, s_nbr=sn, s_dist=sd, proj_node=pn, proj_dist=pd)
    np.savez(tmp_path / "graph.npz", indptr=indptr, nbr=nbr, dist=dist, **extra)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    name = "annembed_test_%d_%s" % (os.getpid(), kind)
    procs = [subprocess.Popen([sys.executable, os.path.join(root, "tests", "embedder_shm_worker.py"), str(tmp_path), str(r), "2", name, kind],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=root) for r in range(2)]
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, (so[-1500:], se[-3000:])
    ya, yb = np.load(tmp_path / ("y_%s_rank0.npy" % kind)), np.load(tmp_path / ("y_%s_rank1.npy" % kind))
    ia, ib = np.load(tmp_path / ("y0_%s_rank0.npy" % kind)), np.load(tmp_path / ("y0_%s_rank1.npy" % kind))
    ca, cb = np.load(tmp_path / ("ce_%s_rank0.npy" % kind)), np.load(tmp_path / ("ce_%s_rank1.npy" % kind))
    assert np.array_equal(ia, ib) and np.array_equal(ya, yb) and np.isfinite(ya).all() and ya.shape == (n, 2)
    assert np.array_equal(ca, cb)  # the sums run in rank order on every rank
    # the same embedding un-sharded, same mode: the sharded run lands within the rounds mode's own envelope of it
    g = A.KGraph(indptr, nbr, dist)
    par = A.EmbedderParams(nb_grad_batch=6, ce_mode=A.AE_CE_HOGWILD, grad_step=1.0)
    if kind == "hier":
        e = A.Embedder.from_hkgraph(A.KGraphProjection(A.KGraph(extra["s_indptr"], extra["s_nbr"], extra["s_dist"]), g, extra["proj_node"], extra["proj_dist"]), par)
    else:
        e = A.Embedder(g, par)
    assert e.embed() == 1
    ce1 = e.get_cross_entropy()
    # CE of the initial embedding: flat = the same dmap initialisation up to its run-to-run rounding; hierarchical = a projection of the
    # first stage's result, itself a (sharded) stochastic optimisation
    assert abs(ca[0] - ce1[0]) < (2e-3 if kind == "flat" else 0.10) * ce1[0], (ca, ce1)   # (hier: two runs of the approximate rounds mode on differently ordered problems: measured 1 ... 5 %)
    assert 0.7 * ce1[1] < ca[1] < 1.3 * ce1[1], (ca, ce1)
