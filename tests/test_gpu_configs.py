"""GPU tests at the sizes of BASELINE.json's configs (run with -m gpu on an MI355X).

What is compared with what: the statistically faithful modes -- the default (AE_CE_AUTO -> the ordered dataflow up to 2^25 samples per
batch, the time-sliced mode beyond), AE_CE_EVENT, AE_CE_SLICED -- are held against the HIP SEQUENTIAL mode (AE_CE_SEQUENTIAL) on the
full schedules of the reference's examples, NOT against the oracle directly: the oracle's sequential loop takes minutes at these sizes.
That is sound only because the sequential mode itself is pinned to the oracle bit for bit elsewhere -- tests/test_gpu_parity.py:
test_sequential_sgd_bit_exact, test_ce_any_dim_and_row_length (every row stride) and
test_full_schedule_bit_exact_vs_oracle_at_config_size (the whole C1 / C2 schedules).  Where a second run is not affordable the default
path is run at full size with size-independent properties.

Tolerances: the sequential loop's own seed-to-seed spread at these sizes was measured at 0.3-2 % (final CE) and 2-5 % (edge
length quartiles); the bars below are 3 % / 5 % unless a comment says otherwise."""
import sys

import os

import numpy as np
import pytest

from tests.util import assert_means_close

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def A():
    import annembed_amd as A
    from annembed_amd import _lib
    _lib.load()
    return A


def _edge_q(indptr, nbr, y, qs=(0.25, 0.5, 0.75)):
    src = np.repeat(np.arange(len(indptr) - 1), np.diff(indptr.astype(np.int64)))
    return np.quantile(np.linalg.norm(y[src] - y[nbr], axis=1), qs)


def _blobs(n, dim=28, ncomp=64, seed=2):
    rng = np.random.default_rng(seed)
    means = rng.normal(size=(ncomp, dim)) * 2.0
    scales = 0.5 + rng.random((ncomp, dim))
    lab = rng.integers(0, ncomp, n)
    x = means[lab] + scales[lab] * rng.normal(size=(n, dim))
    x = (x - x.mean(0)) / x.std(0)
    return np.ascontiguousarray(x.astype(np.float32))


def _mnist_shaped_graph(A, n, k):
    import bench
    x = bench.synth_points(n, 784, seed=1)
    nb_t, ds_t = bench.knn_rows(x, 0, n, k)
    indptr = np.arange(n + 1, dtype=np.uint64) * np.uint64(k)
    nbr, dist = nb_t.cpu().numpy().astype(np.uint32).reshape(-1), ds_t.cpu().numpy().reshape(-1)
    return A.KGraph(indptr, nbr, dist, k), indptr, nbr


def _run_ce(A, g, npar, y0, nb_batch, mode, seed=4664397, hub=None):
    par = A.EmbedderParams(nb_grad_batch=nb_batch, ce_mode=mode, seed=seed, hubness_weighting=hub is not None, grad_step=1.0)
    eo = A.EntropyOptim(g, npar, par, y0, hub_counts=hub)
    S = 10 * eo.get_nb_edges()
    for it in range(1, nb_batch + 1):
        eo.gradient_iteration_threaded(S, 1.0 * (1 - it / nb_batch), it)
    return eo.get_embedded(), eo.ce_compute_threaded(), eo


def _assert_close(A, indptr, nbr, run, ref, tol_ce=0.03, tol_q=0.05):
    (y, ce, _), (yr, cer, _) = run, ref
    assert np.isfinite(y).all()
    q, qr = _edge_q(indptr, nbr, y), _edge_q(indptr, nbr, yr)
    print("close: ce ratio %.4f (bar %.2f), quantile ratios %s (bar %.2f)" % (ce / cer, tol_ce, np.round(q / qr, 3), tol_q))  # (pytest -s)
    assert abs(ce - cer) < tol_ce * cer, (ce, cer)
    assert np.all(np.abs(q - qr) < tol_q * qr), (q, qr)


def _metrics(indptr, nbr, y, ce):
    q = _edge_q(indptr, nbr, y)
    return [ce, q[0], q[1], q[2]]


METRIC_NAMES = ("ce", "q25", "q50", "q75")
SEEDS = (4664397, 12345, 777, 20261003)


def test_k6_blobs_without_hubness_40_batches(A):
    """The case the rounds mode misses by 28 % (final CE 0.72x, median edge 1.76x: 60 k points of 28-d blobs, k = 6, scale_rho 0.75, no
    hubness weighting, 40 batches from a random initialisation) -- and the stiff graph on which the time-sliced mode's repeat rule sits
    worst.  FOUR seeds of every statistically faithful mode against four seeds of the exact mode: the MEAN of the final
    cross entropy and of the median edge within 2 standard errors + 1 % of the exact mode's, the quartiles within 2 SE + 3 %
    (tests/util.py: assert_means_close).  Round 4's six single runs of the time-sliced mode's optimistic path on THIS graph read CE +2.1 %,
    quartiles -6 % in the mean; two four-seed comparisons of round 5 put it at +0.6 % / -1.5 % and +1.8 % / -4 % (2 SE 2-3 % / 4-7 %,
    profiles/r05/r5_fidelity_means.txt): a small bias of that sign is real (a kept repeat of an edge runs a pass later, not back to back),
    floors 2 % / 5 % / 3 % / 5 %, said here.  The class path (forced: the cost model runs 60 k nodes optimistically) read CE +1.7 %, quartiles
    -2.5 ... -3 % over four seeds, +0.4 ... +0.8 % / -1 ... -2 % over sixteen (both launch forms; optimistic path +1.1 % / -1.4 ... -2.8 %;
    r5_fidelity_means.txt).  Round 6 resolved the forms with 32 seeds a side (profiles/r06/r6_blobs_forms.txt): ONE LAUNCH PER CLASS is the
    exact mode's within a standard error whatever the palette (11 / 13 / 15 / 19 classes: CE 0.997-1.002, median edge 0.992-1.006; 2 SE 0.7 % /
    1.2 %) -- floors 1 % / 3 % / 1 % / 3 %; MERGED SLICES as rounds 4-5 ran them carried a bias of CE +1.1 %, quartiles -2 ... -2.5 % (three 32-seed
    runs: CE 1.009-1.011, median edge 0.978-0.981; floors 1.5 % / 3.5 % / 3 % / 3 % then).  Its cause is
    the AGE of the negatives' rows: a merged launch reads them as the slice found them; one launch per class with the negatives read from a
    snapshot taken every 1 / 4 / 16 slices reproduces the sign and a dose-response (CE 1.004 / 1.028 / 1.110, median edge 0.991 / 0.946 / 0.820).
    And CLOSED in round 6 by the class window (a workgroup of a merged launch reads its negatives once the class half a palette before
    its own is through; rows written through): 256 seeds a side against one launch per class, CE +0.12 +- 0.22 %, median edge -0.18 +- 0.43 %
    (without: +0.34 +- 0.23 % / -0.69 +- 0.43 %; profiles/r06/r6_blobs_window256.txt) -- merged slices run with it, and take the floors of
    one launch per class here.
    A four-seed mean of the median edge scatters by ~2 % on this graph (the runs of these modes are not repeatable seed by seed: the overflow
    class is scheduled by races): EIGHT seeds for the class path."""
    n = 60000
    g = A.KGraph.bruteforce_l2(_blobs(n), 6)
    indptr, nbr, _ = g.get_neighbours()
    npar = A.to_proba_edges(g, 0.75, 1.0)
    y0 = (np.random.default_rng(5).random(size=(n, 2)).astype(np.float32) - 0.5)
    assert A.EntropyOptim(g, npar, A.EmbedderParams(), y0).get_ce_mode() == A.AE_CE_ORDERED  # the default at this size

    def rows(mode, seeds=SEEDS):
        out = []
        for sd in seeds:
            y, ce, _ = _run_ce(A, g, npar, y0, 40, mode, seed=sd)
            assert np.isfinite(y).all()
            out.append(_metrics(indptr, nbr, y, ce))
        return out
    exact = rows(A.AE_CE_SEQUENTIAL)
    std = (0.01, 0.03, 0.01, 0.03)
    assert_means_close(rows(A.AE_CE_ORDERED), exact, METRIC_NAMES, std, "k6 blobs, ordered (the default)")
    assert_means_close(rows(A.AE_CE_EVENT), exact, METRIC_NAMES, (0.02, 0.04, 0.03, 0.04), "k6 blobs, event-ordered")   # (measured over seeds: CE +1 ... +2 %, quartiles -2 ... -4 %)
    os.environ.update({"AE_DEBUG_KNOBS": "1", "AE_SL_NO_MATCH": "1"})   # (the cost model takes the class path on a graph with hubs: merged slices)
    try:
        assert_means_close(rows(A.AE_CE_SLICED), exact, METRIC_NAMES, (0.02, 0.05, 0.03, 0.05), "k6 blobs, time-sliced, optimistic path")
    finally:
        os.environ.pop("AE_DEBUG_KNOBS", None)
        os.environ.pop("AE_SL_NO_MATCH", None)
    assert_means_close(rows(A.AE_CE_SLICED), exact, METRIC_NAMES, (0.02, 0.05, 0.03, 0.05), "k6 blobs, time-sliced as the cost model cuts it")
    rounds = _run_ce(A, g, npar, y0, 40, A.AE_CE_HOGWILD)  # evidence: the rounds mode is outside the envelope here
    assert rounds[1] < 0.85 * np.mean([r[0] for r in exact])
    # the CLASS path on a graph with hubs (in-degrees up to ~105): every class is a forest of in-stars (k + 5 = 11 classes whatever the
    # in-degrees; ~2 % of the edge mass finds no colour and runs optimistically); the events of a step that share their target run as a
    # chain through the target's row (ce_slice_kernels.h) -- on this graph the busiest row receives ~4 events per slice.
    # Both launch forms of the class path: every class of a slice in ONE launch, ordered node by node through the dependency words
    # (sl_slice_kernel: what steps this small take by default), and one launch per class (what full steps take).
    forced = {}
    for form, knob in (("merged slices", "AE_SL_MERGE"), ("one launch per class", "AE_SL_NO_MERGE")):
        knobs = {"AE_DEBUG_KNOBS": "1", "AE_SL_FORCE_CLASSES": "1", knob: "1"}
        saved = {k2: os.environ.get(k2) for k2 in knobs}
        os.environ.update(knobs)
        try:
            probe = A.EntropyOptim(g, npar, A.EmbedderParams(ce_mode=A.AE_CE_SLICED, nb_grad_batch=40), y0)
            classes, ov_frac, _, _ = probe.slice_info()
            del probe
            # (k + 5 = 11 classes; four more where the slices run merged -- a launch holds them all, and the overflow class all but empties)
            assert classes == (15 if form == "merged slices" else 11) and 0.0 < ov_frac <= 0.05, (classes, ov_frac)
            forced[form] = rows(A.AE_CE_SLICED, SEEDS + tuple(s + 1 for s in SEEDS))
        finally:
            for k2, v2 in saved.items():
                if v2 is None:
                    os.environ.pop(k2, None)
                else:
                    os.environ[k2] = v2
    floors = {"one launch per class": (0.01, 0.03, 0.01, 0.03), "merged slices": (0.01, 0.03, 0.01, 0.03)}   # (the claims of the docstring)
    for form, got in forced.items():
        assert_means_close(got, exact, METRIC_NAMES, floors[form], "k6 blobs, time-sliced, class path forced, " + form)
    # same events in the same order on every node: the two forms differ only in WHEN a finished row becomes visible to the negatives -- and
    # that is what the merged form's bias was made of (the snapshot experiment of the docstring; the class window bounds it)
    assert_means_close(forced["merged slices"], forced["one launch per class"], METRIC_NAMES, floors["merged slices"], "k6 blobs, merged slices against one launch per class")


@pytest.mark.parametrize("knobs,form", [({}, 2), ({"AE_SL_CHECK_FILL": "1"}, 2), ({"AE_SL_COMPOSITE_KEYS": "1"}, 2), ({"AE_SL_NO_LINES": "1"}, 1), ({"AE_SL_MERGE": "1"}, 5),
                                         ({"AE_SL_MERGE": "1", "AE_SL_WINDOW": "0"}, 3), ({"AE_SL_MERGE": "1", "AE_SL_WINDOW": "2"}, 5)])
def test_class_path_runs_every_event_once_in_every_layout(A, knobs, form):
    """One launch per class in its round-6 layouts -- node lines, the events sorted by their slice bits alone (the edges come in class order: the
    overflow class FIRST, key 0) -- and in the layouts they replaced (composite (slice, class position) keys, dense rows + static records),
    `AE_SL_CHECK_FILL`: the event fill's short cut (an edge without two events in one slice sends its slices out as drawn) against the fill
    with every edge's slices in order, both sorted, compared word for word on the device -- a difference fails the batch;
    and merged slices (with the class window of round 6 at its default half palette, without it, and two classes wide): on a graph with hubs and a real overflow class (200 k Higgs-shaped points, 8 columns, k = 6, the class path forced)
    every event of the batch's Poisson totals runs exactly once (the executed count within 6 sigma of nb_sample: a step pointer one class off
    would drop or double 1/15 of a slice), `ae_entropy_optim_slice_form` names the form, and the layouts end at the same cross entropy (5 %:
    one run against one run, three batches from a random start, the slices' class orders drawn differently: measured 2.1 % apart)."""
    n = 200000
    g = A.KGraph.bruteforce_l2(_blobs(n), 6)
    npar = A.to_proba_edges(g, 1.0, 1.0)
    y0 = A.set_data_box(np.random.default_rng(3).normal(size=(n, 8)).astype(np.float32), 10.0)
    hub = g.hubness()

    def run(extra):
        env = dict({"AE_DEBUG_KNOBS": "1", "AE_SL_FORCE_CLASSES": "1"}, **extra)
        if "AE_SL_MERGE" not in env:
            env["AE_SL_NO_MERGE"] = "1"
        saved = {k2: os.environ.get(k2) for k2 in env}
        os.environ.update(env)
        try:
            eo = A.EntropyOptim(g, npar, A.EmbedderParams(asked_dim=8, nb_grad_batch=10, ce_mode=A.AE_CE_SLICED, grad_step=1.0, hubness_weighting=True), y0, hub_counts=hub)
            S = 10 * eo.get_nb_edges()
            for it in (1, 2, 3):
                eo.gradient_iteration_threaded(S, 1.0 - it / 10, it)
            drawn, _ = eo.samples_drawn()
            classes, ov, _, _ = eo.slice_info()
            return drawn, 3 * S, classes, ov, eo.slice_form(), eo.ce_compute_threaded(), eo.get_embedded()
        finally:
            for k2, v2 in saved.items():
                if v2 is None:
                    os.environ.pop(k2, None)
                else:
                    os.environ[k2] = v2
    drawn, want, classes, ov, got_form, ce, y = run(knobs)
    assert classes >= 11 and 0.0 < ov < 0.05, (classes, ov)
    assert got_form == form, (got_form, form)
    assert np.isfinite(y).all()
    assert abs(drawn - want) < 6 * np.sqrt(want), (drawn, want, (drawn - want) / np.sqrt(want))
    if knobs:
        _, _, _, _, _, ce0, _ = run({})
        assert abs(ce - ce0) < 0.05 * ce0, (ce, ce0)


@pytest.mark.parametrize("k,nb_batch", [(6, 30), (12, 25)])
def test_c1_c2_full_schedule_from_dmap_init(A, k, nb_batch):
    """configs[0] / configs[1] shapes -- 60 000 x 784 MNIST-shaped points, k = 6 / 30 batches (examples/mnist_digits.rs:92-109)
    and k = 12 / 25 batches (examples/mnist_fashion.rs:92-110): Embedder::embed() in the event-ordered mode against the same
    call in the default mode (sequential; same dmap initialisation up to the summation order of its means -- reference-order
    single-lane sums under the bit-exact mode, f64 tree sums under the others: equal to float rounding)."""
    g, indptr, nbr = _mnist_shaped_graph(A, 60000, k)
    out = {}
    for name, mode in (("seq", A.AE_CE_SEQUENTIAL), ("auto", A.AE_CE_EVENT), ("sliced", A.AE_CE_SLICED), ("default", A.AE_CE_AUTO)):
        par = A.EmbedderParams(nb_grad_batch=nb_batch, scale_rho=1.0, beta=1.0, grad_step=1.0, nb_sampling_by_edge=10, dmap_init=True,
                               hubness_weighting=False, ce_mode=mode)
        e = A.Embedder(g, par)
        assert e.embed() == 1
        out[name] = (e.get_embedded(), e.get_cross_entropy()[1], e.get_initial_embedding())
    # (run to run the initialisation itself moves by an ulp in an element or two: the Gram of the CholeskyQR is a sum of f64 atomics)
    assert np.abs(out["seq"][2] - out["auto"][2]).max() < 1e-4 * 5.0 and np.abs(out["auto"][2] - out["sliced"][2]).max() < 1e-6 * 5.0
    assert abs(np.abs(out["auto"][2]).max() - 5.0) < 1e-4
    _assert_close(A, indptr, nbr, out["auto"][:2] + (None,), out["seq"][:2] + (None,))
    _assert_close(A, indptr, nbr, out["sliced"][:2] + (None,), out["seq"][:2] + (None,))
    _assert_close(A, indptr, nbr, out["default"][:2] + (None,), out["seq"][:2] + (None,))  # AE_CE_AUTO -> the ordered dataflow at these sizes


def test_c3_schedule_hierarchical_60k(A):
    """configs[2] schedule (examples/higgs.rs:204-242: hierarchical, grad_factor 5 x 40 batches on the small graph, 40 on the
    large one, scale_rho 0.75, hubness weighting) at 60 k points of the Higgs-shaped generator: the time-sliced, the event-ordered
    and the default mode against the sequential one, FOUR seeds a side through tests/util.py: assert_means_close (round 5: three-seed
    means against fixed 5 % / 18 % and 8 % / 25 % bars).  Two stages of stochastic optimisation in a strongly collapsed regime: the
    sequential pipeline itself moves by 2 % (CE) / 6 % (quartiles) from run to run here (its dmap initialisation is not bitwise
    reproducible), which the standard error carries.  Floors = the systematic distance CLAIMED: time-sliced and default 2 % on the
    cross entropy, 6 % on the quartiles (measured three-seed means over six runs of round 3: sliced CE 0.981-1.000, quartiles
    0.97-1.105; default 0.998-1.023 / 0.91-1.02); event-ordered 5 % / 12 % -- its known bias in this regime (CE +3 ... +5 %, quartiles
    -6 ... -17 %)."""
    n, k = 60000, 6
    x = _blobs(n)
    n_small = n // 24
    large, small = A.KGraph.bruteforce_l2(x, k), A.KGraph.bruteforce_l2(x[:n_small], k)
    x64 = x.astype(np.float64)
    dd = (x64 ** 2).sum(1)[:, None] + (x64[:n_small] ** 2).sum(1)[None, :] - 2 * x64 @ x64[:n_small].T
    pn = dd.argmin(1).astype(np.uint32)
    pd = np.sqrt(np.maximum(dd.min(1), 0)).astype(np.float32)
    pn[:n_small] = np.arange(n_small)
    pd[:n_small] = 0
    indptr, nbr, _ = large.get_neighbours()
    out = {}
    for name, mode in (("seq", A.AE_CE_SEQUENTIAL), ("sliced", A.AE_CE_SLICED), ("event", A.AE_CE_EVENT), ("default", A.AE_CE_AUTO)):
        rows = []
        for rep in range(4):
            par = A.EmbedderParams(asked_dim=2, nb_grad_batch=40, grad_factor=5, scale_rho=0.75, beta=1.0, grad_step=1.0,
                                   nb_sampling_by_edge=10, dmap_init=True, hubness_weighting=True, ce_mode=mode, seed=4664397 + rep)
            emb = A.Embedder.from_hkgraph(A.KGraphProjection(small, large, pn, pd), par)
            assert emb.embed() == 1
            y = emb.get_embedded()
            assert np.isfinite(y).all()
            rows.append([emb.get_cross_entropy()[1]] + [float(v) for v in _edge_q(indptr, nbr, y)])
        out[name] = rows
    names = ["ce", "q25", "q50", "q75"]
    for name, floors in (("sliced", [0.02, 0.06, 0.06, 0.06]), ("default", [0.02, 0.06, 0.06, 0.06]), ("event", [0.05, 0.12, 0.12, 0.12])):
        assert_means_close(out[name], out["seq"], names, floors, "c3 hierarchical 60k, %s / sequential" % name)


def test_c3_full_size_properties(A):
    """configs[2] at full size: 1 650 000 x 28 Higgs-shaped points, k = 6, hierarchical (small graph = first n / 24 points),
    through Embedder.from_hkgraph(...).embed() with the default mode (the ordered dataflow on the small graph, the time-sliced mode --
    merged slices -- on the large one: 99 M samples per batch, beyond AE_CE_AUTO's 2^25).  Size-independent properties: finite, centred initial box, embedding inside the reference's
    clipping envelope, CE reported for both ends."""
    import torch
    n, k = 1650000, 6
    x = _blobs(n)
    n_small = n // 24
    large, small = A.KGraph.bruteforce_l2(x, k), A.KGraph.bruteforce_l2(x[:n_small], k)
    xt = torch.from_numpy(x).cuda()
    xs = xt[:n_small]
    sq_s = (xs * xs).sum(1)
    pn = torch.empty(n, dtype=torch.int64, device="cuda")
    pd = torch.empty(n, dtype=torch.float32, device="cuda")
    for b in range(0, n, 65536):
        e = min(b + 65536, n)
        d2 = (xt[b:e] * xt[b:e]).sum(1)[:, None] + sq_s[None, :] - 2.0 * (xt[b:e] @ xs.T)
        v, i = d2.min(1)
        pn[b:e], pd[b:e] = i, v.clamp_min(0).sqrt()
    pn[:n_small] = torch.arange(n_small, device="cuda")
    pd[:n_small] = 0
    pn_h, pd_h = pn.cpu().numpy().astype(np.uint32), pd.cpu().numpy()
    del xt, xs, pn, pd
    torch.cuda.empty_cache()
    par = A.EmbedderParams(asked_dim=2, nb_grad_batch=40, grad_factor=5, scale_rho=0.75, beta=1.0, grad_step=1.0,
                           nb_sampling_by_edge=10, dmap_init=True, hubness_weighting=True)
    y_probe = np.zeros((n, 2), np.float32)
    probe = A.EntropyOptim(large, A.to_proba_edges(large, 0.75, 1.0), par, y_probe, hub_counts=large.hubness())
    assert probe.get_ce_mode() == A.AE_CE_SLICED  # what AE_CE_AUTO resolves to on the large graph (exact kNN with hubs: ordered 50 ms per batch, sliced -- merged slices -- 34)
    classes, ov, _, _ = probe.slice_info()
    assert classes == 15 and ov < 0.01, (classes, ov)   # the class path with the wide palette of merged slices, not the optimistic one
    del probe, y_probe
    emb = A.Embedder.from_hkgraph(A.KGraphProjection(small, large, pn_h, pd_h), par)
    assert emb.embed() == 1
    y, y0 = emb.get_embedded(), emb.get_initial_embedding()
    assert y.shape == (n, 2) and np.isfinite(y).all() and np.isfinite(y0).all()
    assert np.max(np.abs(y0[n_small:] - y0[pn_h[n_small:]])) <= 2.0 + 1e-5  # projected points start within clip(., 2) of their projection (embedder.rs:265)
    b, a = emb.get_cross_entropy()
    assert np.isfinite(b) and np.isfinite(a) and a > 0
    assert np.abs(y).max() < 1e3  # no blow-up: every step is clipped (c in [-0.49, 2], embedder.rs:1233,1293)


def test_c4_shape_single_gpu_properties(A):
    """configs[3] shape on ONE GPU: 11 M nodes, k = 6, asked_dim 8, on a ring-lattice graph whose node ids are randomly
    PERMUTED (positive edges are not memory-local).  AE_CE_AUTO resolves to the time-sliced mode (660 M samples per batch are
    beyond the sequential mode's default budget); the event-ordered kernel says it does not fit (more nodes than resident
    lanes); ONE batch of the time-sliced mode and of the sequential mode (105 GB of scratch) end at the same cross entropy
    within 1 % -- two independent executions of the same law at full size, which is what caught a dispatch of more than 2^32
    work items in the sequential planner in round 2 --; one batch of the rounds mode (660 M samples) keeps its invariants: samples drawn
    within 6 sigma of nb_sample, finite rows, every row moved, the box stays bounded."""
    n, k, d = 11_000_000, 6, 8
    rng = np.random.default_rng(3)
    perm = rng.permutation(n).astype(np.int64)
    inv = np.empty(n, np.int64)
    inv[perm] = np.arange(n)
    base = inv  # lattice position of every node id
    cols = [perm[(base + o) % n] for o in (1, 2, 3)] + [perm[(base - o) % n] for o in (1, 2, 3)]
    nbr = np.stack(cols, 1).astype(np.uint32).reshape(-1)
    dist = np.sort(rng.gamma(2.0, 1.0, size=(n, k)).astype(np.float32), axis=1).reshape(-1)
    indptr = np.arange(n + 1, dtype=np.uint64) * np.uint64(k)
    g = A.KGraph(indptr, nbr, dist, k)
    del cols, perm, inv
    npar = A.to_proba_edges(g, 1.0, 1.0)
    y0 = A.set_data_box(rng.normal(size=(n, d)).astype(np.float32), 10.0)
    S = 10 * len(nbr)
    ces = {}
    for name, mode in (("auto", A.AE_CE_AUTO), ("sequential", A.AE_CE_SEQUENTIAL)):
        h = A.EntropyOptim(g, npar, A.EmbedderParams(asked_dim=d, ce_mode=mode), y0)
        assert h.get_ce_mode() == (A.AE_CE_SLICED if name == "auto" else A.AE_CE_SEQUENTIAL)
        h.gradient_iteration_threaded(S, 0.5, 1)
        ces[name] = h.ce_compute_threaded()
        assert np.isfinite(h.get_embedded()).all()
        if name == "auto":
            drawn, _ = h.samples_drawn()
            assert abs(drawn - S) < 6 * np.sqrt(S)
            classes, overflow, _, slices = h.slice_info()  # at this size the slices run as conflict-free matchings
            # (forests of in-stars: k + 5 classes, widened while the slices are thinned to resident steps anyway: 14 at this size)
            assert 11 <= classes <= 15 and overflow < 0.05 and slices >= 240, (classes, overflow, slices)
        del h
    assert abs(ces["auto"] - ces["sequential"]) < 0.01 * ces["sequential"], ces
    with pytest.raises(A.AnnembedError):
        ev = A.EntropyOptim(g, npar, A.EmbedderParams(asked_dim=d, ce_mode=A.AE_CE_EVENT), y0)
        ev.gradient_iteration_threaded(1000, 1.0, 1)
    eo = A.EntropyOptim(g, npar, A.EmbedderParams(asked_dim=d, ce_mode=A.AE_CE_HOGWILD), y0)
    eo.gradient_iteration_threaded(S, 0.5, 1)
    y = eo.get_embedded()
    drawn, rounds = eo.samples_drawn()
    assert abs(drawn - S) < 6 * np.sqrt(S) and rounds >= 15
    assert np.isfinite(y).all() and np.abs(y).max() < 1e3
    assert (np.abs(y - y0).max(1) > 0).mean() > 0.999
    # hubness-weighted negatives at this size come from a tile of alias-table draws.  Every node of the lattice has in-degree 6,
    # so the weighted law IS the uniform one: the batch must land where the uniform sampler's batch did
    ce_uniform = eo.ce_compute_threaded()
    del eo
    eh = A.EntropyOptim(g, npar, A.EmbedderParams(asked_dim=d, ce_mode=A.AE_CE_HOGWILD, hubness_weighting=True), y0, hub_counts=g.hubness())
    eh.gradient_iteration_threaded(S, 0.5, 1)
    yh = eh.get_embedded()
    drawn, _ = eh.samples_drawn()
    assert abs(drawn - S) < 6 * np.sqrt(S) and np.isfinite(yh).all() and (np.abs(yh - y0).max(1) > 0).mean() > 0.999
    assert abs(eh.ce_compute_threaded() - ce_uniform) < 0.01 * ce_uniform


def test_c4_global_knn_graph_full_size(A):
    """configs[3] on its OWN graph at full size: the exact GLOBAL kNN graph of 11 M Higgs-shaped points (k = 6; the grouped producer:
    ~22 s) with the node ids shuffled globally -- what HNSW + the reference's IndexSet would hand over (kgraph.rs:440-579).  (1) The
    partitioner at 8 ranks: the graph falls into a few dozen components (next to no edge leaves its cluster), they are packed into the
    ranks with next to nothing crossing, balanced.  (2) One batch in the default mode (AE_CE_AUTO -> the time-sliced mode, hubness-weighted
    negatives as examples/higgs.rs:204-242) against one batch of the sequential mode from the same start: samples drawn within 6 sigma,
    finite rows, cross entropy within 1 %."""
    sys_argv = sys.argv
    sys.argv = ["bench.py"]
    import bench
    sys.argv = sys_argv
    n, k, d = 11_000_000, 6, 8
    gr = bench.config_graphs(A, "c4", permute_seed=9)
    assert gr["edges_leaving_their_cluster"] < 1e-4 and gr["knn_pairs"]["fallback_rows"] < 0.001 * n
    assert gr["knn_pairs"]["pruned_phase"] + gr["knn_pairs"]["inside_clusters"] < 0.05 * float(n) * n   # (brute force: all of n^2)
    g = A.KGraph(gr["indptr"], gr["nbr"], gr["dist"], k)
    hub = g.hubness()
    assert hub.max() >= 100   # real in-degree skew (hubs)
    order, ranges, rep = g.partition(8)
    print("configs[3] graph, 8 ranks:", rep)
    assert 8 <= rep["components"] <= 64 and rep["cross_mass"] < 0.01 and rep["cross_mass_worst_rank"] < 0.02 and rep["imbalance"] < 0.03, rep
    del order
    npar = A.to_proba_edges(g, 1.0, 1.0)
    y0 = A.set_data_box(np.random.default_rng(3).normal(size=(n, d)).astype(np.float32), 10.0)
    S = 10 * len(gr["nbr"])
    ces = {}
    for name, mode in (("auto", A.AE_CE_AUTO), ("sequential", A.AE_CE_SEQUENTIAL)):
        h = A.EntropyOptim(g, npar, A.EmbedderParams(asked_dim=d, ce_mode=mode, hubness_weighting=True), y0, hub_counts=hub)
        assert h.get_ce_mode() == (A.AE_CE_SLICED if name == "auto" else A.AE_CE_SEQUENTIAL)
        h.gradient_iteration_threaded(S, 0.5, 1)
        ces[name] = h.ce_compute_threaded()
        assert np.isfinite(h.get_embedded()).all()
        if name == "auto":
            drawn, _ = h.samples_drawn()
            assert abs(drawn - S) < 6 * np.sqrt(S)
            classes, overflow, _, slices = h.slice_info()
            max_in, _ = h.slice_hub_info()
            assert 11 <= classes <= 15 and overflow < 0.05 and slices >= 240 and max_in == hub.max(), (classes, overflow, slices, max_in)
        del h
    print("configs[3] graph, one batch: CE default / sequential = %.4f" % (ces["auto"] / ces["sequential"]))
    assert abs(ces["auto"] - ces["sequential"]) < 0.01 * ces["sequential"], ces


def test_c5_shape_one_shard_properties(A):
    """configs[4] shape, one GPU's share of it: the full 50 M-node graph (k = 10, ring lattice with randomly PERMUTED node ids)
    and the full 50 M x 16 coordinate replica on the device, this rank owning the first eighth of the nodes (6.25 M sources,
    62.5 M edges, 625 M samples per batch).  The rounds mode, asked for by name (the d = 16 node kernel with tile negatives; the
    faithful mode at this size: test_c5_full_size_default_mode_one_gpu below, un-sharded); one batch keeps its invariants: samples drawn within 6 sigma of the shard's nb_sample,
    finite rows, every owned row moved, NO row outside the shard touched (owner computes), the box stays bounded."""
    n, k, d, world = 50_000_000, 10, 16, 8
    rng = np.random.default_rng(4)
    perm = rng.permutation(n).astype(np.int64)
    inv = np.empty(n, np.int64)
    inv[perm] = np.arange(n)
    nbr = np.empty((n, k), np.uint32)
    for c, o in enumerate((1, 2, 3, 4, 5, -1, -2, -3, -4, -5)):
        nbr[:, c] = perm[(inv + o) % n]
    del perm, inv
    dist = np.sort(rng.random((n, k), dtype=np.float32) + 0.05, axis=1).reshape(-1)
    indptr = np.arange(n + 1, dtype=np.uint64) * np.uint64(k)
    g = A.KGraph(indptr, nbr.reshape(-1), dist, k)
    del nbr, dist
    npar = A.to_proba_edges(g, 1.0, 1.0)
    y0 = A.set_data_box(rng.standard_normal((n, d), dtype=np.float32), 10.0)
    hi = n // world
    h = A.EntropyOptim(g, npar, A.EmbedderParams(asked_dim=d, ce_mode=A.AE_CE_HOGWILD), y0, node_lo=0, node_hi=hi)
    assert h.get_ce_mode() == A.AE_CE_HOGWILD
    S = 10 * h.get_nb_edges()
    assert S == 10 * hi * k
    h.gradient_iteration_threaded(S, 0.5, 1)
    drawn, rounds = h.samples_drawn()
    assert abs(drawn - S) < 6 * np.sqrt(S) and rounds >= 8
    y = h.get_embedded()
    assert np.isfinite(y).all() and np.abs(y).max() < 50
    assert (np.abs(y[:hi] - y0[:hi]).max(1) > 0).all()
    assert np.array_equal(y[hi:], y0[hi:])


def test_c5_full_size_default_mode_one_gpu(A):
    """configs[4] WHOLE, in the mode AE_CE_AUTO resolves to, on this one GPU: 50 M points of the 128-D mixture (1 000 components of
    50 000: SURVEY 8d's generator), exact kNN inside every component (k = 10: 500 M edges, in-degrees to ~5 500), node ids permuted,
    -> 16-D; 5 G samples per batch in 5 segments, ~55 GB of HBM, ~1.5 s per batch.  No second run is affordable at this size (the exact
    mode would take minutes per batch): size-independent properties -- the events executed are the Poisson totals (within 6 sigma of
    nb_sample per batch), every row finite and moved, the layout stays bounded, neighbours come together (median edge length of a
    2 M-edge sample falls below half its start)."""
    import bench
    gr = bench.config_graphs(A, "c5_full")
    n, k, d = gr["n"], gr["k"], 16
    assert n == 50_000_000 and k == 10
    indptr, nbr = gr["indptr"], gr["nbr"]
    g = A.KGraph(indptr, nbr, gr["dist"], k)
    del gr["dist"]
    npar = A.to_proba_edges(g, 1.0, 1.0)
    y0 = A.set_data_box(np.random.default_rng(1).normal(size=(n, d)).astype(np.float32), 10.0)
    eo = A.EntropyOptim(g, npar, A.EmbedderParams(asked_dim=d, nb_grad_batch=20, grad_step=1.0), y0)   # ce_mode: AE_CE_AUTO
    assert eo.get_ce_mode() == A.AE_CE_SLICED
    S = 10 * eo.get_nb_edges()
    assert S == 5_000_000_000
    for it in (1, 2):
        eo.gradient_iteration_threaded(S, 1.0 * (1 - it / 20), it)
    drawn, _ = eo.samples_drawn()
    assert abs(drawn - 2 * S) < 6 * np.sqrt(2 * S), (drawn, 2 * S)
    classes, ov, _, slices = eo.slice_info()
    assert classes == 18 and ov < 0.02 and slices >= 5 * 16, (classes, ov, slices)
    y = eo.get_embedded()
    assert np.isfinite(y).all() and np.abs(y).max() < 200
    pick = np.random.default_rng(3).integers(0, n * k, 2_000_000)
    src = pick // k
    before = np.median(np.linalg.norm(y0[src] - y0[nbr[pick]], axis=1))
    after = np.median(np.linalg.norm(y[src] - y[nbr[pick]], axis=1))
    print("configs[4] whole: median edge length %.3f -> %.3f" % (before, after))
    assert after < 0.5 * before
    assert (np.abs(y[::97] - y0[::97]).max(1) > 0).all()


def test_hub_stress_sliced_1m_nodes(A):
    """A node of in-degree 10 000 in a graph of 1 M nodes (k = 6).  The classes of the time-sliced mode are forests of in-stars: the
    hub's ~420 events of a slice are spread over the classes of its in-edges and run, step by step, as chains through the hub's row
    (handed from lane to lane, across chunk boundaries through memory) -- as the row's lock serialises them in the reference
    (embedder.rs:942,1185-1186,1239,1301).  Bounded time (faster than the sequential mode), every sample executed, CE within 5 % of the
    sequential mode's after the same batches, and the hub itself ends where the sequential mode puts it (within the spread of its
    neighbours)."""
    import time
    sys_argv = sys.argv
    sys.argv = ["bench.py"]
    import bench
    sys.argv = sys_argv
    n, k, d, hubs = 1_000_000, 6, 2, 10_000
    indptr, nbr, dist = bench.lattice_graph(n, k, seed=11, permute=True)
    nbr = nbr.reshape(n, k)
    rng = np.random.default_rng(3)
    src = rng.choice(np.arange(1, n), hubs, replace=False)
    src = src[(nbr[src] != 0).all(1)]          # (rows that already point at node 0 keep their edges: no duplicate neighbours)
    nbr[src, k - 1] = 0                          # the farthest neighbour of 10 000 random nodes becomes node 0
    g = A.KGraph(indptr, nbr.reshape(-1), dist, k)
    hub = g.hubness()
    assert hub[0] >= hubs - 10 and hub[1:].max() < 64
    npar = A.to_proba_edges(g, 1.0, 1.0)
    y0 = A.set_data_box(np.random.default_rng(1).normal(size=(n, d)).astype(np.float32), 10.0)
    S = 10 * n * k
    out = {}
    for name, mode in (("sliced", A.AE_CE_SLICED), ("sequential", A.AE_CE_SEQUENTIAL)):
        eo = A.EntropyOptim(g, npar, A.EmbedderParams(asked_dim=d, ce_mode=mode, nb_grad_batch=10), y0)
        t0 = time.perf_counter()
        for it in (1, 2, 3):
            eo.gradient_iteration_threaded(S, 1.0 - it / 10, it)
        y = eo.get_embedded()
        out[name] = (eo.ce_compute_threaded(), y, (time.perf_counter() - t0) / 3)
        if name == "sliced":
            cl, ovf, _, _ = eo.slice_info()
            drawn, _ = eo.samples_drawn()
            print("hub stress: classes %d overflow %.4f, hub info %s" % (cl, ovf, eo.slice_hub_info()))
            assert cl == 15 and ovf < 0.05 and eo.slice_hub_info()[0] >= hubs - 10   # the class path (merged slices: k + 5 + 4 classes), the hub in it
            assert abs(drawn - 3 * S) < 6 * np.sqrt(3 * S), (drawn, 3 * S)
    ce_s, y_s, t_s = out["sliced"]
    ce_q, y_q, t_q = out["sequential"]
    assert np.isfinite(y_s).all()
    assert abs(ce_s - ce_q) < 0.05 * ce_q, (ce_s, ce_q)
    assert t_s < 0.5, "time-sliced batch with a 10 000-in-degree hub took %.2f s (sequential %.2f s)" % (t_s, t_q)
    # the hub sits inside the cloud of its in-neighbours in both runs
    for y in (y_s, y_q):
        dn = np.linalg.norm(y[src] - y[0], axis=1)
        assert np.median(dn) < 3.0 * np.median(np.linalg.norm(y[src] - y[src].mean(0), axis=1)) + 1.0
    print("hub stress: sliced %.3f s/batch (CE %.4g), sequential %.3f s/batch (CE %.4g)" % (t_s, ce_s, t_q, ce_q))


def test_fashion_mnist_published_quality_if_data_present(A):
    """The only results the reference publishes on this path (src/embedder.rs:585-602: Fashion-MNIST 70 000 images, hierarchical,
    asked_dim 2, nbng 50 -> 20 260 neighbourhoods without a match, 5.069 neighbours conserved, median ratio 0.746).  Needs the IDX
    files (data/mnist/*-idx3-ubyte; reader annembed_amd.io, format src/utils/mnistio.rs:56-147): skipped without them
    (no network on the test boxes).  Bars are wide on purpose -- the kNN graph here is exact where the reference's comes from an
    HNSW, and the layer-1 subset is a random 1/16: 35 % on the count, 10 % on the matches, 35 % on the median ratio."""
    from annembed_amd import io as aio
    d = aio.find_mnist_dir()
    if d is None:
        pytest.skip("no Fashion-MNIST IDX files (AE_MNIST_DIR, data/fashion-mnist, data/mnist)")
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import run_fashion_quality as rfq
    x, _ = aio.mnist_images_as_vectors(d)
    pub = rfq.PUBLISHED[2]
    for mode in (A.AE_CE_SEQUENTIAL, A.AE_CE_AUTO):
        r = rfq.run(x, 2, False, mode, A)
        print("fashion quality, mode %d: %s (published %s)" % (mode, {k: r[k] for k in ("nb_without_match", "mean_nbmatch", "median_ratio")}, pub))
        assert r["embed_rc"] == 0
        assert abs(r["nb_without_match"] - pub["nb_without_match"]) < 0.35 * pub["nb_without_match"]
        assert abs(r["mean_nbmatch"] - pub["mean_nbmatch"]) < 0.10 * pub["mean_nbmatch"]
        assert abs(r["ratio_quantiles"][2] - pub["ratio_quantiles"][2]) < 0.35 * pub["ratio_quantiles"][2]


def test_high_dimensional_hubs_1m_nodes(A):
    """configs[4]'s kind of graph at a size the sequential mode still runs: 1 M points of the 128-D Gaussian mixture (20 components of
    50 000 points, SURVEY 8d's generator), exact kNN inside every component, k = 10, asked_dim 16.  Hubness is of another order in 128-D:
    in-degrees in the thousands (round 3: every faithful mode serialised a hub's events at one launch or ~3 us each -- 6x the lattice's
    time at the shard size).  The time-sliced mode runs them as chains through the hub's row: CE within 5 % of the sequential mode's
    after the same batches, every sample executed, bounded time."""
    import time
    sys_argv = sys.argv
    sys.argv = ["bench.py"]
    import bench
    sys.argv = sys_argv
    n, k, d = 1_000_000, 10, 16
    x, bounds = bench.mixture_points_gpu(n, 128, 20, seed=4, mean_sigma=10.0)
    indptr, nbr, dist = bench.component_knn_graph(A, x, bounds, k, permute_seed=9)
    del x
    g = A.KGraph(indptr, nbr, dist, k)
    hub = g.hubness()
    print("128-D mixture, 1 M nodes: max in-degree %d, 99.9 %% quantile %d, nodes nobody points at %d" % (hub.max(), np.quantile(hub, 0.999), (hub == 0).sum()))
    assert hub.max() >= 2000
    npar = A.to_proba_edges(g, 1.0, 1.0)
    y0 = A.set_data_box(np.random.default_rng(1).normal(size=(n, d)).astype(np.float32), 10.0)
    S = 10 * n * k
    out = {}
    for name, mode in (("sliced", A.AE_CE_SLICED), ("sequential", A.AE_CE_SEQUENTIAL)):
        eo = A.EntropyOptim(g, npar, A.EmbedderParams(asked_dim=d, ce_mode=mode, nb_grad_batch=10), y0)
        eo.gradient_iteration_threaded(S, 0.9, 1)
        eo.get_embedded()
        t0 = time.perf_counter()
        for it in (2, 3):
            eo.gradient_iteration_threaded(S, 1.0 - it / 10, it)
        y = eo.get_embedded()
        out[name] = (eo.ce_compute_threaded(), y, (time.perf_counter() - t0) / 2)
        if name == "sliced":
            cl, ovf, _, _ = eo.slice_info()
            drawn, _ = eo.samples_drawn()
            print("128-D mixture: classes %d overflow %.4f, hub info %s" % (cl, ovf, eo.slice_hub_info()))
            assert cl >= 14 and ovf < 0.05     # the class path (k + 8 classes), not the optimistic one
            assert abs(drawn - 3 * S) < 6 * np.sqrt(3 * S), (drawn, 3 * S)
    (ce_s, y_s, t_s), (ce_q, y_q, t_q) = out["sliced"], out["sequential"]
    print("128-D mixture: sliced %.3f s/batch (CE %.5g), sequential %.3f s/batch (CE %.5g), ratio %.4f" % (t_s, ce_s, t_q, ce_q, ce_s / ce_q))
    assert np.isfinite(y_s).all()
    assert abs(ce_s - ce_q) < 0.05 * ce_q, (ce_s, ce_q)
    assert t_s < 0.25, "time-sliced batch on the 128-D kNN graph took %.2f s (sequential %.2f s)" % (t_s, t_q)


def _run_sharded_sliced(A, tmp_path, indptr, nbr, dist, k, y0, scale_rho, world, exchanges, nb_batch, tag, seed=4664397):
    """`world` processes on this box's one GPU, the library's communicator over shared memory: the faithful sharded schedule"""
    import subprocess
    np.savez(tmp_path / "graph.npz", indptr=indptr, nbr=nbr, dist=dist, k=k, y0=y0, scale_rho=scale_rho)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    name = "annembed_sl_%d_%s_%d" % (os.getpid(), tag, seed % 100000)
    procs = [subprocess.Popen([sys.executable, os.path.join(root, "tests", "sliced_shm_worker.py"), str(tmp_path), str(r), str(world), name,
                               str(exchanges), str(nb_batch), str(seed)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=root) for r in range(world)]
    outs = [p.communicate(timeout=1200) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, (so[-1500:], se[-3000:])
    ys = [np.load(tmp_path / ("y_rank%d.npy" % r)) for r in range(world)]
    infos = [np.load(tmp_path / ("info_rank%d.npy" % r)) for r in range(world)]
    for r in range(1, world):
        assert np.array_equal(ys[0], ys[r])          # every batch ends with an exchange: the replicas agree
        assert infos[r][0] == infos[0][0]            # the cross-entropy sums run in rank order on every rank
    for inf in infos:                                 # every shard executed its own samples (half events are the other shard's)
        assert abs(inf[1] - inf[2]) < 6 * np.sqrt(inf[2]), inf
    return ys[0], float(infos[0][0]), float(infos[0][3])


@pytest.mark.parametrize("world", [2, 8])
def test_sharded_sliced_component_partition(A, tmp_path, world):
    """The faithful sharded CE loop (verdict r3, item 3) on a graph that partitions: 16 well separated components, node ids in component
    order, `world` shards whose boundaries are component boundaries -- no cross-shard edge; every rank runs AE_CE_AUTO (-> the time-sliced
    mode) on its own rows, negatives of other shards' nodes are read from the replica, refreshed 8 times per batch by the in-place
    all-gather.  Held against the un-sharded SEQUENTIAL mode (the reference's loop) on the full 20-batch schedule: final CE within 3 %,
    edge-length quartiles within 5 %.  The reference: one gradient on the current rows of both end points (embedder.rs:1228-1239),
    negatives through try_read (:1257-1265)."""
    sys_argv = sys.argv
    sys.argv = ["bench.py"]
    import bench
    sys.argv = sys_argv
    n, k, d, nb_batch = 64000, 6, 2, 20
    x, bounds = bench.mixture_points_gpu(n, 28, 16, seed=5, mean_sigma=10.0)
    indptr, nbr, dist = bench.component_knn_graph(A, x, bounds, k, permute_seed=None)
    y0 = A.set_data_box(np.random.default_rng(2).normal(size=(n, d)).astype(np.float32), 10.0)
    seeds = SEEDS if world == 2 else SEEDS[:3]
    rows = []
    for sd in seeds:
        y, ce, nbytes = _run_sharded_sliced(A, tmp_path, indptr, nbr, dist, k, y0, 1.0, world, 8, nb_batch, "comp%d" % world, seed=sd)
        assert nbytes == nb_batch * 8 * (n - n // world) * d * 4   # received: the other ranks' rows (equal shares here)
        assert np.isfinite(y).all()
        rows.append(_metrics(indptr, nbr, y, ce))
    g = A.KGraph(indptr, nbr, dist, k)
    npar = A.to_proba_edges(g, 1.0, 1.0)
    exact = []
    for sd in SEEDS:
        yr, cer, _ = _run_ce(A, g, npar, y0, nb_batch, A.AE_CE_SEQUENTIAL, seed=sd)
        exact.append(_metrics(indptr, nbr, yr, cer))
    print("sharded sliced, %d shards, component partition: %d bytes received per rank and batch" % (world, nbytes / nb_batch))
    assert_means_close(rows, exact, METRIC_NAMES, (0.01, 0.03, 0.01, 0.03), "sharded time-sliced, %d shards, component partition" % world)


def test_sharded_sliced_class_path_with_the_tile_on_component_ordered_labels(A, tmp_path):
    """The sharded mode's big-graph path at a size a test can afford: 600 k points in 64 components stored COMPONENT BY COMPONENT, 8
    columns, 2 shards, 4 exchanges per batch, the class path and the tile of negatives forced on (debug knobs, inherited by the rank
    processes; the thresholds use both from a few million nodes).  Attaching the communicator relabels the nodes at random inside every
    rank's range: without that a tile window is 16 nodes of one component, and an 11 M-node run in 2 shards came out at CE
    0.969 / edges +18-23 % of the exact mode's.  Three seeds a side against the un-sharded sequential mode, mean against mean (see the
    assertion for the floors and why)."""
    sys_argv = sys.argv
    sys.argv = ["bench.py"]
    import bench
    sys.argv = sys_argv
    n, k, d, nb_batch, world = 600000, 6, 8, 20, 2
    x, bounds = bench.mixture_points_gpu(n, 28, 64, seed=5, mean_sigma=10.0)
    indptr, nbr, dist = bench.component_knn_graph(A, x, bounds, k, permute_seed=None)
    y0 = A.set_data_box(np.random.default_rng(2).normal(size=(n, d)).astype(np.float32), 10.0)
    knobs = {"AE_DEBUG_KNOBS": "1", "AE_SL_FORCE_CLASSES": "1", "AE_SL_TILE_MIN": "1"}
    saved = {q: os.environ.get(q) for q in knobs}
    os.environ.update(knobs)
    rows = []
    try:
        for sd in SEEDS + (SEEDS[0] + 1, SEEDS[1] + 1):
            y, ce, nbytes = _run_sharded_sliced(A, tmp_path, indptr, nbr, dist, k, y0, 1.0, world, 4, nb_batch, "cls", seed=sd)
            assert nbytes == nb_batch * 4 * (n - n // world) * d * 4
            rows.append(_metrics(indptr, nbr, y, ce))
    finally:
        for q, v in saved.items():
            if v is None:
                os.environ.pop(q, None)
            else:
                os.environ[q] = v
    g = A.KGraph(indptr, nbr, dist, k)
    npar = A.to_proba_edges(g, 1.0, 1.0)
    exact = []
    for sd in SEEDS[:3]:
        par = A.EmbedderParams(asked_dim=d, nb_grad_batch=nb_batch, ce_mode=A.AE_CE_SEQUENTIAL, grad_step=1.0, seed=sd)
        eo = A.EntropyOptim(g, npar, par, y0)
        S = 10 * eo.get_nb_edges()
        for it in range(1, nb_batch + 1):
            eo.gradient_iteration_threaded(S, 1.0 * (1 - it / nb_batch), it)
        exact.append(_metrics(indptr, nbr, eo.get_embedded(), eo.ce_compute_threaded()))
        del eo
    # Round 4 held ONE run to -20 ... +30 % on the quartiles (nine single runs scattered 0.87 ... 1.19: two processes on one GPU interleave
    # differently from run to run, and tight components amplify it).  SIX sharded seeds against three exact ones, mean against mean: the
    # cross entropy within 3 SE + 2 %, the quartiles within 3 SE + 9 % -- the floors are what ce_slice.hip documents for this arrangement
    # (edges +4 ... +9 % at 600 k nodes in 2 shards with 4 exchanges per batch: the other shard's rows are a quarter of a batch old); in the
    # caller's labels, without the relabelling, the quartiles sat at 0.75 ... 0.84.  (Six repeats of the three-seed form of this test,
    # round 5 with merged slices: CE 0.988 ... 1.015, median edge 0.906 ... 1.088 around 1.001 / 0.997 -- and 2 SE estimated from three
    # runs anywhere between 3 % and 11 %: it failed once in ~10 suite runs.  Six seeds and 3 SE.)
    assert_means_close(rows, exact, METRIC_NAMES, (0.02, 0.09, 0.09, 0.09), "sharded class path + tile, 600 k nodes in component order, 2 shards, 4 exchanges", k_se=3.0)


def test_sharded_sliced_locality_partition_with_cross_edges(A, tmp_path):
    """The same with edges that DO cross shards: Higgs-shaped blobs (64 overlapping components, exact GLOBAL kNN graph, k = 6,
    scale_rho 0.75 -- the stiff graph on which the sharded rounds mode ends at 1.3-1.7x the reference's CE), node ids in
    component order, 8 shards: a few per cent of the edge mass crosses; such an edge fires as two half events, each shard moving its own
    end against its replica of the other.  16 exchanges per batch, 40 batches, three seeds against four of the un-sharded sequential mode:
    mean against mean (tests/util.py: assert_means_close) with the un-sharded time-sliced mode's own floors on this graph."""
    n, k, d, nb_batch, world = 60000, 6, 2, 40, 8
    sys_argv = sys.argv
    sys.argv = ["bench.py"]
    import bench
    sys.argv = sys_argv
    x, lab = bench.higgs_shaped_points(n, with_labels=True)
    order = np.argsort(lab, kind="stable")
    g = A.KGraph.bruteforce_l2(np.ascontiguousarray(x[order]), k)
    indptr, nbr, dist = g.get_neighbours()
    src = np.repeat(np.arange(n), k)
    cross = (src * world // n) != (nbr.astype(np.int64) * world // n)
    print("locality partition: %.2f %% of the edges cross shards" % (100 * cross.mean()))
    assert 0.001 < cross.mean() < 0.10
    y0 = (np.random.default_rng(5).random(size=(n, d)).astype(np.float32) - 0.5)
    rows = []
    for sd in SEEDS[:3]:
        y, ce, nbytes = _run_sharded_sliced(A, tmp_path, indptr, nbr, dist, k, y0, 0.75, world, 16, nb_batch, "loc", seed=sd)
        rows.append(_metrics(indptr, nbr, y, ce))
    npar = A.to_proba_edges(g, 0.75, 1.0)
    exact = []
    for sd in SEEDS:
        yr, cer, _ = _run_ce(A, g, npar, y0, nb_batch, A.AE_CE_SEQUENTIAL, seed=sd)
        exact.append(_metrics(indptr, nbr, yr, cer))
    # (the stiff graph: the un-sharded time-sliced mode's own floors here are 3 % / 8 %, test_k6_blobs_without_hubness_40_batches)
    assert_means_close(rows, exact, METRIC_NAMES, (0.03, 0.08, 0.08, 0.08), "sharded time-sliced, 8 shards, 0.9 % cross edges, stiff k = 6 graph")


def test_tile_of_negatives_is_unbiased_on_a_clustered_graph(A):
    """The LDS tile of negatives (ce_slice_kernels.h: TileShape) on the kind of graph that exposed its round 2-4 form: Higgs-shaped
    points in overlapping blobs (configs[3]'s generator), 1 M of them, kNN inside components, ids permuted, k 6 -> 8 columns, 20 batches
    from a random start.  Five slots of a 256-row tile repeat a row in 4 % of the samples and fall into one window in 62 %: measured then (1 M
    nodes) CE 0.90 of the exact mode's and the shortest quarter of the edges 1.5x longer -- lattices and single blobs never showed it.
    A sample now takes no window twice (draw_negatives): measured CE 1.00 ... 1.02 (what the time-sliced mode gives with gathered
    negatives, and its usual place against the exact mode), lower quartile 0.92 ... 1.00.  The tile is forced (the thresholds would
    not use it at this size), once on the class path and once on the optimistic one."""
    _tile_bias_case(A, permute_seed=9)


def test_tile_of_negatives_is_unbiased_when_the_labels_carry_locality(A):
    """The same graph with its nodes in COMPONENT order (64 runs of consecutive ids, one per blob).  A tile window -- 16 consecutive
    ids -- is then 16 points of one blob, and as long as the events were generated in the order of their targets' labels a workgroup's
    256 samples had their sources in one or two blobs as well: all of them met the same few blobs in a launch.  Measured (1 M nodes,
    class path, tile forced): CE 1.06 of the exact mode's, lower quartile 0.70, against 1.015 / 0.92 with gathered negatives.  The
    event-generation order now follows a hash of the target (ce_slice.hip: mix_node): a workgroup's samples are unrelated whatever the
    labels say; measured 1.016 / 0.88 ... 1.04."""
    _tile_bias_case(A, permute_seed=None)


def _tile_bias_case(A, permute_seed):
    import bench
    n, k, d, nb = 1000000, 6, 8, 20   # (the size and schedule of the measurements quoted above: tools/run_tile_bias.py)
    gr = bench.config_graphs(A, "c4", permute_seed=permute_seed, n_override=n)
    indptr, nbr, dist = gr["indptr"], gr["nbr"], gr["dist"]
    g = A.KGraph(indptr, nbr, dist, k)
    npar = A.to_proba_edges(g, 1.0, 1.0)
    y0 = A.set_data_box(np.random.default_rng(1).normal(size=(n, d)).astype(np.float32), 10.0)

    def run(mode, knobs, seed=11):
        saved = {q: os.environ.get(q) for q in knobs}
        os.environ.update(knobs)
        try:
            eo = A.EntropyOptim(g, npar, A.EmbedderParams(asked_dim=d, nb_grad_batch=nb, ce_mode=mode, grad_step=1.0, seed=seed), y0)
            S = 10 * eo.get_nb_edges()
            for it in range(1, nb + 1):
                eo.gradient_iteration_threaded(S, 1.0 * (1 - it / nb), it)
            return eo.get_embedded(), eo.ce_compute_threaded()
        finally:
            for q, v in saved.items():
                if v is None:
                    os.environ.pop(q, None)
                else:
                    os.environ[q] = v

    # the exact mode: the mean of three seeds (its CE scatters by 0.6 % on the permuted graph, 1.5 % on the component-ordered one)
    refs = [run(A.AE_CE_SEQUENTIAL, {}, seed=sd) for sd in (11, 22, 33)]
    cer = float(np.mean([r[1] for r in refs]))
    qr = np.mean([_edge_q(indptr, nbr, r[0]) for r in refs], axis=0)
    for knobs in ({"AE_DEBUG_KNOBS": "1", "AE_SL_TILE_MIN": "1", "AE_SL_FORCE_CLASSES": "1"}, {"AE_DEBUG_KNOBS": "1", "AE_SL_TILE_MIN": "1", "AE_SL_NO_MATCH": "1"}):
        y, ce = run(A.AE_CE_SLICED, knobs)
        q = _edge_q(indptr, nbr, y)
        print("tile forced, %s: ce ratio %.4f, quartile ratios %s" % ("class path" if "AE_SL_FORCE_CLASSES" in knobs else "optimistic path", ce / cer, np.round(q / qr, 3)))
        # the exact mode's own seed-to-seed spread here: CE 0.6-0.9 %, lower quartile 4-8 %, median 0.5-1 %; over ten seeds a side the
        # time-sliced mode's means are the exact mode's within a standard error
        assert 0.96 < ce / cer < 1.04, (ce, cer)                   # (round 4's tile: 0.90 ... 0.95; component order: 1.06)
        # (the lower quartile of ONE run against ONE run scatters by ~8 %; the time-sliced mode's sits at +3 ... +9 % since repeats of an
        # edge inside a slice stay together with the i.i.d. sequence's probability: ce_slice.hip sl_fill_kernel)
        assert 0.80 < q[0] / qr[0] < 1.25, (q, qr)                 # (round 4's tile: 1.25 ... 1.5; component order: 0.70)
        assert 0.95 < q[1] / qr[1] < 1.03, (q, qr)                 # (round 4's tile: 1.03 ... 1.07; component order: 0.93)


def test_negative_rows_are_not_read_torn_in_the_time_sliced_mode(tmp_path):
    """Torn rows (verdict r3, item 7).  The faithful modes read a negative's row while its owner may be rewriting it; the reference never sees
    half a row (`try_read`, embedder.rs:1257-1265).  tools/ubench_torn_rows.hip hammers 4 096 hot rows with 512 writer waves and reads them
    with 1 536 reader waves through the library's access patterns (a row written with one value in every column; a reader that finds two
    values has read it torn).  Bounded here: the time-sliced mode's pattern for rows of 8 / 16 columns -- a row is ONE request of a lane group
    both ways -- and the 2-column row of the ordered mode (one 8-byte granule) are never torn in ~10^9 reads; the other patterns (a row as
    several requests of one lane) do tear under this stress and are reported (DESIGN 9 prices what that means in a real batch)."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "ubench_torn_rows"
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", os.path.join(root, "tools", "ubench_torn_rows.hip"), "-o", str(exe)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-1000:])
    print(r.stdout)
    rows = {}
    for ln in r.stdout.splitlines():
        if ln.startswith("D="):
            d = int(ln[2:4])
            name = ln[6:50].strip()
            torn = int(ln.split("torn")[1].split("of")[0])
            reads = float(ln.split("of")[1].split("row reads")[0])
            rows[(d, name)] = (torn, reads)
    for d in (8, 16):
        torn, reads = rows[(d, "lane-group store / lane-group load (sliced)")]
        assert torn == 0 and reads > 1e8, (d, torn, reads)
    torn, reads = rows[(2, "8-B agent store / load, d = 2: one granule")]
    assert torn == 0 and reads > 1e8
