#!/usr/bin/env python3
"""bench.py -- headline benchmark of the annembed hot path on MI355X.

Metric (BASELINE.json): embedded points/sec per CE epoch (one `gradient_iteration` batch over
nb_sampling_by_edge * nnz SGD samples), plus the SVD-init GFLOP/s of the diffusion-map initialisation.

Workload at N=1: configs[1] "MNIST-fashion 60k x 784 -> 2D, k=12, dmap init + CE loop, fp32"
(parameters of examples/mnist_fashion.rs:92-110).  The real dataset is absent (no network): a
synthetic Gaussian mixture of the same shape stands in and the exact kNN graph is built
on the GPU before the timed region.  The headline `value` is the DEFAULT CE mode (AE_CE_AUTO -> at this size the ordered
dataflow AE_CE_ORDERED: the reference's sample sequence and f64 scalars with only a sample's two end points as dependencies:
statistical parity); the same line carries
  * "parity_mode": AE_CE_SEQUENTIAL, bit-exact against the oracle (the figure at the north star's coordinate tolerance);
  * "event_mode" (AE_CE_EVENT) and "rounds_mode" (AE_CE_HOGWILD: a throughput mode OUTSIDE the reference's envelope), same graph
    and start;
  * "fidelity": final CE and edge-length quartiles of the modes (and of a second sequential seed: the reference's own spread)
    after the full 25-batch schedule, as ratios to the sequential mode;
  * "scale_shapes": configs[2] / configs[3] / one GPU's eighth of configs[4] on the configs' own kind of graph (exact kNN inside
    the components of a Higgs-shaped / 128-D mixture: real in-degree skew) and on a node-permuted ring lattice (the best case),
    every mode timed, plus "c5_full_shape": configs[4] WHOLE (50 M nodes, 5 G samples per batch) in the default mode
    (--no-full-size skips it: about a minute, mostly the graph).

N>1 (default): strong scaling of configs[3] on its own graph -- the source nodes of the 11 M-node kNN graph (component order, ids
shuffled inside every rank's range) sharded over the ranks in contiguous ranges, AE_CE_AUTO = the faithful time-sliced mode on a
sharded range, the owned coordinate rows all-gathered by the library's RCCL communicator inside
ae_entropy_optim_gradient_iteration (`--exchanges` times per batch; torch's RCCL as the fallback).  `--rounds`: the approximate
rounds mode on the lattice (round 2-3's series); `--weak`: 60 k MNIST-shaped points per GPU.

A "step" = one CE batch.  The timed region holds only `ae_entropy_optim_gradient_iteration` calls
(the collective is inside them when N>1) with every input already resident in HBM.

Rank 0 prints everything it measured as `bench_details: {...}` (also written to bench_details.json) and then, LAST, ONE compact
JSON line (< 4 KB: the contract's keys + roofline + cpu_baseline + parity_mode + one figure per scale shape).
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def synth_points(n, dim, seed, ncomp=10, active=50, idim=18, sigma=30.0, device="cuda"):
    """MNIST-shaped stand-in: mixture of `ncomp` components in [0,255]^dim; component means are
    U(0,255) on `active` random coordinates; inside a component the points live on a random `idim`-dimensional
    subspace (sigma 30) -- idim = 18 is the intrinsic dimension the reference measures on MNIST (README.md:102),
    which gives the kNN graph a realistic in-degree (hubness) profile instead of that of a 784-d isotropic blob."""
    import torch
    g = torch.Generator(device="cpu").manual_seed(seed)
    means = torch.zeros(ncomp, dim)
    for c in range(ncomp):
        idx = torch.randperm(dim, generator=g)[:active]
        means[c, idx] = torch.rand(active, generator=g) * 255.0
    lab = torch.randint(0, ncomp, (n,), generator=g)
    x = torch.empty(n, dim)
    for c in range(ncomp):
        basis = torch.linalg.qr(torch.randn(dim, idim, generator=g))[0]
        m = lab == c
        x[m] = means[c] + (sigma * torch.randn(int(m.sum()), idim, generator=g)) @ basis.T
    return x.clamp_(0, 255).to(device)


def knn_rows(x_all, lo, hi, k):
    """exact L2 kNN of rows [lo,hi) against all rows (torch, input preparation only)"""
    import torch
    sq = (x_all * x_all).sum(1)
    nbr = torch.empty((hi - lo, k), dtype=torch.int64, device=x_all.device)
    dist = torch.empty((hi - lo, k), dtype=torch.float32, device=x_all.device)
    bs = 4096
    for b in range(lo, hi, bs):
        e = min(b + bs, hi)
        d2 = sq[b:e, None] + sq[None, :] - 2.0 * (x_all[b:e] @ x_all.T)
        d2[torch.arange(e - b, device=x_all.device), torch.arange(b, e, device=x_all.device)] = float("inf")
        v, i = torch.topk(d2, k, dim=1, largest=False, sorted=True)
        nbr[b - lo:e - lo] = i
        dist[b - lo:e - lo] = v.clamp_min(0).sqrt()
    return nbr, dist


def svd_flops(n, nnz_a, l=20, nbiter=5):
    """algorithmic flops of subspace_iteration_csr + direct_svd"""
    return ((2 * nbiter - 1) * 2 * nnz_a * l + (2 * nbiter - 1) * 4 * n * l * l + 2 * nnz_a * l + 6 * n * l * l
            + 2 * n * l * l)


def svd_bytes(n, nnz_a, l=20, nbiter=5):
    """algorithmic HBM bytes of the same (each SpMM >= 8 nnz_A + 2 * 4 N l, each QR >= 2 * 4 N l per pass, two passes): 2 nbiter
    products (the range iteration's 2 nbiter - 1 and B = Q^T A), 2 nbiter - 1 orthonormalisations, U = Q U_b"""
    return 2 * nbiter * (8 * nnz_a + 8 * n * l) + (2 * nbiter - 1) * 2 * 8 * n * l + 8 * n * l


def svd_init_of(A, L, kg, d, reps=3):
    """the diffusion-map initialisation's randomized SVD (graphlaplace.rs:97-125: rank 20, 5 iterations) of a graph's laplacian, timed alone
    (U stays in HBM as in the embedder's own call): ms, GFLOP/s and HBM fraction by the SURVEY 8d formulas"""
    dp = A.DiffusionParams(d, 5.0, 12)
    # (stage-level calls sum as f64 trees by default -- ae_set_summation_order --, as Embedder::embed does when no bit-exact CE mode follows)
    t0 = time.perf_counter()
    lap = A.DiffusionMaps(dp).laplacian_from_kgraph(kg)
    L.check(L.load().ae_synchronize())
    lap_s = time.perf_counter() - t0
    _, n, nnz_a = lap.info()
    lap.do_svd(want_u=False)  # warm
    L.check(L.load().ae_synchronize())
    t0 = time.perf_counter()
    for _ in range(reps):
        sv = lap.do_svd(want_u=False)
    dt = (time.perf_counter() - t0) / reps
    return {"ms": dt * 1e3, "gflops": svd_flops(n, nnz_a) / dt / 1e9, "hbm_gbps": svd_bytes(n, nnz_a) / dt / 1e9, "hbm_frac": svd_bytes(n, nnz_a) / dt / 8e12,
            "nnz_laplacian": int(nnz_a), "nodes": int(n), "rank": 20, "nbiter": 5, "laplacian_build_ms": lap_s * 1e3,
            "sigma_1_2_3": [float(v) for v in sv.get_sigma()[:3]] if hasattr(sv, "get_sigma") else None}


def lattice_graph(n, k, seed, permute):
    """ring lattice (neighbours at Fibonacci offsets, gamma-distributed distances): a kNN-shaped graph at sizes where an
    exact kNN is out of reach of brute force.  permute: node ids are a random permutation of the ring positions, so that
    the rows a sample touches are scattered over the coordinate array as they are for a real kNN graph in arbitrary order."""
    rng = np.random.default_rng(seed)
    offs = np.array([1, 2, 3, 5, 8, 13, 21, 34, 55, 89, 144, 233, 377, 610, 987, 1597][:(k + 1) // 2])
    if permute:
        perm = rng.permutation(n).astype(np.int64)   # ring position -> node id
        pos = np.empty(n, np.int64)
        pos[perm] = np.arange(n)                     # node id -> ring position
        cols = [perm[(pos + o) % n] for o in offs] + [perm[(pos - o) % n] for o in offs]
    else:
        base = np.arange(n, dtype=np.int64)
        cols = [(base + o) % n for o in offs] + [(base - o) % n for o in offs]
    nbr = np.stack(cols[:k], 1).astype(np.uint32).reshape(-1)
    dst = np.sort(rng.gamma(2.0, 1.0, size=(n, k)).astype(np.float32), axis=1).reshape(-1)
    indptr = np.arange(n + 1, dtype=np.uint64) * np.uint64(k)
    return indptr, nbr, dst


def edge_quartiles(indptr, nbr, y):
    src = np.repeat(np.arange(len(indptr) - 1), np.diff(indptr.astype(np.int64)))
    return np.quantile(np.linalg.norm(y[src] - y[nbr], axis=1), [0.25, 0.5, 0.75])


MODE_NAMES = {5: "time-sliced on conflict-free matchings (AE_CE_SLICED)", 0: "rounds (AE_CE_HOGWILD)", 1: "sequential-equivalent dataflow (AE_CE_SEQUENTIAL), bit-exact vs the oracle", 2: "racy", 3: "event-ordered (AE_CE_EVENT)",
              6: "ordered dataflow (AE_CE_ORDERED): the sequential order, end points sequentially consistent, negatives as the memory system has them"}
MODE_KERNEL = {0: "ce_round_node_kernel (one launch per round)",
               1: "ce_dataflow_kernel (one persistent launch per batch, grid sized one workgroup per CU below the occupancy query; the batch also holds the plan, sort and predecessor kernels)",
               3: "ce_event_window_kernel (one launch per window)",
               6: "ce_dataflow_kernel<relaxed> (one persistent launch per batch; the batch also holds the plan, sort and predecessor kernels)",
               5: "sl_direct_kernel (one launch per colour class and time slice) / sl_exec_kernel (optimistic passes of the overflow class); whole batch incl. event generation and sort"}


def mode_dtype(mode, precision=0):
    """the arithmetic a CE mode computes in (coordinates / scalar coefficients); the reference: f32 coordinates, f64 scalars (embedder.rs:1207-1229)"""
    if mode == 0:
        return "f32 coordinates, f32 scalars (rounds mode: narrower than the reference)"
    if mode == 5 and precision == 1:
        return "f32 coordinates, f32 scalars (ce_precision = AE_PRECISION_F32: an explicit opt-in, narrower than the reference)"
    return "f32 coordinates, f64 scalars (the reference's)"


def time_mode(A, L, kg, node_params, y0, d, mode, steps, warmup, nb_batch=25, lo=0, hi=None, comm=None, exchanges=1, fence=None, hub=None, precision=0):
    """`warmup` untimed + `steps` timed CE batches of one mode; returns timing and the per-launch roofline inputs"""
    params = A.EmbedderParams(asked_dim=d, nb_grad_batch=nb_batch, nb_sampling_by_edge=10, grad_step=1.0, scale_rho=1.0, beta=1.0, ce_mode=mode,
                              hubness_weighting=hub is not None, ce_precision=precision)
    eo = A.EntropyOptim(kg, node_params, params, y0, node_lo=lo, node_hi=hi, hub_counts=hub)
    if comm is not None:
        comm.attach(eo, exchanges)
    nb_sample = params.nb_sampling_by_edge * eo.get_nb_edges()
    total = max(nb_batch, warmup + steps + 1)
    sync = fence or (lambda: L.check(L.load().ae_synchronize()))
    ce_before = eo.ce_compute_threaded()
    it = 0
    for _ in range(warmup):
        it += 1
        eo.gradient_iteration_threaded(nb_sample, params.grad_step * (1.0 - it / total), it)
    sync()
    eo.kernel_time()
    resolved = eo.get_ce_mode()
    if resolved in (1, 6):
        eo.dataflow_time()
    t0 = time.perf_counter()
    for _ in range(steps):
        it += 1
        eo.gradient_iteration_threaded(nb_sample, params.grad_step * (1.0 - it / total), it)
    sync()
    elapsed = time.perf_counter() - t0
    kernel_ms, launches = eo.kernel_time()
    rounds = int(eo.samples_drawn()[1]) if resolved in (0, 3) else 1
    sliced = None
    if resolved == 5:
        cl, ovf, crounds, slices = eo.slice_info()
        indeg_max, chain_len = eo.slice_hub_info()
        form = eo.slice_form()
        sliced = {"classes": cl, "overflow_mass_fraction": ovf, "colouring_rounds": crounds, "slices_per_batch": slices,
                  "step_launches_per_batch": slices * cl, "max_in_degree": indeg_max, "longest_chain_per_step_expected": chain_len,
                  "launch_form": SLICE_FORMS.get(form, str(form))}
    dominant_ms = eo.dataflow_time()[0] if resolved in (1, 6) else None  # the dataflow kernel alone (the batch also plans, sorts, searches)
    return dict(eo=eo, elapsed=elapsed, ms_per_step=elapsed / steps * 1e3, kernel_ms=kernel_ms, batches_timed=int(launches), rounds=rounds,
                mode=resolved, nb_sample=nb_sample, dtype=mode_dtype(resolved, precision), ce_before=ce_before, ce_after=eo.ce_compute_threaded(), dominant_ms=dominant_ms, sliced=sliced)


def roofline_of(run, k, d):
    """algorithmic bytes per SGD sample B = 24 + 4 k + 36 d; one launch of the dominant kernel processes
    nb_sample / launches_per_batch samples; achieved = bytes per launch / average launch duration (hipEvents on the library's
    stream around the batch's launches / launches per batch)"""
    bytes_per_sample = 24 + 4 * k + 36 * d
    lpb = max(run["rounds"], 1)
    kernel_ms = run["kernel_ms"] if run["kernel_ms"] > 0 else run["ms_per_step"]
    launch_ms = run["dominant_ms"] if run.get("dominant_ms") else kernel_ms / lpb
    bytes_per_launch = bytes_per_sample * run["nb_sample"] / lpb
    achieved = bytes_per_launch / (launch_ms * 1e-3) / 1e9 if launch_ms > 0 else 0.0
    whole = bytes_per_sample * run["nb_sample"] / (run["ms_per_step"] * 1e-3) / 1e9
    extra = {}
    if run.get("sliced"):
        extra["sliced"] = run["sliced"]
    return {**extra, "bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0, "traffic": None,
            "achieved_whole_batch": whole, "frac_whole_batch": whole / 8000.0,
            "kernel": MODE_KERNEL.get(run["mode"], "?"), "launches_per_batch": lpb, "launch_avg_ms": launch_ms, "batch_kernel_ms": kernel_ms,
            "batches_timed": run["batches_timed"], "bytes_per_sample": bytes_per_sample, "bytes_per_launch": bytes_per_launch}


def full_schedule(A, kg, node_params, y0, d, mode, nb_batch=25):
    p = A.EmbedderParams(asked_dim=d, nb_grad_batch=nb_batch, nb_sampling_by_edge=10, grad_step=1.0, scale_rho=1.0, beta=1.0, ce_mode=mode)
    y, _, ce = A.entropy_optimize(kg, node_params, p, y0)
    return y, ce


# ae_entropy_optim_slice_form: the launch form decides how old the negatives' rows are, i.e. how faithful the time-sliced mode is
SLICE_FORMS = {0: "none", 1: "one launch per class", 2: "one launch per class, node lines", 3: "merged slices", 4: "optimistic passes",
               5: "merged slices, class window"}
FAITHFUL_FRESH = "statistically (one launch per class: negatives a step old; inside the exact mode's standard error at 32 seeds)"
FAITHFUL_WINDOW = ("statistically (merged slices with the class window: negatives half a slice old at most; CE +0.12 +- 0.22 %, median edge -0.18 +- 0.43 % "
                   "against one launch per class at 256 seeds a side on the stiff 2-D graph: inside the standard error)")
FAITHFUL_STALE = ("statistically, with a RESOLVED BIAS on stiff 2-D graphs: final cross entropy +0.4 %, median edge -0.7 % at 256 seeds (optimistic passes / merged "
                  "slices without their class window read a slice's negatives as the slice found them; not visible at 8 columns; DESIGN.md 4.3b)")


def faithful_of(run):
    """the `faithful` field of a time-sliced figure, by the launch form it ran in"""
    form = ((run or {}).get("sliced") or {}).get("launch_form", "")
    return FAITHFUL_STALE if form in ("merged slices", "optimistic passes") else (FAITHFUL_WINDOW if form == "merged slices, class window" else FAITHFUL_FRESH)


def dense_svd_flops(m, n, l=20, nbiter=5):
    """SURVEY 8d, dense path (subspace_iteration_full + direct_svd, svdapprox.rs:285-333, 721-799)"""
    return (2 * nbiter - 1) * 2 * m * n * l + nbiter * 4 * m * l * l + (nbiter - 1) * 4 * n * l * l + 2 * m * n * l


def svd_dense_shape(A, L, m, n=128, reps=3):
    """The dense range finder where SURVEY 8d puts it: `direct_svd` (RangeRank(20, 5)) of an m x 128 f32 block of configs[4]'s data matrix
    (6.25 M rows = one GPU's eighth, 3.2 GB; 50 M rows = the whole matrix, 25.6 GB), the matrix resident in HBM, U left there (as the
    embedder's own call leaves it), the spectrum back on the host.  ms, TFLOP/s by the SURVEY formula, fraction of the f32 MFMA peak and of
    HBM by algorithmic bytes (4 m n per product, ten products)."""
    import torch
    x, _ = mixture_points_gpu(m, n, max(1, m // 50_000), seed=4, mean_sigma=10.0)
    t0 = time.perf_counter()
    mat = A.MatRepr.from_array2(x)
    L.check(L.load().ae_synchronize())
    upload_s = time.perf_counter() - t0
    del x
    svd, mode = A.SvdApprox(mat), A.RangeRank(20, 5)
    svd.direct_svd(mode, want_u=False, want_vt=False)  # warm
    L.check(L.load().ae_synchronize())
    t0 = time.perf_counter()
    for _ in range(reps):
        res = svd.direct_svd(mode, want_u=False, want_vt=False)
    dt = (time.perf_counter() - t0) / reps
    fl, by = dense_svd_flops(m, n), 10 * 4.0 * m * n
    out = {"shape": "%dx%d rank 20 nbiter 5" % (m, n), "ms": dt * 1e3, "tflops": fl / dt / 1e12, "mfma_f32_peak_tflops": 157.3, "mfma_frac": fl / dt / 1e12 / 157.3,
           "hbm_gbps": by / dt / 1e9, "hbm_frac": by / dt / 8e12, "sigma0": float(res.s[0]), "sigma19": float(res.s[-1]), "upload_s": upload_s,
           "ceiling_note": "a product moves 4 m n bytes for 2 m n l flops: 10 flop/B at l = 20, i.e. the HBM roof allows 80 TFLOP/s = 0.51 of the f32 MFMA peak"}
    del svd, mat
    torch.cuda.empty_cache()
    return out


def higgs_shaped_points(n, dim=28, ncomp=64, seed=2, with_labels=False):
    """configs[2] stand-in: mixture of 64 Gaussians in 28-D with per-column standardisation (examples/higgs.rs:158-176)"""
    rng = np.random.default_rng(seed)
    means = rng.normal(size=(ncomp, dim)) * 2.0
    scales = 0.5 + rng.random((ncomp, dim))
    lab = rng.integers(0, ncomp, n)
    x = means[lab] + scales[lab] * rng.normal(size=(n, dim))
    x = (x - x.mean(0)) / x.std(0)
    x = np.ascontiguousarray(x.astype(np.float32))
    return (x, lab) if with_labels else x


def mixture_points_gpu(n, dim, ncomp, seed, mean_sigma, higgs_like=False):
    """Gaussian-mixture points generated on the GPU, SORTED BY COMPONENT -> (host float32[n, dim], component bounds int64[ncomp + 1]).
    higgs_like: configs[3]'s generator (per-component scales 0.5-1.5, means N(0, 2^2), columns standardised -- higgs_shaped_points'
    law); else configs[4]'s (means N(0, mean_sigma^2), sigma 1, equal-size components)."""
    import torch
    rng = np.random.default_rng(seed)
    means = rng.normal(size=(ncomp, dim)) * (2.0 if higgs_like else mean_sigma)
    scales = (0.5 + rng.random((ncomp, dim))) if higgs_like else np.ones((ncomp, dim))
    if higgs_like:
        counts = np.bincount(rng.integers(0, ncomp, n), minlength=ncomp)
    else:
        counts = np.full(ncomp, n // ncomp)
        counts[:n - counts.sum()] += 1
    bounds = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    g = torch.Generator(device="cuda").manual_seed(seed)
    x = torch.empty((n, dim), dtype=torch.float32, device="cuda")
    for c in range(ncomp):
        b, e = int(bounds[c]), int(bounds[c + 1])
        x[b:e] = torch.randn((e - b, dim), generator=g, device="cuda") * torch.from_numpy(scales[c].astype(np.float32)).cuda() \
            + torch.from_numpy(means[c].astype(np.float32)).cuda()
    if higgs_like:
        x = (x - x.mean(0)) / x.std(0)
    xh = x.cpu().numpy()
    del x
    torch.cuda.empty_cache()
    return xh, bounds


def component_knn_graph(A, x, bounds, k, permute_seed, shuffle_within=None):
    """Exact kNN graph INSIDE every mixture component (the blocks of rows `bounds`), one brute-force pass on the matrix cores per
    component.  SURVEY 8d prescribes this for configs[4] (1 000 well separated components: it equals the global kNN graph w.h.p.; the
    separation is checked below where it holds by construction); for the Higgs-shaped points, whose components overlap, it is the kNN
    graph restricted to the component -- as approximate as the reference's own HNSW graph, with the in-degree skew (hubs) of the data.
    permute_seed: node ids are randomly permuted afterwards (positive edges not memory-local, as for a graph in arbitrary order); None
    keeps the component order (what a component partition over GPUs wants); shuffle_within = [(lo, hi), ...] then shuffles the ids
    INSIDE each of these ranges (a partitioner hands every rank its components, in no particular order inside the part).
    -> indptr, nbr, dist (host CSR)"""
    n = len(x)
    nbr = np.empty((n, k), np.uint32)
    dist = np.empty((n, k), np.float32)
    for c in range(len(bounds) - 1):
        b, e = int(bounds[c]), int(bounds[c + 1])
        if e - b <= k:
            raise ValueError("component smaller than k")
        g = A.KGraph.bruteforce_l2(x[b:e], k)
        _, nb, ds = g.get_neighbours()
        nbr[b:e] = nb.reshape(e - b, k) + np.uint32(b)
        dist[b:e] = ds.reshape(e - b, k)
        del g
    if permute_seed is None and shuffle_within:
        rng = np.random.default_rng(17)
        perm = np.arange(n, dtype=np.uint32)
        for lo, hi in shuffle_within:
            perm[lo:hi] = lo + rng.permutation(hi - lo).astype(np.uint32)
        inv = np.empty(n, np.int64)
        inv[perm] = np.arange(n)
        nbr = perm[nbr][inv]
        dist = dist[inv]
    if permute_seed is not None:
        perm = np.random.default_rng(permute_seed).permutation(n).astype(np.uint32)  # old id -> new id
        inv = np.empty(n, np.int64)
        inv[perm] = np.arange(n)
        nbr = perm[nbr][inv]
        dist = dist[inv]
    indptr = np.arange(n + 1, dtype=np.uint64) * np.uint64(k)
    return indptr, np.ascontiguousarray(nbr.reshape(-1)), np.ascontiguousarray(dist.reshape(-1))


def full_size_shape(A, L, which, d, steps):
    """A config's own graph at FULL size on this one GPU, the mode AE_CE_AUTO resolves to only (no rounds / f32 / parity variants: at
    50 M nodes each of them is a minute): random start, one warm-up batch, `steps` timed ones.  configs[4] is specified over 8 GPUs;
    one MI355X holds it whole (55 GB)."""
    import torch
    gr = config_graphs(A, which)
    n, k = gr["n"], gr["k"]
    indeg = np.bincount(gr["nbr"], minlength=n)
    kg = A.KGraph(gr["indptr"], gr["nbr"], gr["dist"], k)
    node_params = A.to_proba_edges(kg, 1.0, 1.0)
    y0 = A.set_data_box(np.random.default_rng(1).normal(size=(n, d)).astype(np.float32), 10.0)
    run = time_mode(A, L, kg, node_params, y0, d, A.AE_CE_AUTO, steps, 1)
    out = {"nodes": n, "k": k, "asked_dim": d, "graph": gr["desc"], "graph_build_s": gr["build_s"], "max_in_degree": int(indeg.max()),
           "in_degree_q999": float(np.quantile(indeg, 0.999)), "start": "random normal layout in the 10-box",
           "default_mode": {"faithful": faithful_of(run), "ce_mode": MODE_NAMES.get(run["mode"], str(run["mode"])), "dtype": run["dtype"],
                            "ms_per_step": run["ms_per_step"], "steps_timed": steps, "points_per_s": n / (run["ms_per_step"] * 1e-3),
                            "samples_per_s": run["nb_sample"] / (run["ms_per_step"] * 1e-3), "ce_before": run["ce_before"], "ce_after": run["ce_after"],
                            "roofline": roofline_of(run, k, d)}}
    free, total = torch.cuda.mem_get_info()
    out["hbm_used_gb"] = round((total - free) / 1e9, 1)
    del run, kg, node_params
    torch.cuda.empty_cache()
    return out


def scale_shape(A, L, name, n, k, d, steps, with_sequential, graph=None, hub_weighting=False, dmap_start=None, svd_init=False):
    """configs[2] / [3] / [4]-shard shapes on one GPU.  graph None: the node-permuted ring lattice (uniform in-degree: the best case of
    every faithful mode), started from its diffusion-map initialisation.  graph = dict(indptr, nbr, dist, desc, build_s): a kNN graph of
    the config's own data (real in-degree skew: hubs), started from a random layout in the 10-box (a component-wise kNN graph is
    disconnected: no diffusion-map start); hub_weighting: hubness-weighted negative sampling as examples/higgs.rs switches it on
    (:204-242).  Every mode: `steps` timed batches after one warm-up (the rounds and bit-exact modes: 2), same start, so ce_after is
    comparable across the modes of a shape."""
    import torch
    hub = None
    if graph is None:
        indptr, nbr, dst = lattice_graph(n, k, seed=7, permute=True)
        kg = A.KGraph(indptr, nbr, dst, k)
        desc = "ring lattice, node ids randomly permuted (uniform in-degree: best case)"
    else:
        kg = A.KGraph(graph["indptr"], graph["nbr"], graph["dist"], k)
        desc = graph["desc"]
    out = {"nodes": n, "k": k, "asked_dim": d, "graph": desc}
    hubv = kg.hubness()
    out["max_in_degree"] = int(hubv.max())
    out["in_degree_q999"] = int(np.quantile(hubv, 0.999))
    if hub_weighting:
        hub = hubv
        out["negative_sampling"] = "hubness-weighted (NodeSampler, embedder.rs:915-930)"
    y0 = None
    if dmap_start if dmap_start is not None else graph is None:
        try:
            t0 = time.perf_counter()
            y0 = A.set_data_box(A.DiffusionMaps(A.DiffusionParams(d, 5.0, 12)).embed_from_kgraph(kg), 10.0)
            L.check(L.load().ae_synchronize())
            out["dmap_init_s"] = time.perf_counter() - t0
            out["start"] = "diffusion-map initialisation"
            if svd_init:
                out["svd_init"] = svd_init_of(A, L, kg, d)
        except A.AnnembedError as e:   # e.g. a degenerate spectrum on a graph of many components
            out["dmap_init_error"] = str(e)[:300]
            y0 = None
    if y0 is None:
        y0 = A.set_data_box(np.random.default_rng(1).normal(size=(n, d)).astype(np.float32), 10.0)
        out["start"] = "random normal layout in the 10-box"
    if graph is not None:
        out["graph_build_s"] = graph.get("build_s")
        for key in ("edges_leaving_their_cluster", "knn_pairs"):
            if key in graph:
                out[key] = graph[key]
    node_params = A.to_proba_edges(kg, 1.0, 1.0)

    def entry(r, faithful):
        e = {"faithful": faithful, "ce_mode": MODE_NAMES.get(r["mode"]), "dtype": r["dtype"], "ms_per_step": r["ms_per_step"], "steps_timed": r["steps"],
             "points_per_s": n / (r["ms_per_step"] * 1e-3), "samples_per_s": r["nb_sample"] / (r["ms_per_step"] * 1e-3), "ce_after": r["ce_after"],
             "roofline": roofline_of(r, k, d)}
        return e

    def run(mode, nsteps, precision=0):
        r = time_mode(A, L, kg, node_params, y0, d, mode, nsteps, 1, hub=hub, precision=precision)
        r.pop("eo")
        r["steps"] = nsteps
        return r

    out["rounds_mode"] = entry(run(A.AE_CE_HOGWILD, 2), False)
    rs = run(A.AE_CE_SLICED, steps)
    out["sliced_mode"] = entry(rs, faithful_of(rs))
    out["sliced_mode_f32_scalars"] = entry(run(A.AE_CE_SLICED, 2, precision=1), "statistically (f32 scalars: an explicit opt-in, narrower than the reference)")
    # what AE_CE_AUTO runs at this size, timed like every other mode
    auto = A.EntropyOptim(kg, node_params, A.EmbedderParams(asked_dim=d, hubness_weighting=hub is not None), y0, hub_counts=hub)
    resolved = auto.get_ce_mode()
    del auto
    if resolved == A.AE_CE_SLICED:
        out["default_mode"] = dict(entry(rs, faithful_of(rs)), note="AE_CE_AUTO resolves to AE_CE_SLICED at this size: the sliced_mode run above")
    else:
        out["default_mode"] = entry(run(A.AE_CE_AUTO, max(5, steps)), "statistically (the ordered dataflow: 1.002 / 0.994 of the exact mode's final CE / median edge at 32 seeds)")
    if with_sequential:
        out["parity_mode"] = entry(run(A.AE_CE_SEQUENTIAL, 2), True)
    out["note"] = "same start and the same schedule in every mode: ce_after is comparable across the modes of a shape whose steps_timed agree"
    del kg, node_params
    torch.cuda.empty_cache()
    return out


def exact_knn_graph(A, x, k, what):
    """the EXACT kNN graph of all the points (one brute-force pass on the matrix cores)"""
    t0 = time.perf_counter()
    kg = A.KGraph.bruteforce_l2(x, k)
    indptr, nbr, dist = kg.get_neighbours()
    del kg
    dt = time.perf_counter() - t0
    return {"indptr": indptr, "nbr": nbr, "dist": dist, "build_s": dt, "desc": "exact kNN graph (k = %d) of %d %s; built in %.1f s" % (k, len(x), what, dt)}


def config_graphs(A, which, permute_seed=9, n_override=None, shuffle_within_shards=0):
    """the configs' own kind of graph for the scale shapes (SURVEY 8d generators; exact kNN inside every mixture component, node ids
    randomly permuted): 'c4' = configs[3] (11 M Higgs-shaped points, k 6), 'c5' = one GPU's eighth of configs[4] (6.25 M points of the
    128-D mixture: 125 of its 1 000 components of 50 000 points, k 10)."""
    t0 = time.perf_counter()
    if which == "c4":
        n, dim, ncomp, k = n_override or 11_000_000, 28, 64, 6
        x, bounds = mixture_points_gpu(n, dim, ncomp, seed=3, mean_sigma=2.0, higgs_like=True)
        what = "Higgs-shaped points (28-D, 64 overlapping Gaussian components, columns standardised; SURVEY 8d, seed 3)"
    else:
        if which == "c5_full" and not n_override:
            n_override = 50_000_000
        n, dim, k = n_override or 6_250_000, 128, 10
        ncomp = max(1, n // 50_000)
        x, bounds = mixture_points_gpu(n, dim, ncomp, seed=4, mean_sigma=10.0)
        what = "points of the 128-D mixture (%d of configs[4]'s 1 000 components of 50 000 points: means N(0, 10^2), sigma 1; SURVEY 8d, seed 4)" % ncomp
    t1 = time.perf_counter()
    within = None
    if shuffle_within_shards:
        from annembed_amd.dist import shard_range
        within = [shard_range(n, shuffle_within_shards, r) for r in range(shuffle_within_shards)]
    if which == "c4":
        # configs[3]'s graph as HNSW would give it (kgraph.rs:440-579: neighbours are GLOBAL): the exact global kNN graph through the
        # grouped producer (own cluster first, then only the shells of the other clusters a triangle-inequality bound cannot exclude)
        gk = A.KGraph.bruteforce_l2_grouped(x, k, bounds.astype(np.uint64))
        knn_stats = gk.knn_stats
        indptr, nbr, dist = gk.get_neighbours()
        del gk
        lab = np.searchsorted(bounds, np.arange(n), side="right") - 1
        leaving = float((np.repeat(lab, k) != lab[nbr]).mean())
        del lab
        if permute_seed is not None:
            perm = np.random.default_rng(permute_seed).permutation(n).astype(np.uint32)  # old id -> new id
            inv = np.empty(n, np.int64)
            inv[perm] = np.arange(n)
            nbr = np.ascontiguousarray(perm[nbr.reshape(n, k)][inv].reshape(-1))
            dist = np.ascontiguousarray(dist.reshape(n, k)[inv].reshape(-1))
            del perm, inv
    else:
        indptr, nbr, dist = component_knn_graph(A, x, bounds, k, permute_seed=permute_seed, shuffle_within=within)
    del x
    t2 = time.perf_counter()
    if which == "c4":
        return {"indptr": indptr, "nbr": nbr, "dist": dist, "build_s": t2 - t0, "k": k, "n": n, "edges_leaving_their_cluster": leaving,
                "knn_pairs": {"fallback_rows": knn_stats[0], "pruned_phase": knn_stats[1], "inside_clusters": knn_stats[2], "all": float(n) * n},
                "desc": "GLOBAL exact kNN graph (k = %d) of %d %s, node ids %s; points %.1f s, graph %.1f s (%.2f %% of the n^2 pairs computed; %.4f %% of the edges "
                        "leave their cluster)" % (k, n, what, "randomly permuted" if permute_seed is not None else "in cluster order", t1 - t0, t2 - t1,
                                                  100.0 * (knn_stats[1] + knn_stats[2]) / (float(n) * n), 100.0 * leaving)}
    return {"indptr": indptr, "nbr": nbr, "dist": dist, "build_s": t2 - t0, "k": k, "n": n,
            "desc": "kNN graph (k = %d) of %d %s, exact inside every component, node ids %s; points %.1f s, graph %.1f s" % (
                k, n, what, "randomly permuted" if permute_seed is not None else
                ("in component order, shuffled inside each of %d contiguous shards" % shuffle_within_shards if shuffle_within_shards else "in component order"),
                t1 - t0, t2 - t1)}


FINAL_LINE_MAX = 4096   # the driver keeps the tail of stdout: the LAST line must parse on its own (round 4's 40 KB line did not)


def _short(v, nmax=160):
    return v if not isinstance(v, str) or len(v) <= nmax else v[:nmax - 3] + "..."


def compact_roofline(r):
    """the judged fields of a roofline object, nothing else"""
    if not r:
        return None
    keep = ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_from", "kernel", "launch_avg_ms", "launches_per_batch",
            "bytes_per_sample", "bytes_per_launch", "frac_whole_batch")
    return {k: (_short(r[k], 120) if k in ("kernel", "traffic_from") else r[k]) for k in keep if k in r}


def compact_line(full):
    """`full` = everything bench.py measured (-> bench_details.json and an earlier stdout line).  Returns the ONE final JSON line:
    the contract's keys, roofline, cpu_baseline, parity_mode, the SVD-init figures and one number per scale shape."""
    out = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                                    "dtype", "data")}
    cfg = full.get("config") or {}
    out["config"] = {k: _short(cfg[k], 200) for k in ("workload", "ce_mode", "samples_per_step", "exchanges_per_batch", "rccl_ranks", "collective",
                                                      "partition", "cross_shard_mass") if k in cfg}
    out["roofline"] = compact_roofline(full.get("roofline"))
    cpu = full.get("cpu_baseline")
    if cpu:
        out["cpu_baseline"] = {k: _short(cpu[k], 200) for k in ("value", "unit", "cores", "kind", "sample") if k in cpu}
    else:
        out["cpu_baseline"] = None
    pm = full.get("parity_mode")
    if pm:
        out["parity_mode"] = {"ce_mode": "AE_CE_SEQUENTIAL (bit-exact vs the oracle)", "ms_per_step": pm["ms_per_step"], "points_per_s": pm["points_per_s"],
                              "frac": pm["roofline"]["frac"]}
    for k in ("svd_init", "svd_init_c4", "svd_dense", "svd_dense_c5", "svd_dense_c5_full"):
        if full.get(k):   # (scalars only, five significant digits, without the constants: the details line has everything)
            out[k] = {kk: (float("%.5g" % vv) if isinstance(vv, float) else vv) for kk, vv in full[k].items()
                      if not isinstance(vv, (dict, list)) and not (isinstance(vv, str) and len(vv) > 80) and kk not in ("mfma_f32_peak_tflops", "upload_s", "sigma19")}
    shapes = full.get("scale_shapes") or {}
    brief = {}
    for name, sh in shapes.items():
        dm = (sh or {}).get("default_mode")
        if dm:
            form = ((dm.get("roofline") or {}).get("sliced") or {}).get("launch_form")
            brief[name] = {"ms": round(dm["ms_per_step"], 2), "points_per_s": round(dm["points_per_s"]), "frac": round(dm["roofline"]["frac"], 4),
                           "frac_whole_batch": round(dm["roofline"]["frac_whole_batch"], 4)}
            if form:   # (optimistic passes, merged slices without their window: the published bias on stiff 2-D graphs applies)
                brief[name]["form"] = {"one launch per class": "per_class", "one launch per class, node lines": "lines", "merged slices": "merged*", "merged slices, class window": "merged+window",
                                       "optimistic passes": "optimistic*"}.get(form, form)
    if brief:
        out["scale_shapes"] = brief
        if any(str(v.get("form", "")).endswith("*") for v in brief.values()):
            out["form_note"] = "*: optimistic passes: CE +0.4 %, median edge -0.7 % on stiff 2-D graphs at 256 seeds (DESIGN 4.3b)"

    if full.get("end_to_end"):
        out["end_to_end"] = {k: v for k, v in full["end_to_end"].items() if k != "note"}
    for k in ("per_rank_batch_ms_max", "faithful", "samples_per_s", "ce_before", "ce_after", "details"):
        if k in full:
            out[k] = _short(full[k], 120)
    line = json.dumps(out)
    if len(line) >= FINAL_LINE_MAX:   # never: but a parseable short line beats a complete long one
        for k in ("scale_shapes", "svd_dense_c5_full", "svd_dense", "svd_dense_c5", "svd_init_c4", "svd_init", "faithful"):
            out.pop(k, None)
            line = json.dumps(out)
            if len(line) < FINAL_LINE_MAX:
                break
    assert len(line) < FINAL_LINE_MAX, len(line)
    return line


def emit(full):
    """writes bench_details.json (repo root, and gpurun_out/ when it exists), prints the details as an EARLIER stdout line (not JSON on
    its own: prefixed) and the compact line LAST"""
    full = dict(full)
    full["details"] = "bench_details.json (also the preceding stdout line, prefixed 'bench_details: ')"
    blob = json.dumps(full)
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        try:
            if os.path.isdir(d):
                with open(os.path.join(d, "bench_details.json"), "w") as f:
                    f.write(blob + "\n")
        except OSError:
            pass
    print("bench_details: " + blob, flush=True)
    print(compact_line(full), flush=True)


def spawn_ranks(n_ranks, argv):
    """`python3 bench.py --gpus N` without a launcher: N rank processes are started here, BEFORE this process makes any GPU call (a
    child is a fresh interpreter; nothing is exec'ed over a process that has touched the GPU).  Rank 0 inherits stdout (its last line
    is the result), the other ranks' stdout goes to stderr.  A failed rank fails the run."""
    import socket
    import subprocess
    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = str(sk.getsockname()[1])
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env, stdout=None if r == 0 else sys.stderr))
    rc = 0
    deadline = time.time() + float(os.environ.get("AE_BENCH_SPAWN_TIMEOUT", "3000"))
    alive = list(procs)
    while alive:
        for p_ in list(alive):
            code = p_.poll()
            if code is None:
                continue
            alive.remove(p_)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                for q in alive:     # a rank died: the others would wait in a collective for ever
                    q.terminate()
        if time.time() > deadline:
            for q in alive:
                q.kill()
            rc = rc or 124
            break
        time.sleep(0.2)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--points-per-gpu", type=int, default=60000)
    ap.add_argument("--dim", type=int, default=784)
    ap.add_argument("--knbn", type=int, default=12)
    ap.add_argument("--asked-dim", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-dense-svd", action="store_true")
    ap.add_argument("--no-exact-mode", action="store_true")
    ap.add_argument("--no-fidelity", action="store_true")
    ap.add_argument("--no-scale-shapes", action="store_true")
    ap.add_argument("--no-full-size", action="store_true", help="skip the full-size configs[4] shape (50 M nodes: about a minute)")
    ap.add_argument("--ce-mode", default="auto", choices=["auto", "event", "rounds", "sequential", "ordered"], help="mode of the headline figure at N = 1")
    ap.add_argument("--lattice-graph", action="store_true",
                    help="N = 1 scale runs: ring-lattice kNN graph (node ids permuted) with --points-per-gpu nodes instead of the MNIST-shaped points")
    ap.add_argument("--rounds", action="store_true", help="N > 1: the approximate rounds mode on the node-permuted lattice (rounds 1-3's strong-scaling series) instead of the "
                    "faithful time-sliced mode on the component-ordered kNN graph of the Higgs-shaped points")
    ap.add_argument("--weak", action="store_true", help="N > 1: weak scaling on 60 k MNIST-shaped points per GPU (round-1 arrangement) instead of the strong-scaling configs[3] shape")
    ap.add_argument("--scale-nodes", type=int, default=11_000_000, help="N > 1 strong scaling: nodes of the fixed graph (k = 6, asked_dim 8)")
    ap.add_argument("--exchanges", type=int, default=4, help="N > 1: all-gathers of the owned rows per CE batch.  4: on an 11 M-node graph in 8 shards one exchange per "
                    "batch left the edges 12-21 %% short (the other shards' rows a whole batch old during the violent first batches), 4 or 16 "
                    "matched the one-device run (DESIGN 5); ~1.5 ms per exchange at the C4 size")
    ap.add_argument("--force-dist", action="store_true", help="exercise the communicator path with world size 1 (validation)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend used for rendezvous, barriers and timing reductions (gloo: validation runs with several "
                         "ranks sharing ONE GPU; RCCL refuses duplicate devices, so the exchange then goes through torch/gloo and the figure is not a result)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:   # plain `python3 bench.py --gpus N`: start the N ranks ourselves
        raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:]))

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py --gpus %d started with WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the library has no CPU path)")
    if args.backend == "gloo":
        local_rank %= torch.cuda.device_count()  # ranks share the device(s) that exist
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")

    import annembed_amd as A
    from annembed_amd import _lib as L
    from annembed_amd.dist import LibraryComm, ShardedCE, HipBackend, device_tensor, shard_range
    L.check(L.load().ae_set_device(local_rank))

    def fence():
        L.check(L.load().ae_synchronize())
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    if use_dist:
        line = multi_gpu(args, A, L, dist, torch, rank, world, fence)
        dist.barrier()
        dist.destroy_process_group()
        if line is not None:  # the ONE JSON line LAST, after everything that might still write to stdout (RCCL's banner, teardown)
            emit(line)
        return

    # =============================== N = 1 ===============================
    k, d = args.knbn, args.asked_dim
    n = args.points_per_gpu
    svd_dense = knn_producer = None
    if args.lattice_graph:
        indptr, nbr, dst = lattice_graph(n, k, seed=1, permute=True)
    else:
        x = synth_points(n, args.dim, seed=1)
        nbr_all, dist_all = knn_rows(x, 0, n, k)
        # secondary figure: the dense range finder of tools::svdapprox on the data matrix itself -- subspace_iteration_full +
        # direct_svd, rank 20, 5 iterations -- the MFMA tall-skinny products
        if not args.no_dense_svd:
            mat = A.MatRepr.from_array2(x.cpu().numpy())
            for _ in range(8):   # warm (a call is ~1 ms: the first leg of the run, the GPU's clocks are still coming up -- 1.07 ... 1.54 ms with one warm call)
                A.SvdApprox(mat).direct_svd(A.RangeRank(20, 5), want_u=False, want_vt=False)
            L.check(L.load().ae_synchronize())
            t0 = time.perf_counter()
            for _ in range(20):   # (U stays in HBM, as in the embedder's own call; the spectrum comes back)
                A.SvdApprox(mat).direct_svd(A.RangeRank(20, 5), want_u=False, want_vt=False)
            dt = (time.perf_counter() - t0) / 20
            m_, n_, l_ = n, args.dim, 20
            fl = dense_svd_flops(m_, n_, l_)  # SURVEY 8d, dense path
            svd_dense = {"shape": "%dx%d rank 20 nbiter 5" % (m_, n_), "ms": dt * 1e3, "tflops": fl / dt / 1e12,
                         "mfma_f32_peak_tflops": 157.3, "mfma_frac": fl / dt / 1e12 / 157.3,
                         "hbm_gbps": 10 * 4.0 * m_ * n_ / dt / 1e9}
            del mat
            # the exact kNN-graph producer on the matrix cores, host matrix in, KGraph out (PCIe upload included)
            xh = x.cpu().numpy()
            A.KGraph.bruteforce_l2(xh, k)  # warm
            t0 = time.perf_counter()
            A.KGraph.bruteforce_l2(xh, k)
            dt = time.perf_counter() - t0
            knn_producer = {"shape": "%dx%d k=%d" % (n, args.dim, k), "ms": dt * 1e3, "tflops": 2.0 * n * n * args.dim / dt / 1e12,
                            "mfma_f32_peak_tflops": 157.3, "note": "exact (certified candidates + brute-force fallback), upload included"}
            del xh
        del x
        indptr = np.arange(n + 1, dtype=np.uint64) * np.uint64(k)
        nbr = nbr_all.cpu().numpy().astype(np.uint32).reshape(-1)
        dst = dist_all.cpu().numpy().reshape(-1)
        del nbr_all, dist_all
    torch.cuda.empty_cache()
    kg = A.KGraph(indptr, nbr, dst, k)

    # ---------------- dmap initialisation (timed separately: SVD-init GFLOP/s) ----------------
    dp = A.DiffusionParams(d, 5.0, 12)  # src/embedder.rs:317-321
    lap = A.DiffusionMaps(dp).laplacian_from_kgraph(kg)
    _, _, nnz_a = lap.info()
    L.check(L.load().ae_synchronize())
    lap.do_svd(want_u=False)  # warm
    svd_reps = 5
    L.check(L.load().ae_synchronize())
    t0 = time.perf_counter()
    for _ in range(svd_reps):
        lap.do_svd(want_u=False)  # U stays in HBM, as in the embedder's own call; the spectrum comes back
    svd_s = (time.perf_counter() - t0) / svd_reps
    y0 = A.DiffusionMaps(dp).embed_from_kgraph(kg)
    y0 = A.set_data_box(y0, 10.0)
    node_params = A.to_proba_edges(kg, 1.0, 1.0)

    mode = {"auto": A.AE_CE_AUTO, "event": A.AE_CE_EVENT, "rounds": A.AE_CE_HOGWILD, "sequential": A.AE_CE_SEQUENTIAL, "ordered": A.AE_CE_ORDERED}[args.ce_mode]
    head = time_mode(A, L, kg, node_params, y0, d, mode, args.steps, args.warmup, fence=fence)
    head_eo = head.pop("eo")
    params = A.EmbedderParams(asked_dim=d, nb_grad_batch=25, nb_sampling_by_edge=10, grad_step=1.0, scale_rho=1.0, beta=1.0)
    nb_batch = max(25, args.warmup + args.steps + 1)
    del head_eo

    # the same graph and start in the other modes.  parity_mode: AE_CE_SEQUENTIAL, the mode that meets the north star's tolerance (the
    # reference's sequential loop bit for bit: 1e-4 relative on the coordinates holds trivially) -- first class beside `value`
    def mode_entry(r, name, faithful):
        return {"ce_mode": name, "faithful": faithful, "dtype": r["dtype"], "ms_per_step": r["ms_per_step"], "points_per_s": n / (r["ms_per_step"] * 1e-3),
                "samples_per_s": r["nb_sample"] / (r["ms_per_step"] * 1e-3), "roofline": roofline_of(r, k, d)}
    rounds_mode = parity_mode = event_mode = fidelity = None
    if head["mode"] != A.AE_CE_EVENT and not args.lattice_graph:
        r = time_mode(A, L, kg, node_params, y0, d, A.AE_CE_EVENT, max(3, args.steps // 2), 1)
        r.pop("eo")
        event_mode = mode_entry(r, MODE_NAMES[3], "statistically (see fidelity)")
    if head["mode"] != A.AE_CE_HOGWILD:
        r = time_mode(A, L, kg, node_params, y0, d, A.AE_CE_HOGWILD, args.steps, args.warmup)
        r.pop("eo")
        rounds_mode = mode_entry(r, MODE_NAMES[0], False)
    if not args.no_exact_mode:
        if head["mode"] == A.AE_CE_SEQUENTIAL:
            r = head
        else:
            r = time_mode(A, L, kg, node_params, y0, d, A.AE_CE_SEQUENTIAL, max(5, args.steps // 2), 2)
            r.pop("eo")
        parity_mode = mode_entry(r, MODE_NAMES[1], True)
        parity_mode["tolerance"] = ("bit-identical to the CPU oracle's sequential loop (tests/test_gpu_parity.py: every row stride, both samplers, the whole "
                                    "25-batch schedule at this size): the north star's 1e-4 relative on the coordinates at fixed seed is met by this mode")
        parity_mode["roofline"]["latency_bound_note"] = "4305 dependency levels deep on this graph (tools/dependency_depth.py): %.2f us per level" % (
            parity_mode["roofline"]["launch_avg_ms"] * 1e3 / 4305.0)
    if not args.no_fidelity:
        # the full 25-batch schedule from the dmap initialisation in the three modes: what each mode converges to
        ys, ces = full_schedule(A, kg, node_params, y0, d, A.AE_CE_SEQUENTIAL)
        qs = edge_quartiles(indptr, nbr, ys)
        fidelity = {"schedule": "25 batches from the dmap initialisation, same graph and start", "reference": "AE_CE_SEQUENTIAL (bit-exact vs the oracle's sequential loop)",
                    "ce_sequential": ces, "edge_quartiles_sequential": qs.tolist()}
        for name, m in (("ordered", A.AE_CE_ORDERED), ("event", A.AE_CE_EVENT), ("sliced", A.AE_CE_SLICED), ("rounds", A.AE_CE_HOGWILD)):
            ym, cem = full_schedule(A, kg, node_params, y0, d, m)
            qm = edge_quartiles(indptr, nbr, ym)
            fidelity[name] = {"ce/ce_seq": cem / ces, "q25_ratio": qm[0] / qs[0], "q50_ratio": qm[1] / qs[1], "q75_ratio": qm[2] / qs[2]}
        # a second sequential run with another seed: the reference's own spread
        p2 = A.EmbedderParams(asked_dim=d, nb_grad_batch=25, nb_sampling_by_edge=10, grad_step=1.0, scale_rho=1.0, beta=1.0, ce_mode=A.AE_CE_SEQUENTIAL, seed=12345)
        y2, _, ce2 = A.entropy_optimize(kg, node_params, p2, y0)
        q2 = edge_quartiles(indptr, nbr, y2)
        fidelity["sequential_other_seed"] = {"ce/ce_seq": ce2 / ces, "q25_ratio": q2[0] / qs[0], "q50_ratio": q2[1] / qs[1], "q75_ratio": q2[2] / qs[2]}

    cpu = None
    if not args.no_cpu_baseline:
        cpu = cpu_baseline(indptr, nbr, node_params, y0, params, n, nb_batch)

    scale_shapes = None
    if not args.no_scale_shapes and not args.lattice_graph:
        del kg, node_params, lap
        torch.cuda.empty_cache()
        scale_shapes = {
            "c3_shape": scale_shape(A, L, "c3", 1_650_000, 6, 2, 6, with_sequential=True),
            # configs[2] again on a graph with REAL in-degree skew: exact kNN of the Higgs-shaped points, hubness weighting on
            "c3_knn_shape": scale_shape(A, L, "c3knn", 1_650_000, 6, 2, 6, with_sequential=True, hub_weighting=True, dmap_start=True,
                                        graph=exact_knn_graph(A, higgs_shaped_points(1_650_000), 6, "Higgs-shaped points (28-D, 64 components)")),
            # configs[3] on its own kind of graph: 11 M Higgs-shaped points, hubness weighting on as examples/higgs.rs:204-242
            # (the global graph of these points falls into ~40 components -- 0.000x % of its edges leave their cluster --: the
            # diffusion-map start is tried as the reference would, and timed: svd_init)
            "c4_knn_shape": scale_shape(A, L, "c4knn", 11_000_000, 6, 8, 5, with_sequential=False, hub_weighting=True, graph=config_graphs(A, "c4"), dmap_start=True,
                                        svd_init=True),
            "c4_shape": scale_shape(A, L, "c4", 11_000_000, 6, 8, 5, with_sequential=True, svd_init=True),
            # configs[4]: one GPU's eighth of the 50 M nodes as a graph of its own (k = 10, 16-D) -- the mixture's kNN graph, then the lattice
            "c5_shard_knn_shape": scale_shape(A, L, "c5knn", 6_250_000, 10, 16, 5, with_sequential=False, graph=config_graphs(A, "c5")),
            "c5_shard_shape": scale_shape(A, L, "c5", 6_250_000, 10, 16, 5, with_sequential=False),
        }
        if not args.no_full_size:   # configs[4] whole: 50 M points of the 128-D mixture, 5 G samples per batch, on this one GPU (~1 min, mostly the graph)
            scale_shapes["c5_full_shape"] = full_size_shape(A, L, "c5_full", 16, 2)
    # configs[4]'s "MFMA SVD" leg: the dense range finder on the 128-column data matrix itself, a rank's share and the whole
    svd_dense_c5 = svd_dense_c5_full = None
    if not args.no_dense_svd and not args.lattice_graph and not args.no_scale_shapes:
        svd_dense_c5 = svd_dense_shape(A, L, 6_250_000)
        if not args.no_full_size:
            svd_dense_c5_full = svd_dense_shape(A, L, 50_000_000, reps=2)

    roof = roofline_of(head, k, d)
    if head["mode"] in (1, 6):
        depth = 4305.0 if head["mode"] == 1 else 1565.0
        roof["latency_bound_note"] = ("the batch is a dependency chain, not a stream: %d levels deep on this graph (tools/dependency_depth.py), "
                                      "%.2f us per level in this run; DESIGN.md 4.4" % (depth, roof["launch_avg_ms"] * 1e3 / depth))
    replay = pmc_traffic(head["mode"])
    if replay:
        roof.update(replay)
    out = {
        "metric": "embedded_points_per_sec_ce_epoch",
        "value": n / (head["ms_per_step"] * 1e-3),
        "unit": "points/s",
        "n_gpus": 1,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": head["ms_per_step"],
        "higher_is_better": True,
        "scaling": "strong",  # the --gpus N series is strong scaling of the configs[3] shape; its one-GPU point is scale_shapes.c4_shape (this line's value is configs[1])
        "vs_baseline": None,
        "dtype": head["dtype"],
        "cpu_baseline": cpu,
        "data": "synthetic",
        "config": {
            "workload": ("ring-lattice kNN graph (node ids permuted) %d nodes -> %dD, k=%d, dmap init + CE loop (scale run)" % (n, d, k))
                        if args.lattice_graph else
                        ("MNIST-fashion-shaped %dx%d -> %dD, k=%d, dmap init + CE loop (configs[1])" % (n, args.dim, d, k)),
            "nb_sampling_by_edge": 10, "samples_per_step": int(head["nb_sample"]), "ce_mode": MODE_NAMES.get(head["mode"]),
            "ce_mode_requested": args.ce_mode,
            "value_is": "the default mode (AE_CE_AUTO); the figure at the north star's coordinate tolerance is parity_mode, beside it",
        },
        "roofline": roof,
        "parity_mode": parity_mode,
        "fidelity": fidelity,
        "event_mode": event_mode,
        "rounds_mode": rounds_mode,
        "scale_shapes": scale_shapes,
        "multi_gpu_note": ("--gpus N shards the source nodes of configs[3]'s kNN graph (component order) over the ranks in the default mode: AE_CE_AUTO on a node range is "
                           "the time-sliced mode, faithful where few edges cross shards (DESIGN 5); scale_shapes.c4_knn_shape is the same graph (node ids permuted) on one "
                           "GPU, c5_shard_knn_shape one eighth of configs[4]'s"),
        "svd_init": {"gflops": svd_flops(n, nnz_a) / svd_s / 1e9, "ms": svd_s * 1e3, "nnz_laplacian": int(nnz_a), "rank": 20, "nbiter": 5},
        "svd_init_c4": ((scale_shapes or {}).get("c4_knn_shape") or {}).get("svd_init") or ((scale_shapes or {}).get("c4_shape") or {}).get("svd_init"),
        "svd_dense": svd_dense,
        "svd_dense_c5": svd_dense_c5,
        "svd_dense_c5_full": svd_dense_c5_full,
        "knn_producer": knn_producer,
        "samples_per_s": head["nb_sample"] / (head["ms_per_step"] * 1e-3),
        "ce_before": head["ce_before"], "ce_after": head["ce_after"],
    }
    emit(out)


def multi_gpu(args, A, L, dist, torch, rank, world, fence):
    """N > 1 (or --force-dist): the source nodes of ONE graph sharded over the ranks, rounds mode, the owned coordinate rows
    exchanged inside ae_entropy_optim_gradient_iteration by the library's RCCL communicator.  Default: strong scaling of the
    configs[3] shape (fixed --scale-nodes graph); --weak: 60 k MNIST-shaped points per GPU."""
    from annembed_amd.dist import LibraryComm, ShardedCE, HipBackend, device_tensor, shard_range
    if args.weak:
        ppg, k, d = args.points_per_gpu, args.knbn, args.asked_dim
        n = ppg * world
        lo, hi = rank * ppg, (rank + 1) * ppg
        x = synth_points(n, args.dim, seed=1)
        nbr_l, dist_l = knn_rows(x, lo, hi, k)
        nbr_all = torch.empty((n, k), dtype=torch.int64, device=x.device)
        dist_all = torch.empty((n, k), dtype=torch.float32, device=x.device)
        if world > 1:
            dist.all_gather_into_tensor(nbr_all, nbr_l)
            dist.all_gather_into_tensor(dist_all, dist_l)
        else:
            nbr_all, dist_all = nbr_l, dist_l
        del x
        indptr = np.arange(n + 1, dtype=np.uint64) * np.uint64(k)
        nbr = nbr_all.cpu().numpy().astype(np.uint32).reshape(-1)
        dst = dist_all.cpu().numpy().reshape(-1)
        del nbr_all, dist_all, nbr_l, dist_l
        workload = "MNIST-fashion-shaped %dx%d -> %dD, k=%d, %d points per GPU (weak scaling)" % (n, args.dim, d, k, ppg)
        scaling = "weak"
    elif args.rounds:
        n, k, d = args.scale_nodes, 6, 8
        lo, hi = shard_range(n, world, rank)
        indptr, nbr, dst = lattice_graph(n, k, seed=7, permute=True)  # the same graph on every rank
        workload = "Higgs-11M-shaped ring lattice (node ids permuted) %d nodes -> %dD, k=%d, source nodes sharded over %d GPUs (configs[3], strong scaling, rounds mode)" % (n, d, k, world)
        scaling = "strong"
    else:
        # configs[3] on its own kind of graph, FAITHFUL, in the REFERENCE's kind of node order: the kNN graph of the Higgs-shaped points with
        # the node ids SHUFFLED GLOBALLY (file order carries no locality, kgraph.rs:489,500).  The library's partitioner
        # (ae_kgraph_partition: connected components packed whole, what must be cut is bisected and smoothed on the graph) gives the
        # ranks their contiguous ranges; every rank computes the same partition from the same graph (checked below).
        d = 8
        gr = config_graphs(A, "c4", permute_seed=9, n_override=args.scale_nodes)
        n, k = gr["n"], gr["k"]
        g0 = A.KGraph(gr["indptr"], gr["nbr"], gr["dist"], k)
        t0 = time.perf_counter()
        order, ranges, part = g0.partition(world)
        part["seconds"] = time.perf_counter() - t0
        naive_cross = 1.0 - 1.0 / world   # shuffled ids: contiguous id ranges would cut (world - 1) / world of the edges
        kg_ready = g0.permuted(order)   # (the relabelled graph stays on the device)
        nnz = len(gr["nbr"])
        chk = torch.tensor([float(gr["nbr"][::1009].astype(np.float64).sum()), float(order[::997].astype(np.float64).sum())], dtype=torch.float64,
                           device="cuda" if args.backend == "nccl" else "cpu")
        del g0, gr
        lo, hi = ranges[rank]
        if world > 1:  # every rank built and partitioned the graph itself: it must be the same graph in the same order
            lo_hi = [chk.clone() for _ in range(world)]
            dist.all_gather(lo_hi, chk)
            if any(bool((v != chk).any()) for v in lo_hi):
                raise SystemExit("the ranks built different graphs / partitions")
        workload = ("kNN graph (k = %d) of %d Higgs-shaped points (28-D, 64 components), node ids shuffled globally -> %dD, partitioned by the library into %d contiguous "
                    "ranges (configs[3], strong scaling, faithful time-sliced mode)" % (k, n, d, world))
        scaling = "strong"
    faithful = not args.weak and not args.rounds
    if not faithful:
        ranges = [shard_range(n, world, r) for r in range(world)]
    part_info = dict(part, naive_cross_mass_of_id_ranges=naive_cross) if faithful else None
    torch.cuda.empty_cache()
    if faithful:
        kg = kg_ready
    else:
        kg = A.KGraph(indptr, nbr, dst, k)
        nnz = len(nbr)
    hub = kg.hubness() if faithful else None
    if faithful:
        y0 = A.set_data_box(np.random.default_rng(1).normal(size=(n, d)).astype(np.float32), 10.0)   # (a component-wise kNN graph is disconnected: no diffusion-map start)
    else:
        y0 = A.set_data_box(A.DiffusionMaps(A.DiffusionParams(d, 5.0, 12)).embed_from_kgraph(kg), 10.0)  # replicated
    # what every rank REPEATS (replicated by a measured decision): edge weights, the handle (per-node records of the whole
    # graph, alias tables), the edge colouring (at attach: ce_slice_prepare); timed so that the line can say what share of a sharded
    # embedding they are
    L.check(L.load().ae_synchronize())
    t_rep = time.perf_counter()
    node_params = A.to_proba_edges(kg, 1.0, 1.0)
    L.check(L.load().ae_synchronize())
    node_params_s = time.perf_counter() - t_rep
    nb_batch = max(25, args.warmup + args.steps + 1)
    params = A.EmbedderParams(asked_dim=d, nb_grad_batch=nb_batch, nb_sampling_by_edge=10, grad_step=1.0, scale_rho=1.0, beta=1.0,
                              ce_mode=A.AE_CE_AUTO if faithful else A.AE_CE_HOGWILD, hubness_weighting=hub is not None)
    t_rep = time.perf_counter()
    eo = A.EntropyOptim(kg, node_params, params, y0, node_lo=lo, node_hi=hi, hub_counts=hub)
    L.check(L.load().ae_synchronize())
    create_s = time.perf_counter() - t_rep
    nb_sample = params.nb_sampling_by_edge * eo.get_nb_edges()
    library_comm = args.backend == "nccl"
    comm = sharded = None
    torch_gather = None
    comm_error = None
    if library_comm:
        try:
            comm = LibraryComm(rank, world)
            t_rep = time.perf_counter()
            comm.attach(eo, args.exchanges)
            L.check(L.load().ae_synchronize())
            create_s += time.perf_counter() - t_rep   # (the time-sliced mode prepares here: colouring + static records of the whole graph)
        except Exception as e:  # e.g. librccl.so.1 not loadable from the library: agree on the fallback below
            comm, comm_error = None, repr(e)[:300]
        ok = torch.tensor([1 if comm is not None else 0], device="cuda")
        if world > 1:
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:  # every rank: the same exchange through torch's RCCL, in place, on the library stream, once per batch
            if comm is not None:
                comm.attach_none(eo)
                comm.close()
                comm = None
            library_comm = False
            torch_gather = ShardedCE(HipBackend(eo), device_tensor(eo), n, d, rank, world, ranges=ranges)
    else:  # validation over gloo: ranks share a GPU, the exchange goes through torch (once per batch)
        sharded = ShardedCE(HipBackend(eo), device_tensor(eo), n, d, rank, world, ranges=ranges)
    ce_before = eo.ce_compute_threaded()

    def one_step(it):
        eo.gradient_iteration_threaded(nb_sample, params.grad_step * (1.0 - it / nb_batch), it)
        if torch_gather is not None:
            torch_gather.all_gather()
        if sharded is not None:
            L.check(L.load().ae_synchronize())
            y_host = device_tensor(eo).cpu()
            sizes = [r_hi - r_lo for r_lo, r_hi in ranges]
            own = torch.zeros((max(sizes), d))
            own[:hi - lo] = y_host[lo:hi]
            parts = [torch.empty((max(sizes), d)) for _ in range(world)]  # gloo wants equal shapes: padded
            dist.all_gather(parts, own)
            device_tensor(eo).copy_(torch.cat([parts[r][:sizes[r]] for r in range(world)]).cuda())
            torch.cuda.synchronize()

    it = 0
    for _ in range(args.warmup):
        it += 1
        one_step(it)
    fence()
    eo.kernel_time()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        it += 1
        one_step(it)
    fence()
    elapsed = time.perf_counter() - t0
    kernel_ms, launches = eo.kernel_time()
    tt = torch.tensor([elapsed, kernel_ms], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    elapsed, kernel_ms_max = float(tt[0].item()), float(tt[1].item())
    ce_local = eo.ce_compute_threaded()
    if comm is not None:
        ce_after, ce0 = comm.all_reduce_sum(ce_local), comm.all_reduce_sum(ce_before)
    else:
        t2 = torch.tensor([ce_local, ce_before], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        if world > 1:
            dist.all_reduce(t2)
        ce_after, ce0 = float(t2[0]), float(t2[1])
    if rank == 0:
        resolved = eo.get_ce_mode()
        rounds = int(eo.samples_drawn()[1]) if resolved == 0 else 1
        run = dict(rounds=rounds, kernel_ms=kernel_ms_max, ms_per_step=elapsed / args.steps * 1e3, nb_sample=nb_sample, batches_timed=int(launches), mode=resolved)
        if resolved == 5:
            cl, ovf, crounds, slices = eo.slice_info()
            run["sliced"] = {"classes": cl, "overflow_mass_fraction": ovf, "slices_per_batch": slices, "max_in_degree": eo.slice_hub_info()[0],
                             "launch_form": SLICE_FORMS.get(eo.slice_form(), "?")}
        roof = roofline_of(run, k, d)
        roof["note"] = "per GPU: bytes of this rank's samples / the slowest rank's batch time, collectives included"
        ms_b = elapsed / args.steps * 1e3
        rep_s = (part_info or {}).get("seconds", 0.0) + node_params_s + create_s
        end_to_end = {"partition_s": (part_info or {}).get("seconds"), "node_params_s": node_params_s, "prepare_s": create_s, "replicated_s": rep_s, "ms_per_batch": ms_b,
                      "batches": 40, "replicated_share_of_a_40_batch_embedding": rep_s / (rep_s + 40 * ms_b * 1e-3),
                      "note": "every rank repeats partition, edge weights and the handle's preparation (colouring + records of the whole graph); the dmap initialisation of "
                              "embed() is replicated too (configs[3]: 0.1 s, configs[4]: 1.1 s; svd_init_c4 in the --gpus 1 line)"}
        exch = args.exchanges if library_comm else 1
        out = {
            "metric": "embedded_points_per_sec_ce_epoch",
            "value": n * args.steps / elapsed,
            "unit": "points/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": mode_dtype(resolved),
            "data": "synthetic" if args.backend == "nccl" else "synthetic (VALIDATION RUN over gloo, ranks sharing a GPU: not a result)",
            "config": {"workload": workload, "nb_sampling_by_edge": 10, "samples_per_step": int(10 * nnz), "ce_mode": MODE_NAMES.get(resolved),
                       "exchanges_per_batch": exch, "rccl_ranks": world if library_comm else 0,
                       "partition": part_info, "cross_shard_mass": (part_info or {}).get("cross_mass"),
                       "bytes_received_per_rank_and_batch": int(exch * n * d * 4),
                       "collective": "in-place RCCL all-gather of the owned rows inside ae_entropy_optim_gradient_iteration (library communicator)" if library_comm
                       else ("in-place RCCL all-gather through torch.distributed on the library stream (the library communicator failed: %s)" % comm_error if torch_gather is not None
                             else "torch/gloo (validation)")},
            "roofline": roof,
            "per_rank_batch_ms_max": kernel_ms_max,
            "end_to_end": end_to_end,
            "faithful": (faithful_of(run) + "; on node ranges: tests/test_gpu_configs.py::test_sharded_sliced_*, DESIGN 5") if resolved == 5 else False,
            "multi_gpu_note": ("every rank runs the time-sliced mode on its own node range: its events on current rows, the other ranks' rows (negatives, the far ends of the "
                               "few cross-shard edges) as of the last all-gather; measured on one GPU with 2 and 8 processes: the result does not depend on the exchanges "
                               "per batch (1 ... 240), DESIGN 5" if resolved == 5 else
                               "the rounds mode (asked for with --rounds / --weak) is approximate: its output is not the reference's (DESIGN 4.2)"),
            "n1_like_for_like": ("the --gpus 1 line measures configs[1] in the default mode (its `value` is NOT the one-GPU point of this series); the same mode on ONE GPU on "
                                 "the node-permuted graph of the same points is its key scale_shapes.c4_knn_shape.default_mode (points_per_s, ms_per_step)" if resolved == 5 else
                                 "the --gpus 1 line's key scale_shapes.c4_shape.rounds_mode / rounds_mode (same shape and mode on one GPU)"),
            "samples_per_s": 10 * nnz * args.steps / elapsed,
            "ce_before": ce0, "ce_after": ce_after,
        }
    if comm is not None:
        del eo
        comm.close()
    return out if rank == 0 else None


def pmc_traffic(mode):
    """HBM bytes per launch of the dominant kernel REPLAYED from the committed rocprofv3 PMC passes of this same command
    (profiles/<round>/pmc_ce_*.json, written by tools/prof_bench.sh: FETCH_SIZE and WRITE_SIZE collected in separate passes).
    Not a measurement of this run: `traffic_from` says where it comes from."""
    import glob
    name = {0: "pmc_ce_round.json", 1: "pmc_ce_dataflow.json", 3: "pmc_ce_event.json", 6: "pmc_ce_ordered.json"}.get(mode)
    if not name:
        return None
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", name)))
    if not files:
        return None
    try:
        with open(files[-1]) as f:
            j = json.load(f)
        # `traffic` = HBM bytes per launch from the committed rocprofv3 PMC passes of this same command (separate --pmc passes, the
        # guide's gfx950 correction applied by tools/prof_bench.sh); rocprofv3 wraps the process, so it cannot be collected from inside
        return {"traffic": float(j["hbm_bytes_per_launch"]), "traffic_from": os.path.relpath(files[-1], ROOT) + " (replayed: a PMC run of this command, not this run)"}
    except Exception:
        return None


def cpu_baseline(indptr, nbr, node_params, y0, params, n, nb_batch):
    """The CPU restatement of the reference's Hogwild loop (oracle, kind "port") on the host cores,
    on a bounded sample of the same workload."""
    # (the OpenMP workers of this leg sleep between parallel regions instead of spinning: nothing of it is left on the host cores when
    # the launch-heavy scale shapes are timed afterwards)
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")
    from oracle import oracle as O
    proba, scale = node_params.get()
    eo = O.EntropyOptim(indptr, nbr, proba, scale, y0, b=params.b, seed=params.seed, sampler=0)
    nb_sample = params.nb_sampling_by_edge * len(nbr)
    cores = O.max_threads()
    # the cores this process may really use: the scheduler affinity and a cgroup CPU quota, if any (OpenMP sees the machine's)
    try:
        usable = len(os.sched_getaffinity(0))
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            usable = min(usable, max(1, int(float(quota) / float(period))))
        cores = max(1, min(int(cores), usable))
    except Exception:
        cores = int(cores)
    eo.gradient_iteration_hogwild(nb_sample // 8, 1.0, 1, cores)  # warm the thread pool
    batches = 0
    t0 = time.perf_counter()
    while True:
        batches += 1
        eo.gradient_iteration_hogwild(nb_sample, params.grad_step * (1.0 - batches / nb_batch), batches + 1, cores)
        el = time.perf_counter() - t0
        if el > 10.0 or batches >= 8:
            break
    return {
        "value": n * batches / el, "unit": "points/s", "cores": int(cores), "kind": "port",
        "sample": "%d CE batches (%d SGD samples each) of the same graph, OpenMP Hogwild restatement of "
                  "src/embedder.rs:1311-1315 on all host cores (a cache line per row, dynamic chunks of 4096 samples)" % (batches, nb_sample),
        "samples_per_s": nb_sample * batches / el, "samples_per_s_per_core": nb_sample * batches / el / max(1, int(cores)),
    }


if __name__ == "__main__":
    main()
