#!/usr/bin/env python3
"""bench.py -- headline benchmark of the annembed hot path on MI355X.

Metric (BASELINE.json): embedded points/sec per CE epoch (one `gradient_iteration` batch over
nb_sampling_by_edge * nnz SGD samples), plus the SVD-init GFLOP/s of the diffusion-map initialisation.

Workload at N=1: configs[1] "MNIST-fashion 60k x 784 -> 2D, k=12, dmap init + CE loop, fp32"
(parameters of examples/mnist_fashion.rs:92-110).  The real dataset is absent (no network): a
synthetic Gaussian mixture of the same shape stands in and the exact kNN graph is built
on the GPU before the timed region.  N>1: weak scaling -- every rank owns 60k source nodes of an
N*60k point graph, coordinates are replicated and all-gathered (RCCL) once per CE batch.

A "step" = one CE batch.  The timed region holds only `ae_entropy_optim_gradient_iteration` launches
(+ the per-batch all-gather when N>1) with every input already resident in HBM.

Prints ONE JSON line on rank 0.
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
    ap.add_argument("--lattice-graph", action="store_true",
                    help="scale runs (C3 / C4 shapes): ring-lattice kNN graph with gamma-distributed distances instead of an exact kNN of synthetic points (an 11 M-point exact kNN is out of reach of brute force)")
    ap.add_argument("--force-dist", action="store_true", help="exercise the collective path with world size 1 (validation)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="validation only: gloo lets several ranks share ONE GPU (RCCL refuses duplicate devices), so the whole "
                         "N > 1 code path -- sharded node ranges, gathered kNN rows, per-batch all-gather, max-over-ranks timing -- "
                         "can be run end to end on a single-GPU box; the figure it prints is not a benchmark result")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
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
    L.check(L.load().ae_set_device(local_rank))

    ppg, k, d = args.points_per_gpu, args.knbn, args.asked_dim
    n = ppg * world
    lo, hi = rank * ppg, (rank + 1) * ppg

    # ---------------- input preparation (untimed) ----------------
    if args.lattice_graph:
        rng = np.random.default_rng(1)
        base = np.arange(n, dtype=np.int64)
        offs = np.array([1, 2, 3, 5, 8, 13, 21, 34, 55, 89, 144, 233, 377, 610, 987, 1597][:(k + 1) // 2])
        cols = [(base + o) % n for o in offs] + [(base - o) % n for o in offs]
        nbr = np.stack(cols[:k], 1).astype(np.uint32).reshape(-1)
        dst = np.sort(rng.gamma(2.0, 1.0, size=(n, k)).astype(np.float32), axis=1).reshape(-1)
        indptr = np.arange(n + 1, dtype=np.uint64) * np.uint64(k)
        svd_dense = None
        knn_producer = None
    else:
        x = synth_points(n, args.dim, seed=1)
        nbr_l, dist_l = knn_rows(x, lo, hi, k)
        if world > 1:
            nbr_all = torch.empty((n, k), dtype=torch.int64, device=x.device)
            dist_all = torch.empty((n, k), dtype=torch.float32, device=x.device)
            dist.all_gather_into_tensor(nbr_all, nbr_l)
            dist.all_gather_into_tensor(dist_all, dist_l)
        else:
            nbr_all, dist_all = nbr_l, dist_l
        # secondary figure (rank 0, N = 1): the dense range finder of tools::svdapprox on the data matrix itself --
        # subspace_iteration_full + direct_svd, rank 20, 5 iterations -- the MFMA tall-skinny products
        svd_dense = None
        if world == 1 and not args.no_dense_svd:
            mat = A.MatRepr.from_array2(x.cpu().numpy())
            A.SvdApprox(mat).direct_svd(A.RangeRank(20, 5))  # warm
            L.check(L.load().ae_synchronize())
            t0 = time.perf_counter()
            for _ in range(3):
                A.SvdApprox(mat).direct_svd(A.RangeRank(20, 5))
            dt = (time.perf_counter() - t0) / 3
            m_, n_, l_ = n, args.dim, 20
            fl = 9 * 2 * m_ * n_ * l_ + 5 * 4 * m_ * l_ * l_ + 4 * 4 * n_ * l_ * l_ + 2 * m_ * n_ * l_  # SURVEY 8d, dense path
            svd_dense = {"shape": "%dx%d rank 20 nbiter 5" % (m_, n_), "ms": dt * 1e3, "tflops": fl / dt / 1e12,
                         "mfma_f32_peak_tflops": 157.3, "mfma_frac": fl / dt / 1e12 / 157.3,
                         "hbm_gbps": 10 * 4.0 * m_ * n_ / dt / 1e9}
            del mat
        # secondary figure (rank 0, N = 1): the exact kNN-graph producer on the matrix cores, host matrix in,
        # KGraph out (includes the PCIe upload of the points)
        knn_producer = None
        if world == 1 and not args.no_dense_svd:
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
        del nbr_all, dist_all, nbr_l, dist_l
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

    params = A.EmbedderParams(asked_dim=d, nb_grad_batch=25, nb_sampling_by_edge=10, grad_step=1.0, scale_rho=1.0, beta=1.0)
    eo = A.EntropyOptim(kg, node_params, params, y0, node_lo=lo, node_hi=hi)
    nnz_shard = eo.get_nb_edges()
    nb_sample = params.nb_sampling_by_edge * nnz_shard
    nb_batch = max(params.nb_grad_batch, args.warmup + args.steps + 1)
    ce_before = eo.ce_compute_threaded()

    y_all = None
    lib_stream = None
    if use_dist:
        ptr, nn, dd = eo.device_coords()

        class _Arr:  # wraps the library's device buffer as a torch tensor (no copy)
            __cuda_array_interface__ = {"shape": (nn, dd), "typestr": "<f4", "data": (ptr, False), "version": 2}
        y_all = torch.as_tensor(_Arr(), device="cuda")
        # the library's HIP stream as a torch stream: the collective is ordered after the batch's kernels and before the
        # next batch's by stream events (torch's NCCL work waits on / is waited by the current stream) -- no host sync
        import ctypes
        sp = ctypes.c_void_p()
        L.check(L.load().ae_get_stream(ctypes.byref(sp)))
        lib_stream = torch.cuda.ExternalStream(sp.value)
        y_gather = torch.empty((nn, dd), dtype=torch.float32, device="cuda")

    def one_step(it):
        eo.gradient_iteration_threaded(nb_sample, params.grad_step * (1.0 - it / nb_batch), it)
        if use_dist:
            with torch.cuda.stream(lib_stream):
                # the collective works on torch-owned buffers (no assumption about RCCL and memory it did not see
                # allocated); the gathered replica is copied into the library's coordinate array on the same stream
                dist.all_gather_into_tensor(y_gather, y_all[lo:hi].clone())
                y_all.copy_(y_gather)

    def fence():
        L.check(L.load().ae_synchronize())
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    it = 0
    for _ in range(args.warmup):
        it += 1
        one_step(it)
    fence()
    eo.kernel_time()  # reset the event accumulators
    t0 = time.perf_counter()
    for _ in range(args.steps):
        it += 1
        one_step(it)
    fence()
    elapsed = time.perf_counter() - t0
    kernel_ms, launches = eo.kernel_time()
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    ce_after = eo.ce_compute_threaded()

    # secondary figure: the exact mode (AE_CE_SEQUENTIAL, bit-exact against the oracle's sequential loop -- tests),
    # same graph, same batch size
    exact_mode = None
    if world == 1 and not args.no_exact_mode:
        pe = A.EmbedderParams(asked_dim=d, nb_grad_batch=25, nb_sampling_by_edge=10, grad_step=1.0, scale_rho=1.0, beta=1.0,
                              ce_mode=A.AE_CE_SEQUENTIAL)
        ex = A.EntropyOptim(kg, node_params, pe, y0)
        ex.gradient_iteration_threaded(nb_sample, 0.9, 1)  # warm (allocations)
        L.check(L.load().ae_synchronize())
        t0 = time.perf_counter()
        reps = 3
        for r in range(reps):
            ex.gradient_iteration_threaded(nb_sample, 0.9, 2 + r)
        L.check(L.load().ae_synchronize())
        dt = (time.perf_counter() - t0) / reps
        exact_mode = {"ce_mode": "sequential (device-scheduled dataflow, bit-exact vs the oracle)", "ms_per_step": dt * 1e3,
                      "points_per_s": n / dt, "samples_per_s": nb_sample / dt}
        del ex

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        points_per_s = n * args.steps / elapsed
        bytes_per_sample = 24 + 4 * k + 36 * d  # SURVEY 8d / DESIGN.md
        # dominant kernel: ce_round_node_kernel, one launch per round; kernel_ms = hipEvent duration of one batch
        # (= `rounds` back-to-back launches) on the library's stream, averaged over the timed steps
        rounds = int(eo.samples_drawn()[1])
        launch_ms = kernel_ms / rounds if rounds else 0.0
        bytes_per_launch = bytes_per_sample * nb_sample / max(rounds, 1)
        achieved = bytes_per_launch / (launch_ms * 1e-3) / 1e9 if launch_ms > 0 else 0.0
        # the committed PMC passes are of the default single-GPU workload only
        default_workload = (world == 1 and not args.lattice_graph and ppg == 60000 and k == 12 and d == 2 and args.dim == 784)
        traffic = pmc_traffic() if default_workload else None
        out = {
            "metric": "embedded_points_per_sec_ce_epoch",
            "value": points_per_s,
            "unit": "points/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic" if args.backend == "nccl" else "synthetic (VALIDATION RUN over gloo, ranks sharing a GPU: not a result)",
            "config": {
                "workload": ("ring-lattice kNN graph %d nodes -> %dD, k=%d, dmap init + CE loop (scale run); %d points per GPU" % (n, d, k, ppg))
                            if args.lattice_graph else
                            ("MNIST-fashion-shaped %dx%d -> %dD, k=%d, dmap init + CE loop (configs[1]); %d points per GPU"
                             % (n, args.dim, d, k, ppg)),
                "nb_sampling_by_edge": 10, "samples_per_step": int(nb_sample * world), "sampler": "rowcdf", "ce_mode": "hogwild",
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                "traffic": traffic, "kernel": "ce_round_node_kernel (one launch per round, `rounds` launches per CE batch)",
                "rounds": rounds, "launch_avg_ms": launch_ms, "batch_kernel_ms": kernel_ms, "batches_timed": int(launches),
                "bytes_per_sample": bytes_per_sample, "bytes_per_launch": bytes_per_launch,
            },
            "svd_init": {
                "gflops": svd_flops(n, nnz_a) / svd_s / 1e9, "ms": svd_s * 1e3, "nnz_laplacian": int(nnz_a), "rank": 20, "nbiter": 5,
            },
            "svd_dense": svd_dense,
            "knn_producer": knn_producer,
            "exact_mode": exact_mode,
            "samples_per_s": nb_sample * world * args.steps / elapsed,
            "ce_before": ce_before, "ce_after": ce_after,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(indptr, nbr, node_params, y0, params, n, nb_batch)
        print(json.dumps(out))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def pmc_traffic():
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes of this same command
    (profiles/<round>/pmc_ce_round.json, written by tools/prof_bench.sh: FETCH_SIZE and WRITE_SIZE collected in separate
    passes, KiB -> bytes); None when no profile of the current kernel is committed."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_ce_round.json")))
    if not files:
        return None
    try:
        with open(files[-1]) as f:
            j = json.load(f)
        return float(j["hbm_bytes_per_launch"])
    except Exception:
        return None


def cpu_baseline(indptr, nbr, node_params, y0, params, n, nb_batch):
    """The CPU restatement of the reference's Hogwild loop (oracle, kind "port") on the host cores,
    on a bounded sample of the same workload."""
    from oracle import oracle as O
    proba, scale = node_params.get()
    eo = O.EntropyOptim(indptr, nbr, proba, scale, y0, b=params.b, seed=params.seed, sampler=0)
    nb_sample = params.nb_sampling_by_edge * len(nbr)
    cores = O.max_threads()
    eo.gradient_iteration_hogwild(nb_sample // 8, 1.0, 1, 0)  # warm the thread pool
    batches = 0
    t0 = time.perf_counter()
    while True:
        batches += 1
        eo.gradient_iteration_hogwild(nb_sample, params.grad_step * (1.0 - batches / nb_batch), batches + 1, 0)
        el = time.perf_counter() - t0
        if el > 10.0 or batches >= 8:
            break
    return {
        "value": n * batches / el, "unit": "points/s", "cores": int(cores), "kind": "port",
        "sample": "%d CE batches (%d SGD samples each) of the same graph, OpenMP Hogwild restatement of "
                  "src/embedder.rs:1311-1315 on all host cores" % (batches, nb_sample),
        "samples_per_s": nb_sample * batches / el,
    }


if __name__ == "__main__":
    main()
