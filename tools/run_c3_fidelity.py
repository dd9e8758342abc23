"""Fidelity of the fast mode on the C3 schedule (examples/higgs.rs:204-242: hierarchical, 5 x 40 batches on the small graph,
40 on the large one, scale_rho 0.75, hubness weighting) at a size the CPU oracle finishes: GPU fast mode vs GPU bit-exact
mode (= the oracle's sequential loop) vs the oracle's OpenMP Hogwild run (true racy updates on all host cores: the
reference's own rayon behaviour).  Same graphs, same projection.  usage: python tools/run_c3_fidelity.py [n] [out.json]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import annembed_amd as A  # noqa: E402
from oracle import oracle as O  # noqa: E402  (checker only: this is a measurement tool, not the product path)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60000
    out_path = sys.argv[2] if len(sys.argv) > 2 else None
    dim, k, ncomp = 28, 6, 64
    g = torch.Generator(device="cpu").manual_seed(2)
    means = torch.randn(ncomp, dim, generator=g) * 2.0
    scales = 0.5 + torch.rand(ncomp, dim, generator=g)
    lab = torch.randint(0, ncomp, (n,), generator=g)
    x = means[lab] + scales[lab] * torch.randn(n, dim, generator=g)
    x = (x - x.mean(0)) / x.std(0)
    xh = np.ascontiguousarray(x.numpy().astype(np.float32))
    n_small = n // 24
    large, small = A.KGraph.bruteforce_l2(xh, k), A.KGraph.bruteforce_l2(xh[:n_small], k)
    x64 = xh.astype(np.float64)
    dd = (x64 ** 2).sum(1)[:, None] + (x64[:n_small] ** 2).sum(1)[None, :] - 2 * x64 @ x64[:n_small].T
    pn = dd.argmin(1).astype(np.uint32)
    pd = np.sqrt(np.maximum(dd.min(1), 0)).astype(np.float32)
    pn[:n_small] = np.arange(n_small)
    pd[:n_small] = 0
    del dd
    res = {"n": n, "n_small": n_small, "runs": {}}

    def quality(y):
        q = A.quality_estimate_from_edge_length(large, y, 6)
        return {"nb_without_match": q.nb_without_match, "mean_matches": q.mean_nbmatch, "median_ratio": q.median_ratio}

    modes = (("gpu_fast", A.AE_CE_HOGWILD),) if os.environ.get("AE_FID_FAST_ONLY") else (("gpu_fast", A.AE_CE_HOGWILD), ("gpu_sequential", A.AE_CE_SEQUENTIAL))
    hub = os.environ.get("AE_FID_HUB", "1") == "1"
    rho = float(os.environ.get("AE_FID_RHO", "0.75"))
    nbatch = int(os.environ.get("AE_FID_BATCH", "40"))
    for name, mode in modes:
        par = A.EmbedderParams(asked_dim=2, nb_grad_batch=nbatch, grad_factor=5, scale_rho=rho, beta=1.0, grad_step=1.0,
                               nb_sampling_by_edge=10, dmap_init=True, hubness_weighting=hub, ce_mode=mode)
        if os.environ.get("AE_FID_FLAT"):
            par.dmap_init = os.environ.get("AE_FID_FLAT") == "dmap"
            emb = A.Embedder(large, par)
        else:
            emb = A.Embedder.from_hkgraph(A.KGraphProjection(small, large, pn, pd), par)
        t0 = time.perf_counter()
        emb.embed()
        dt = time.perf_counter() - t0
        ce = emb.get_cross_entropy()
        res["runs"][name] = dict(embed_s=dt, ce_before=ce[0], ce_after=ce[1], **quality(emb.get_embedded()))
    if os.environ.get("AE_FID_FAST_ONLY") or os.environ.get("AE_FID_NO_ORACLE"):
        print(json.dumps(res))
        return
    sm, lg = small.get_neighbours(), large.get_neighbours()
    op = O.EmbedderParams(asked_dim=2, nb_grad_batch=40, grad_factor=5, scale_rho=0.75, beta=1.0, grad_step=1.0,
                          nb_sampling_by_edge=10, dmap_init=True, hubness_weighting=True)
    t0 = time.perf_counter()
    rc, ref = O.h_embed((sm[0], sm[1], sm[2], k), (lg[0], lg[1], lg[2], k), pn, pd, op, hogwild_threads=0)
    dt = time.perf_counter() - t0
    res["runs"]["oracle_openmp_hogwild"] = dict(embed_s=dt, threads=int(O.max_threads()), ce_before=ref["ce_before"], ce_after=ref["ce_after"],
                                                **quality(ref["y"]))
    print(json.dumps(res))
    if out_path:
        with open(out_path, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
