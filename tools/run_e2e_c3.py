"""C3 end to end (BASELINE.json configs[2]): Higgs-15 %-shaped 1 650 000 x 28 -> 2-D, hierarchical initialisation
(examples/higgs.rs:204-242: nb_grad_batch 40, grad_factor 5, scale_rho 0.75, hubness weighting, knbn 6, projection on
layer 1), on one MI355X.  The real data set is absent: a 64-component Gaussian mixture, column-standardised as
examples/higgs.rs:158-176, stands in; the small graph is the first n/24 points (the share of HNSW layers >= 1 at the
reference's level scale), the projection is the nearest small point.  Prints one JSON line.

usage: python tools/run_e2e_c3.py [n] [out.json] [sequential]   (sequential: AE_CE_SEQUENTIAL, the bit-exact mode, for A/B)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import annembed_amd as A  # noqa: E402
from annembed_amd import _lib as L  # noqa: E402


def sync():
    L.check(L.load().ae_synchronize())
    torch.cuda.synchronize()


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1650000
    out_path = sys.argv[2] if len(sys.argv) > 2 else None
    exact = len(sys.argv) > 3 and sys.argv[3] == "sequential"
    dim, k, ncomp = 28, 6, 64
    g = torch.Generator(device="cpu").manual_seed(2)
    means = torch.randn(ncomp, dim, generator=g) * 2.0
    scales = 0.5 + torch.rand(ncomp, dim, generator=g)
    lab = torch.randint(0, ncomp, (n,), generator=g)
    x = means[lab] + scales[lab] * torch.randn(n, dim, generator=g)
    x = (x - x.mean(0)) / x.std(0)
    xh = np.ascontiguousarray(x.numpy().astype(np.float32))
    n_small = n // 24
    res = {"workload": "Higgs-15%%-shaped %d x %d -> 2-D, hierarchical (small graph %d nodes), k = %d" % (n, dim, n_small, k)}

    t0 = time.perf_counter()
    large = A.KGraph.bruteforce_l2(xh, k)
    sync()
    res["knn_large_s"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    small = A.KGraph.bruteforce_l2(xh[:n_small], k)
    sync()
    res["knn_small_s"] = time.perf_counter() - t0

    # projection of every point on its nearest small point (kgproj.rs: the points of the upper layers project on themselves)
    t0 = time.perf_counter()
    xd = torch.from_numpy(xh).cuda()
    xs = xd[:n_small]
    sq_s = (xs * xs).sum(1)
    proj_node = torch.empty(n, dtype=torch.int64, device="cuda")
    proj_dist = torch.empty(n, dtype=torch.float32, device="cuda")
    for b in range(0, n, 16384):
        e = min(b + 16384, n)
        d2 = (xd[b:e] * xd[b:e]).sum(1)[:, None] + sq_s[None, :] - 2.0 * (xd[b:e] @ xs.T)
        v, i = d2.min(1)
        proj_node[b:e] = i
        proj_dist[b:e] = v.clamp_min(0).sqrt()
    proj_node[:n_small] = torch.arange(n_small, device="cuda")
    proj_dist[:n_small] = 0.0
    pn, pd = proj_node.cpu().numpy().astype(np.uint32), proj_dist.cpu().numpy()
    del xd, xs, proj_node, proj_dist
    torch.cuda.empty_cache()
    res["projection_s"] = time.perf_counter() - t0

    par = A.EmbedderParams(asked_dim=2, nb_grad_batch=40, grad_factor=5, scale_rho=0.75, beta=1.0, grad_step=1.0,
                           nb_sampling_by_edge=10, dmap_init=True, hubness_weighting=True)
    if exact:
        par.ce_mode = A.AE_CE_SEQUENTIAL
    res["ce_mode"] = "sequential" if exact else "auto (the sequential-equivalent dataflow at both stages of this size)"
    proj = A.KGraphProjection(small, large, pn, pd)
    emb = A.Embedder.from_hkgraph(proj, par)
    sync()
    t0 = time.perf_counter()
    rc = emb.embed()
    sync()
    res["embed_s"] = time.perf_counter() - t0
    res["embed_rc"] = int(rc)
    y = emb.get_embedded()
    res["finite"] = bool(np.isfinite(y).all())
    res["cross_entropy"] = emb.get_cross_entropy()
    t0 = time.perf_counter()
    try:
        q = emb.get_quality_estimate_from_edge_length(6)
        sync()
        res["quality_s"] = time.perf_counter() - t0
        res["quality"] = {"nb_without_match": int(q.nb_without_match), "mean_matches": q.mean_nbmatch, "median_ratio": q.median_ratio,
                          "mean_ratio": q.mean_ratio, "ratio_quantiles": [float(v) for v in q.ratio_quantiles]}
    except Exception as ex:  # the exact radius graph is O(N^2): report instead of failing the run
        res["quality_error"] = str(ex)[:200]
    res["reference_prose"] = "Higgs 28 vars, 15 % subsample incl. quality estimate: 16 min on a 24-core i9 (README.md:148)"
    print(json.dumps(res))
    if out_path:
        with open(out_path, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
