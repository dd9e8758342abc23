"""Results parity against the only numbers the reference publishes on this path (src/embedder.rs:585-618): Fashion-MNIST, 70 000 images,
hierarchical embedding (examples/mnist_fashion.rs:92-125: knbn 6, nb_grad_batch 20, grad_factor 4, projection on layer 1; the flat
variant: knbn 12, nb_grad_batch 25), `get_quality_estimate_from_edge_length(50)`:

    asked_dim 2 : nb neighbourhoods without a match 20260, mean number of neighbours conserved 5.069, ratio quantiles .05 5.40e-2,
                  .25 3.28e-1, .5 7.46e-1, .75 1.57, .85 2.38, .95 4.50
    asked_dim 15: 9124 / 5.585 / median ratio 4.36e-1

Needs the four IDX files (train-/t10k- images-idx3-ubyte, labels-idx1-ubyte; format src/utils/mnistio.rs:56-147) in $AE_MNIST_DIR,
data/fashion-mnist/ or data/mnist/ -- there is no network here, so without them the tool says so and exits 0.  Differences from the
reference's run that remain: the kNN graph is EXACT (the reference takes it from an HNSW: ef 200, 16 connections), the small graph of
the hierarchical run is a random 1/16 of the points (the expected share of HNSW layers >= 1 with 16 connections) with its own exact kNN
graph and every point projected on its nearest small point (kgproj.rs), and the RNG streams are this build's.

usage: python tools/run_fashion_quality.py [out.json] [--dim 2|15] [--flat]
Prints one JSON line: per CE mode (AE_CE_SEQUENTIAL = the bit-exact replay of the reference's loop; the default mode) the quality
estimate beside the published one."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

PUBLISHED = {  # src/embedder.rs:585-618 (hierarchical case, nbng 50)
    2: {"nb_without_match": 20260, "mean_nbmatch": 5.069, "ratio_quantiles": [5.40e-2, 3.28e-1, 7.46e-1, 1.57, 2.38, 4.50],
        "radii_quantiles": [3.15e-2, 4.52e-2, 5.68e-2, 7.80e-2, 9.32e-2, 1.36e-1]},
    15: {"nb_without_match": 9124, "mean_nbmatch": 5.585, "ratio_quantiles": [5.03e-2, 2.24e-1, 4.36e-1, 8.06e-1, 1.13, 1.92],
         "radii_quantiles": [5.55e-2, 8.66e-2, 1.15e-1, 1.53e-1, 1.80e-1, 2.41e-1]},
}


def nearest_small(x, small_idx):
    """projection of every point on its nearest small point (kgproj.rs:376-410): (node in the small graph, distance)"""
    import torch
    xd = torch.from_numpy(x).cuda()
    xs = xd[torch.from_numpy(small_idx.astype(np.int64)).cuda()]
    sq_s = (xs * xs).sum(1)
    n = len(x)
    pn = torch.empty(n, dtype=torch.int64, device="cuda")
    pd = torch.empty(n, dtype=torch.float32, device="cuda")
    for b in range(0, n, 8192):
        e = min(b + 8192, n)
        d2 = (xd[b:e] * xd[b:e]).sum(1)[:, None] + sq_s[None, :] - 2.0 * (xd[b:e] @ xs.T)
        v, i = d2.min(1)
        pn[b:e] = i
        pd[b:e] = v.clamp_min(0).sqrt()
    return pn.cpu().numpy().astype(np.uint32), pd.cpu().numpy()


def run(x, dim, flat, mode, A):
    """one embedding as examples/mnist_fashion.rs runs it -> (quality report dict, seconds)"""
    n = len(x)
    par = A.EmbedderParams(asked_dim=dim, nb_grad_batch=25, scale_rho=1.0, beta=1.0, grad_step=1.0, nb_sampling_by_edge=10, dmap_init=True,
                           hubness_weighting=False, ce_mode=mode)   # mnist_fashion.rs:92-100
    t0 = time.perf_counter()
    if flat:
        kg = A.KGraph.bruteforce_l2(x, 12)  # :110-112
        emb = A.Embedder(kg, par)
    else:
        par.nb_grad_batch = 20               # :117
        par.grad_factor = 4                  # :119
        # the points are reordered so that the small graph's nodes come first (KGraphProjection numbers the upper layers first)
        rng = np.random.default_rng(16)
        small = np.sort(rng.choice(n, n // 16, replace=False))
        rest = np.setdiff1d(np.arange(n), small)
        order = np.concatenate([small, rest])
        xo = np.ascontiguousarray(x[order])
        large = A.KGraph.bruteforce_l2(xo, 6)            # :116
        smallg = A.KGraph.bruteforce_l2(xo[:len(small)], 6)
        pn, pd = nearest_small(xo, np.arange(len(small)))
        pn[:len(small)] = np.arange(len(small), dtype=np.uint32)
        pd[:len(small)] = 0.0
        proj = A.KGraphProjection(smallg, large, pn, pd)
        emb = A.Embedder.from_hkgraph(proj, par)
    t_graph = time.perf_counter() - t0
    t0 = time.perf_counter()
    rc = emb.embed()
    t_embed = time.perf_counter() - t0
    q = emb.get_quality_estimate_from_edge_length(50)
    return {"embed_rc": int(rc), "graph_s": t_graph, "embed_s": t_embed, "ce_mode_resolved": None, "cross_entropy": emb.get_cross_entropy(),
            "nb_without_match": int(q.nb_without_match), "mean_nbmatch": float(q.mean_nbmatch),
            "ratio_quantiles": [float(v) for v in q.ratio_quantiles], "radii_quantiles": [float(v) for v in q.radii_quantiles],
            "median_ratio": float(q.median_ratio)}


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    dim = int(sys.argv[sys.argv.index("--dim") + 1]) if "--dim" in sys.argv else 2
    if "--dim" in sys.argv:
        args = [a for a in args if a != str(dim)]
    flat = "--flat" in sys.argv
    from annembed_amd import io as aio
    d = aio.find_mnist_dir()
    if d is None:
        print(json.dumps({"skipped": "no IDX files: put train-/t10k- images-idx3-ubyte and labels-idx1-ubyte of Fashion-MNIST under "
                                     "$AE_MNIST_DIR, data/fashion-mnist/ or data/mnist/"}))
        return 0
    import annembed_amd as A
    x, _ = aio.mnist_images_as_vectors(d)
    out = {"data": d, "n": int(len(x)), "asked_dim": dim, "variant": "flat (knbn 12, 25 batches)" if flat else "hierarchical (knbn 6, 20 batches, grad_factor 4)",
           "published_hierarchical": PUBLISHED.get(dim), "published_at": "src/embedder.rs:585-618"}
    for name, mode in (("sequential", A.AE_CE_SEQUENTIAL), ("default", A.AE_CE_AUTO)):
        out[name] = run(x, dim, flat, mode, A)
    print(json.dumps(out))
    if args:
        with open(args[0], "w") as f:
            json.dump(out, f, indent=1)
    return 0


if __name__ == "__main__":
    sys.exit(main())
