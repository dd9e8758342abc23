#!/usr/bin/env python3
"""Generates tests/golden/*.npz -- run in the build container:  python tests/golden/make_golden.py

The reference (Rust) cannot be built or imported here (no cargo/rustc, no Python implementation),
so the fixtures come from an INDEPENDENT numpy restatement of the same reference lines written in
this file (vectorised, no shared code with oracle/annembed_oracle.c).  tests/test_oracle_golden.py
checks the C oracle against these vectors: two independent restatements of the cited lines agreeing
is the pin for the stages the reference holds no numeric test for.  The known-answer
vectors of the reference's own tests (wiki matrix, sigma_1 = 10.6811457) are stored too.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from tests.util import synthetic_graph  # noqa: E402

F = np.float32
PROBA_MIN = F(1.0e-4)


def seqsum(a):
    """left-to-right f32 sum (Rust iter().sum::<f32>())"""
    s = F(0)
    for v in np.asarray(a, F):
        s = F(s + v)
    return s


def np_to_proba_edges(indptr, nbr, dist, scale_rho, beta):
    """src/tools/kdumap.rs:132-235, row by row with numpy ops"""
    n = len(indptr) - 1
    proba = np.zeros(len(nbr), F)
    scale = np.zeros(n, F)
    first = dist[indptr[:-1].astype(np.int64)]
    for i in range(n):
        b, e = int(indptr[i]), int(indptr[i + 1])
        d = dist[b:e]
        rho = np.concatenate([first[nbr[b:e]], [d[0]]]).astype(F)  # :149-154
        mean_rho = F(seqsum(rho) / F(len(rho)))  # :155
        sc = F(F(scale_rho) * mean_rho)  # :159
        scale[i] = sc
        pos = np.nonzero(d > 0)[0]
        if len(pos) and d[pos[-1]] > d[0]:  # :164-178
            w = np.exp(-np.power(np.maximum(d - d[0], F(0)) / sc, F(beta), dtype=F), dtype=F)  # :172-174
            w = np.maximum(w, PROBA_MIN)  # :185
            assert w[-1] / w[0] >= PROBA_MIN  # :209
            proba[b:e] = w / seqsum(w)  # :215-218
        else:
            proba[b:e] = F(1.0) / F(e - b)  # :224-230
    return proba, scale


def np_embedded_scales(scale):
    """src/embedder.rs:1356-1366"""
    mean = F(seqsum(scale) / F(len(scale)))
    return (F(0.2) * np.maximum(np.minimum(scale / mean, F(4.0)), F(0.25))).astype(F)


def np_set_data_box(y, box):
    """src/embedder.rs:1376-1408"""
    y = y.astype(F).copy()
    for j in range(y.shape[1]):
        y[:, j] -= F(seqsum(y[:, j]) / F(y.shape[0]))
    mm = F(np.abs(y).max() / F(F(box) / F(2)))
    return (y / mm).astype(F)


def np_ce(indptr, nbr, proba, emb_scale, y, b=1.0):
    """src/embedder.rs:1127-1163, 1322-1345"""
    n = len(indptr) - 1
    src = np.repeat(np.arange(n), np.diff(indptr.astype(np.int64)))
    diff = (y[src] - y[nbr]).astype(F)
    acc = np.zeros(len(nbr), F)
    for c in range(y.shape[1]):  # f32 left-to-right accumulation
        acc = (acc + diff[:, c] * diff[:, c]).astype(F)
    sc = emb_scale[src].astype(np.float64)
    d = acc.astype(np.float64) / (sc * sc)
    d = np.power(d, b)
    w = (1.0 / (1.0 + d)).astype(F)
    w = np.where(w < F(1), w, F(1) - np.finfo(F).eps).astype(np.float64)
    wij = proba.astype(np.float64)
    t = np.where(w > 0, -wij * np.log(np.where(w > 0, w, 1.0)), 0.0) + np.where(w < 1, -(1 - wij) * np.log1p(-w), 0.0)
    return float(np.sum(t))


def np_dmap(indptr, nbr, dist, max_nbng, nbng, alfa, beta, epsil):
    """src/diffmaps.rs:752-849 (scales + kernel), :898-942 (density, CSR branch), :513-584 (laplacian, CSR
    branch) with scipy.sparse: S = elementwise max(P, P^T), multiplicity 2 for mutual/diagonal entries."""
    import scipy.sparse as sp
    n = len(indptr) - 1
    lens = np.diff(indptr.astype(np.int64))
    src = np.repeat(np.arange(n), lens)
    # local scales :1020-1043, :801-822
    pos_in_row = np.arange(len(nbr)) - np.repeat(indptr[:-1].astype(np.int64), lens)
    local = np.zeros(n, F)
    for i in range(n):
        b, e = int(indptr[i]), int(indptr[i + 1])
        d = dist[b:min(e, b + nbng)]
        local[i] = np.sqrt(F(seqsum(d * d) / F(e - b)))
    mean = F(seqsum(local) / F(n))
    local = np.where(local <= 0, mean, local).astype(F)
    normed = (local / mean).astype(F)

    def kernel(scales):  # :590-675, :831-834
        ls = np.sqrt(scales[nbr] * scales[src], dtype=F)
        x = (dist / (np.sqrt(F(epsil)) * ls)).astype(F)
        w = np.exp(-(x * x), dtype=F)
        w = np.maximum(w, PROBA_MIN)
        kself = np.ones(n, F)
        for i in range(n):  # all-equal rows :618-647
            b, e = int(indptr[i]), int(indptr[i + 1])
            d = dist[b:e]
            p = np.nonzero(d > 0)[0]
            if not (len(p) and d[p[-1]] > d[0]):
                w[b:e] = F(1.0) / F(e - b + 1)
                kself[i] = F(1.0) / F(e - b + 1)
        P = sp.csr_matrix((w, nbr.astype(np.int64), indptr.astype(np.int64)), shape=(n, n)) + sp.diags(kself).tocsr()
        return P.tocsr().astype(F)

    def symmetrise(P):  # triplet list semantics of :527-544 summed by TriMat::to_csr
        Pt = P.T.tocsr()
        smax = P.maximum(Pt).tocsr()
        mult = ((P != 0).astype(np.int32) + (Pt != 0).astype(np.int32)).tocsr()
        return smax, mult

    P1 = kernel(local)
    smax, mult = symmetrise(P1)
    q = np.asarray(smax.multiply(mult).sum(axis=1)).ravel().astype(F)  # :923,:928
    q = (q / F(max_nbng)).astype(F)  # :931
    q = (q / F(q.sum(dtype=np.float64) / n)).astype(F)  # :932-933 (f64 sum: order-free reference value)
    beta_scales = (np.power(q, F(beta), dtype=F) * mean).astype(F)  # :938-942
    P2 = kernel(beta_scales)
    smax, mult = symmetrise(P2)
    T = smax.multiply(mult).tocsr()
    q2 = np.asarray(T.sum(axis=1)).ravel().astype(F)
    q2 = (q2 / F(q2.sum(dtype=np.float64) / max_nbng)).astype(F)  # :546-548
    coo = smax.tocoo()
    v = (coo.data / np.power(q2[coo.row] * q2[coo.col], F(alfa), dtype=F)).astype(F)  # :553-557
    m = np.asarray(mult.tocsr()[coo.row, coo.col]).ravel().astype(F)
    deg = np.zeros(n, np.float64)
    np.add.at(deg, coo.row, (v * m).astype(np.float64))  # :561-564
    sw = np.sqrt(deg.astype(F))
    v = (v / (sw[coo.row] * sw[coo.col])).astype(F)  # :566-570
    A = sp.csr_matrix(((v * m).astype(F), (coo.row, coo.col)), shape=(n, n))
    A.sort_indices()
    return dict(local=local, normed=normed, mean_scale=mean, q=q, beta_scales=beta_scales, normalizer=sw.astype(F),
                lap_indptr=A.indptr.astype(np.uint64), lap_indices=A.indices.astype(np.uint32), lap_values=A.data.astype(F))


def main():
    out = {}
    # ---- known-answer vectors held by the reference's own tests ----
    out["wiki"] = np.array([[1, 0, 0, 0, 2], [0, 0, 3, 0, 0], [0, 0, 0, 0, 0], [0, 2, 0, 0, 0]], np.float64)  # svdapprox.rs:1316
    out["wiki_sigma"] = np.array([3.0, np.sqrt(5.0), 2.0, 0.0])  # svdapprox.rs:1335
    out["spectral_mat"] = np.array([[9., -1., 2.], [-2., 8., 4.], [1., 1., 8.]])  # svdapprox.rs:1038
    out["spectral_sigma1"] = np.array(10.6811457)  # svdapprox.rs:1042
    # ---- a fixed 300-node / k=6 graph, connected (one blob) ----
    indptr, nbr, dist, x, _ = synthetic_graph(n=300, dim=6, k=6, seed=11, ncomp=1)
    # make a few degenerate rows: all-equal distances, zero distances (Higgs-like duplicates, kdumap.rs:163)
    dist = dist.copy()
    dist[indptr[5]:indptr[6]] = dist[indptr[5]]
    dist[indptr[9]:indptr[10]] = 0.0
    out.update(g_indptr=indptr, g_nbr=nbr, g_dist=dist)
    proba, scale = np_to_proba_edges(indptr, nbr, dist, 1.0, 1.0)
    out.update(proba=proba, scale=scale)
    proba2, scale2 = np_to_proba_edges(indptr, nbr, dist, 0.75, 2.0)
    out.update(proba_rho075_beta2=proba2, scale_rho075_beta2=scale2)
    es = np_embedded_scales(scale)
    out["emb_scale"] = es
    rng = np.random.default_rng(5)
    y = np_set_data_box(rng.normal(size=(300, 2)), 10.0)
    out["y_box"] = y
    out["y_raw"] = rng.normal(size=(300, 3)).astype(F)
    out["y_raw_box"] = np_set_data_box(out["y_raw"], 10.0)
    out["ce_value"] = np.array(np_ce(indptr, nbr, proba, es, y))
    d = np_dmap(indptr, nbr, dist, 6, 6, 0.5, -0.1, 2.0)
    out.update({"dmap_" + k: v for k, v in d.items()})
    np.savez_compressed(os.path.join(HERE, "golden_v1.npz"), **out)
    print("wrote golden_v1.npz with", len(out), "arrays")


if __name__ == "__main__":
    main()
