#!/usr/bin/env python3
"""Generates tests/golden/golden_v2.npz -- run in the build container:  python tests/golden/make_golden_v2.py

Second, INDEPENDENT restatement (numpy, no code shared with oracle/annembed_oracle.c or oracle/oracle.py) of the stages
golden_v1 does not cover -- the hot function first:

  * ce_optim_edge_shannon, src/embedder.rs:1167-1302 (b = 1 and b != 1), 1 000 sequential samples drawn with the build's
    Philox4x32-10 stream (Salmon et al., SC'11; checked below against the Random123 known-answer vectors):
    `sgd_plan` (the 7 nodes of every sample), `sgd_y_after_b1`, `sgd_y_after_b08`;
  * the dense branch (n <= 5000) of the diffusion-map laplacian, src/diffmaps.rs:427-508, 855-892: `dense_q`,
    `dense_beta_scales`, `dense_lap`, its 20 leading singular values, and the diffusion-map coordinates `dense_y0`
    (src/diffmaps.rs:1213-1236; columns are defined up to sign: stored with a positive first non-zero entry);
  * the projection initialisation of h_embed, src/embedder.rs:245-269: `proj_y0`.

tests/test_oracle_golden.py checks the C oracle against these vectors; tests/test_gpu_parity.py checks the HIP path
(AE_CE_SEQUENTIAL, ae_projection_init, the dense laplacian) against them.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

F = np.float32
U32 = np.uint32
M32 = 0xFFFFFFFF


# ------------------------------------------------------------------------------------------------
# Philox4x32-10 on python ints; the build's stream convention: key = (seed lo, seed hi), counter =
# (sample lo, sample hi, batch, block), words consumed in order
# ------------------------------------------------------------------------------------------------
def philox(ctr, key):
    c0, c1, c2, c3 = ctr
    k0, k1 = key
    for _ in range(10):
        p0 = 0xD2511F53 * c0
        p1 = 0xCD9E8D57 * c2
        c0, c1, c2, c3 = ((p1 >> 32) ^ c1 ^ k0) & M32, p1 & M32, ((p0 >> 32) ^ c3 ^ k1) & M32, p0 & M32
        k0, k1 = (k0 + 0x9E3779B9) & M32, (k1 + 0xBB67AE85) & M32
    return [c0, c1, c2, c3]


class Stream:
    def __init__(self, seed, c01, c2):
        self.key = (seed & M32, (seed >> 32) & M32)
        self.c = (c01 & M32, (c01 >> 32) & M32, c2)
        self.blk, self.buf = 0, []

    def u32(self):
        if not self.buf:
            self.buf = philox([self.c[0], self.c[1], self.c[2], self.blk], self.key)
            self.blk += 1
        return self.buf.pop(0)

    def index(self, n):  # high 64 bits of a 64 x 64 product
        hi, lo = self.u32(), self.u32()
        return (((hi << 32) | lo) * n) >> 64

    def f32(self):
        return F(self.u32() >> 8) * F(1.0 / 16777216.0)


def sample_nodes(st, indptr, nbr, proba, n):
    """the node set of one sample: source uniform, edge by the row's running f32 cdf (the ROWCDF sampler: same law as the alias
    table of :987 because every row sums to 1), five admissible negatives (:1241-1253)"""
    i = st.index(n)
    u = st.f32()
    b, e = int(indptr[i]), int(indptr[i + 1])
    acc, m = F(0), e - b - 1
    for t in range(e - b):
        acc = F(acc + proba[b + t])
        if u < acc:
            m = t
            break
    j = int(nbr[b + m])
    row = set(int(v) for v in nbr[b:e])
    ks = []
    while len(ks) < 5:
        k = st.index(n)
        if k == i or k == j or k in row:
            continue
        ks.append(k)
    return i, j, ks, proba[b + m]


def sgd_sample(y, emb_scale, i, j, ks, w, step, b):
    """embedder.rs:1167-1302 on numpy rows: coordinates f32, scalars f64"""
    yi, yj = y[i].copy(), y[j].copy()
    grad = np.zeros_like(yi)
    scale = np.float64(emb_scale[i])
    df = (yi - yj).astype(F)
    acc = F(0)
    for t in range(len(df)):
        acc = F(acc + F(df[t] * df[t]))
    ds = np.float64(acc) / (scale * scale)
    if b != 1.0:
        cw = 1.0 / (1.0 + ds ** b) if ds > 0 else 1.0
        coeff = 2.0 * b * cw * (ds ** (b - 1.0) if ds > 0 else (np.inf if b < 1 else 0.0)) / (scale * scale)
    else:
        coeff = 2.0 * b * (1.0 / (1.0 + ds)) / (scale * scale)
    if ds > 0:
        rep = 1.0 / max(ds * ds, np.float64(F(1.0) / F(1.0e-4)))
        cij = max(step * coeff * (-np.float64(w) + (1.0 - np.float64(w)) * rep), -0.49)
        grad = ((yj - yi) * F(cij)).astype(F)
    yi = (yi - grad).astype(F)
    yj = (yj + grad).astype(F)
    y[j] = yj
    for k in ks:
        yk = y[k].copy()
        dk = (yi - yk).astype(F)
        acc = F(0)
        for t in range(len(dk)):
            acc = F(acc + F(dk[t] * dk[t]))
        dks = np.float64(acc) / (scale * scale)
        if b != 1.0:
            cw = 1.0 / (1.0 + dks ** b) if dks > 0 else 1.0
            cf2 = 2.0 * b * cw * (dks ** (b - 1.0) if dks > 0 else 0.0) / (scale * scale)
        else:
            cf2 = 2.0 * b * (1.0 / (1.0 + dks)) / (scale * scale)
        if acc > 0:
            cik = min(step * cf2 * (1.0 / max(dks * dks, 1.0 / 16.0)), 2.0)
            grad = ((yk - yi) * F(cik)).astype(F)
        yi = (yi - grad).astype(F)
    y[i] = yi


def run_sgd(g1, nb, step, it, b, seed=4664397):
    indptr, nbr, proba, es, y0 = g1["g_indptr"], g1["g_nbr"], g1["proba"], g1["emb_scale"], g1["y_box"]
    n = len(indptr) - 1
    y = y0.astype(F).copy()
    plan = np.zeros((nb, 7), U32)
    for s in range(nb):
        st = Stream(seed, s, it)
        i, j, ks, w = sample_nodes(st, indptr, nbr, proba, n)
        plan[s] = [i, j] + ks
        sgd_sample(y, es, i, j, ks, w, step, b)
    return plan, y


# ------------------------------------------------------------------------------------------------
# dense-branch diffusion map
# ------------------------------------------------------------------------------------------------
def seqsum(a):
    s = F(0)
    for v in np.asarray(a, F):
        s = F(s + v)
    return s


def ndarray_sum(a):
    """ndarray's Array1::sum for a contiguous f32 array: eight interleaved partial sums (numeric_util::unrolled_fold), the
    partials combined pairwise ((p0+p4)+(p1+p5))+((p2+p6)+(p3+p7)), then the tail left to right"""
    a = np.asarray(a, F)
    p = [F(0)] * 8
    nb = len(a) // 8
    for blk in range(nb):
        for t in range(8):
            p[t] = F(p[t] + a[8 * blk + t])
    acc = F(F(F(p[0] + p[4]) + F(p[1] + p[5])) + F(F(p[2] + p[6]) + F(p[3] + p[7])))
    for v in a[8 * nb:]:
        acc = F(acc + v)
    return acc


def dense_dmap(indptr, nbr, dist, max_nbng, nbng, alfa, beta, epsil):
    n = len(indptr) - 1
    local = np.zeros(n, F)
    for i in range(n):
        b, e = int(indptr[i]), int(indptr[i + 1])
        d = dist[b:min(e, b + nbng)]
        local[i] = np.sqrt(F(seqsum((d * d).astype(F)) / F(e - b)))        # :1032-1039
    mean = F(seqsum(local) / F(n))                                              # :801
    local = np.where(local <= 0, mean, local).astype(F)                        # :806-810
    normed = (local / mean).astype(F)                                          # :815-822

    def kernel(scales):                                                        # :590-675
        P = np.zeros((n, n), F)
        for i in range(n):
            b, e = int(indptr[i]), int(indptr[i + 1])
            d = dist[b:e]
            pos = np.nonzero(d > 0)[0]
            if len(pos) and d[pos[-1]] > d[0]:
                ls = np.sqrt((scales[nbr[b:e]] * scales[i]).astype(F)).astype(F)
                x = (d / (np.sqrt(F(epsil)) * ls)).astype(F)
                P[i, nbr[b:e]] = np.maximum(np.exp(-(x * x).astype(F)).astype(F), F(1.0e-4))
                P[i, i] = F(1.0)
            else:
                P[i, nbr[b:e]] = F(1.0) / F(e - b + 1)
                P[i, i] = F(1.0) / F(e - b + 1)
        return P

    def rowsum(S):
        return np.array([ndarray_sum(S[i]) for i in range(n)], F)

    P1 = kernel(local)
    S1 = ((P1 + P1.T).astype(F) * F(0.5)).astype(F)                            # :865-887
    q = rowsum(S1)
    q = (q / F(max_nbng)).astype(F)                                            # :888-890
    q = (q / F(ndarray_sum(q) / F(n))).astype(F)                               # :891
    beta_scales = (np.power(q, F(beta)).astype(F) * mean).astype(F)            # :938-942
    P2 = kernel(beta_scales)
    S = ((P2 + P2.T).astype(F) * F(0.5)).astype(F)                             # :447-460
    q2 = rowsum(S)
    q2 = (q2 / F(ndarray_sum(q2) / F(max_nbng))).astype(F)                     # :469-470
    S = (S / np.power((q2[:, None] * q2[None, :]).astype(F), F(alfa)).astype(F)).astype(F)  # :476
    deg = rowsum(S)                                                            # :478
    sw = np.sqrt(deg).astype(F)
    S = (S / (sw[:, None] * sw[None, :]).astype(F)).astype(F)                  # :482-487
    return dict(q=q, beta_scales=beta_scales, lap=S, normalizer=sw, normed=normed)


def embed_from_svd(s, u, normalizer, normed, asked_dim, t):
    nl = (s / s[0]).astype(F)                                                  # :1213
    sum_diag = F(seqsum(normalizer) / F(len(normalizer)))                      # :1223
    w = (normed * np.sqrt((normalizer / sum_diag).astype(F)).astype(F)).astype(F)  # :1228
    y = np.zeros((len(normed), asked_dim), F)
    for c in range(asked_dim):
        y[:, c] = np.clip((np.power(nl[c + 1], F(t)).astype(F) * u[:, c + 1] / w).astype(F), F(-10), F(10))  # :1232
    return y


def sign_fix(y):
    y = y.copy()
    for c in range(y.shape[1]):
        nz = np.nonzero(y[:, c])[0]
        if len(nz) and y[nz[0], c] < 0:
            y[:, c] = -y[:, c]
    return y


# ------------------------------------------------------------------------------------------------
# projection init
# ------------------------------------------------------------------------------------------------
def gaussian_fill(count, seed, tag):
    out = np.zeros(count, F)
    key = (seed & M32, (seed >> 32) & M32)
    for blk in range((count + 3) // 4):
        w = philox([blk & M32, (blk >> 32) & M32, tag, 0], key)
        z = []
        for a, b in ((w[0], w[1]), (w[2], w[3])):
            u1 = F((a >> 8) + 1) * F(1.0 / 16777216.0)
            u2 = F(b >> 8) * F(1.0 / 16777216.0)
            r = np.sqrt(F(-2.0) * np.log(u1, dtype=F), dtype=F)
            ang = F(6.28318530717958647692) * u2
            z += [F(r * np.cos(ang, dtype=F)), F(r * np.sin(ang, dtype=F))]
        for t in range(4):
            if 4 * blk + t < count:
                out[4 * blk + t] = z[t]
    return out


def projection_init(y_small, n_large, proj_node, proj_dist, median, seed):
    n_small, dim = y_small.shape
    z = gaussian_fill(n_large * dim, seed, 0xFFFF0002).reshape(n_large, dim)
    y0 = np.zeros((n_large, dim), F)
    y0[:n_small] = y_small
    for i in range(n_small, n_large):
        corr = np.sqrt(F(F(proj_dist[i] / median) / F(dim)))                   # :262-263
        y0[i] = y_small[proj_node[i]] + np.clip((corr * z[i]).astype(F), F(-2), F(2))  # :265-267
    return y0


def main():
    # Random123 known-answer vectors for Philox4x32-10
    assert philox([0, 0, 0, 0], (0, 0)) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert philox([M32] * 4, (M32, M32)) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], (0xa4093822, 0x299f31d0)) == [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]
    g1 = np.load(os.path.join(HERE, "golden_v1.npz"))
    out = {}
    plan, y1 = run_sgd(g1, 1000, 0.9, 3, 1.0)
    out.update(sgd_plan=plan, sgd_y_after_b1=y1, sgd_step=np.array(0.9), sgd_iter=np.array(3))
    _, y2 = run_sgd(g1, 1000, 0.9, 3, 0.8)
    out["sgd_y_after_b08"] = y2
    # a graph of its own for the spectral stages: three clusters in a chain (150 / 100 / 50 points, 3-D, separations 2.5 and 4),
    # connected, whose two slow modes are well apart from each other and from the rest (gaps sigma_k - sigma_k+1 of
    # 4e-3, 1.2e-2, 2e-2 for k = 1, 2, 3) -- the single-blob graph of golden_v1 has gaps of 1e-3, too small to compare
    # singular VECTORS of an f32 matrix at 1e-4
    from tests.util import knn_graph
    rg = np.random.default_rng(21)
    xg = np.concatenate([rg.normal(0, 1, (150, 3)), rg.normal(0, 1, (100, 3)) + [2.5, 0, 0], rg.normal(0, 1, (50, 3)) + [6.5, 0, 0]]).astype(F)
    gi, gn, gd = knn_graph(xg, 6)
    out.update(gap_indptr=gi, gap_nbr=gn, gap_dist=gd)
    d = dense_dmap(gi, gn, gd, 6, 6, 0.5, -0.1, 2.0)
    out.update(dense_q=d["q"], dense_beta_scales=d["beta_scales"], dense_lap=d["lap"], dense_normalizer=d["normalizer"])
    u, s, _ = np.linalg.svd(d["lap"].astype(np.float64))
    out["dense_sigma"] = s[:20]
    y0 = embed_from_svd(s.astype(F), u.astype(F), d["normalizer"], d["normed"], 2, 5.0)
    out["dense_y0"] = sign_fix(y0)
    out["dense_gap"] = np.array([s[0] - s[1], s[1] - s[2], s[2] - s[3]])
    rng = np.random.default_rng(9)
    n_small, n_large = 40, 300
    pn = rng.integers(0, n_small, n_large).astype(U32)
    pd = rng.gamma(2.0, 1.0, n_large).astype(F)
    pn[:n_small] = np.arange(n_small)
    pd[:n_small] = 0
    ys = rng.normal(size=(n_small, 3)).astype(F)
    med = np.sort(pd[n_small:])[(n_large - n_small - 1) // 2]
    out.update(proj_node=pn, proj_dist=pd, proj_y_small=ys, proj_median=np.array(med, F), proj_y0=projection_init(ys, n_large, pn, pd, F(med), 4664397))
    np.savez_compressed(os.path.join(HERE, "golden_v2.npz"), **out)
    print("wrote golden_v2.npz:", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
