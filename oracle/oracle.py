"""
oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Python face of the CPU oracle: ctypes bindings to oracle/liboracle.so (the plain-C restatement in
annembed_oracle.c) plus the LAPACK-backed parts of the reference path restated with numpy/scipy:

  * randomized SVD  src/tools/svdapprox.rs:285-408, 721-799  (scipy.linalg.qr = geqrf+orgqr,
    scipy.linalg.svd(lapack_driver="gesdd") = the routines the reference reaches through `lax`)
  * full SVD        src/graphlaplace.rs:296-344
  * dense-branch laplacian / density  src/diffmaps.rs:445-508, 865-892
  * drivers         src/diffmaps.rs:397-422, 1047-1075, 1145-1243; src/embedder.rs:194-371, 794-904

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
Parity pinning: see the header of annembed_oracle.c and DESIGN.md.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

FULL_MAT_REPR = 5000        # src/graphlaplace.rs:13
FULL_SVD_SIZE_LIMIT = 5000  # src/graphlaplace.rs:15
OMEGA_SEED = 4664397        # src/tools/svdapprox.rs:70
TAG_OMEGA = 0xFFFF0001

u64p = np.ctypeslib.ndpointer(np.uint64, flags="C")
u32p = np.ctypeslib.ndpointer(np.uint32, flags="C")
f32p = np.ctypeslib.ndpointer(np.float32, flags="C")


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
        _LIB.orc_ce_compute.restype = C.c_double
    return _LIB


class OrcCE(C.Structure):
    _fields_ = [
        ("n", C.c_uint64), ("dim", C.c_uint64), ("nnz", C.c_uint64),
        ("indptr", C.c_void_p), ("nbr", C.c_void_p), ("proba", C.c_void_p),
        ("emb_scale", C.c_void_p), ("y", C.c_void_p),
        ("b", C.c_double), ("seed", C.c_uint64), ("sampler", C.c_int),
        ("node_lo", C.c_uint64), ("node_hi", C.c_uint64),
        ("edge_odds", C.c_void_p), ("edge_alias", C.c_void_p), ("edge_src", C.c_void_p),
        ("hub_odds", C.c_void_p), ("hub_alias", C.c_void_p),
    ]


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


# ------------------------------------------------------------------------------------------------
# thin wrappers over the C functions
# ------------------------------------------------------------------------------------------------
def philox(ctr, key):
    out = np.zeros(4, np.uint32)
    lib().orc_philox(_p(np.asarray(ctr, np.uint32)), _p(np.asarray(key, np.uint32)), _p(out))
    return out


def gaussian_fill(count, seed, tag=TAG_OMEGA):
    out = np.zeros(count, np.float32)
    lib().orc_gaussian_fill(_p(out), C.c_uint64(count), C.c_uint64(seed), C.c_uint32(tag))
    return out


def kgraph_from_ragged(point_id, row_ptr, nbr_data_id, nbr_dist, nbng):
    n = len(point_id)
    point_id = np.ascontiguousarray(point_id, np.uint64)
    row_ptr = np.ascontiguousarray(row_ptr, np.uint64)
    nbr_data_id = np.ascontiguousarray(nbr_data_id, np.uint64)
    nbr_dist = np.ascontiguousarray(nbr_dist, np.float32)
    indptr = np.zeros(n + 1, np.uint64)
    nbr = np.zeros(n * nbng, np.uint32)
    dist = np.zeros(n * nbng, np.float32)
    ids = np.zeros(n, np.uint64)
    rc = lib().orc_kgraph_from_ragged(_p(point_id), _p(row_ptr), _p(nbr_data_id), _p(nbr_dist), C.c_uint64(n),
                                      C.c_uint32(nbng), _p(indptr), _p(nbr), _p(dist), _p(ids))
    if rc:
        return rc, None
    nnz = int(indptr[-1])
    return 0, (indptr, nbr[:nnz].copy(), dist[:nnz].copy(), ids)


def knn_bruteforce_l2(x, k):
    """Exact L2 kNN graph as the build's producer defines it (no reference counterpart: the reference takes its graph
    from hnsw_rs, kgraph.rs:496-546): F(i, j) = f32 sum over the coordinates IN ORDER of (x_i[t] - x_j[t])^2,
    row i = the k points j != i with the smallest (F, j), ascending, distance sqrt(F) in f32.
    Returns CSR (indptr u64, nbr u32, dist f32)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    n, dim = x.shape
    nbr = np.zeros((n, k), np.uint32)
    dist = np.zeros((n, k), np.float32)
    bs = max(1, min(n, (1 << 24) // max(n, 1)))
    for b in range(0, n, bs):
        e = min(n, b + bs)
        f = np.zeros((e - b, n), np.float32)
        for t in range(dim):  # sequential f32 accumulation over the coordinates
            df = x[b:e, t][:, None] - x[None, :, t]
            f += df * df
        f[np.arange(e - b), np.arange(b, e)] = np.inf
        order = np.argsort(f, axis=1, kind="stable")[:, :k]  # stable: the smaller index wins among equal F
        nbr[b:e] = order
        dist[b:e] = np.sqrt(np.take_along_axis(f, order, 1))
    indptr = np.arange(n + 1, dtype=np.uint64) * np.uint64(k)
    return indptr, nbr.reshape(-1), dist.reshape(-1)


def hubness(indptr, nbr):
    n = len(indptr) - 1
    counts = np.zeros(n, np.uint32)
    lib().orc_hubness(C.c_uint64(n), _p(indptr), _p(nbr), _p(counts))
    return counts


def to_proba_edges(indptr, nbr, dist, scale_rho, beta):
    n = len(indptr) - 1
    proba = np.zeros(len(nbr), np.float32)
    scale = np.zeros(n, np.float32)
    rc = lib().orc_to_proba_edges(C.c_uint64(n), _p(indptr), _p(nbr), _p(dist), C.c_float(scale_rho),
                                  C.c_float(beta), _p(proba), _p(scale))
    return rc, proba, scale


def perplexity(indptr, proba):
    n = len(indptr) - 1
    out = np.zeros(n, np.float32)
    lib().orc_perplexity(C.c_uint64(n), _p(indptr), _p(proba), _p(out))
    return out


def dmap_local_scales(indptr, dist, nbng):
    n = len(indptr) - 1
    ls = np.zeros(n, np.float32)
    normed = np.zeros(n, np.float32)
    mean = C.c_float(0)
    rc = lib().orc_dmap_local_scales(C.c_uint64(n), _p(indptr), _p(dist), C.c_uint32(nbng), _p(ls), _p(normed),
                                     C.byref(mean))
    assert rc == 0
    return ls, normed, np.float32(mean.value)


def dmap_kernel(indptr, nbr, dist, scales, epsil):
    n = len(indptr) - 1
    nnz = len(nbr)
    kindptr = np.zeros(n + 1, np.uint64)
    kcols = np.zeros(nnz + n, np.uint32)
    kvals = np.zeros(nnz + n, np.float32)
    low = C.c_uint64(0)
    rc = lib().orc_dmap_kernel(C.c_uint64(n), _p(indptr), _p(nbr), _p(dist), _p(np.ascontiguousarray(scales, np.float32)),
                               C.c_float(epsil), _p(kindptr), _p(kcols), _p(kvals), C.byref(low))
    return rc, kindptr, kcols, kvals, low.value


def dmap_density_csr(kindptr, kcols, kvals, max_nbng, beta, mean_scale):
    n = len(kindptr) - 1
    q = np.zeros(n, np.float32)
    bs = np.zeros(n, np.float32)
    lib().orc_dmap_density_csr(C.c_uint64(n), _p(kindptr), _p(kcols), _p(kvals), C.c_uint32(max_nbng),
                               C.c_float(beta), C.c_float(mean_scale), _p(q), _p(bs))
    return q, bs


def dmap_laplacian_csr(kindptr, kcols, kvals, max_nbng, alfa):
    n = len(kindptr) - 1
    cap = 2 * int(kindptr[-1])
    lindptr = np.zeros(n + 1, np.uint64)
    lcols = np.zeros(cap, np.uint32)
    lvals = np.zeros(cap, np.float32)
    norm = np.zeros(n, np.float32)
    lnnz = C.c_uint64(0)
    lib().orc_dmap_laplacian_csr(C.c_uint64(n), _p(kindptr), _p(kcols), _p(kvals), C.c_uint32(max_nbng),
                                 C.c_float(alfa), _p(lindptr), _p(lcols), _p(lvals), C.byref(lnnz), _p(norm))
    k = lnnz.value
    return lindptr, lcols[:k].copy(), lvals[:k].copy(), norm


def embed_from_svd(s, u, normalizer, normed_scales, asked_dim, t=None):
    n, r = u.shape
    y0 = np.zeros((n, min(asked_dim, r - 1)), np.float32)
    rd = C.c_uint64(0)
    rc = lib().orc_embed_from_svd(C.c_uint64(n), C.c_uint64(r), _p(np.ascontiguousarray(s, np.float32)),
                                  _p(np.ascontiguousarray(u, np.float32)), _p(normalizer), _p(normed_scales),
                                  C.c_uint64(asked_dim), C.c_float(0.0 if t is None else t),
                                  C.c_int(0 if t is None else 1), _p(y0), C.byref(rd))
    return rc, y0


def set_data_box(y, box_size=10.0):
    y = np.ascontiguousarray(y, np.float32).copy()
    lib().orc_set_data_box(_p(y), C.c_uint64(y.shape[0]), C.c_uint64(y.shape[1]), C.c_float(box_size))
    return y


def embedded_scales(scale):
    out = np.zeros_like(scale)
    lib().orc_embedded_scales(_p(scale), C.c_uint64(len(scale)), _p(out))
    return out


def alias_build(w):
    w = np.ascontiguousarray(w, np.float32)
    odds = np.zeros(len(w), np.float32)
    alias = np.zeros(len(w), np.uint32)
    lib().orc_alias_build(_p(w), C.c_uint64(len(w)), _p(odds), _p(alias))
    return odds, alias


def node_sampler_weights(counts):
    """NodeSampler weights from hubness counts, src/embedder.rs:826-833, 915-919."""
    upper = np.float32(len(counts))
    f = np.minimum(np.maximum(counts.astype(np.float32), np.float32(1.0)), upper)
    s = np.float32(0)
    for x in f:  # sequential f32 sum as iter().sum::<f32>()
        s = np.float32(s + x)
    mean = np.float32(s / np.float32(len(f)))
    return (f / mean).astype(np.float32)


def projection_init(y_small, n_large, proj_node, proj_dist, median_dist, seed):
    n_small, dim = y_small.shape
    y0 = np.zeros((n_large, dim), np.float32)
    lib().orc_projection_init(_p(np.ascontiguousarray(y_small, np.float32)), C.c_uint64(n_small), C.c_uint64(n_large),
                              C.c_uint64(dim), _p(np.ascontiguousarray(proj_node, np.uint32)),
                              _p(np.ascontiguousarray(proj_dist, np.float32)), C.c_float(median_dist),
                              C.c_uint64(seed), _p(y0))
    return y0


def random_init(n, dim, size, seed):
    y = np.zeros((n, dim), np.float32)
    lib().orc_random_init(_p(y), C.c_uint64(n), C.c_uint64(dim), C.c_float(size), C.c_uint64(seed))
    return y


class EntropyOptim:
    """EntropyOptim, src/embedder.rs:936-1315, on CSR NodeParams."""

    def __init__(self, indptr, nbr, proba, scale, y0, b=1.0, seed=OMEGA_SEED, sampler=0, hub_counts=None,
                 node_lo=0, node_hi=None):
        self.indptr = np.ascontiguousarray(indptr, np.uint64)
        self.nbr = np.ascontiguousarray(nbr, np.uint32)
        self.proba = np.ascontiguousarray(proba, np.float32)
        self.n = len(indptr) - 1
        self.y = np.ascontiguousarray(y0, np.float32).copy()
        self.dim = self.y.shape[1]
        self.emb_scale = embedded_scales(np.ascontiguousarray(scale, np.float32))
        node_hi = self.n if node_hi is None else node_hi
        self._keep = []
        c = OrcCE()
        c.n, c.dim, c.nnz = self.n, self.dim, len(nbr)
        c.indptr, c.nbr, c.proba = _p(self.indptr), _p(self.nbr), _p(self.proba)
        c.emb_scale, c.y = _p(self.emb_scale), _p(self.y)
        c.b, c.seed, c.sampler = b, seed, sampler
        c.node_lo, c.node_hi = node_lo, node_hi
        if sampler == 1:
            e0, e1 = int(self.indptr[node_lo]), int(self.indptr[node_hi])
            odds, alias = alias_build(self.proba[e0:e1])
            src = np.repeat(np.arange(self.n, dtype=np.uint32), np.diff(self.indptr).astype(np.int64))[e0:e1].copy()
            self._keep += [odds, alias, src]
            c.edge_odds, c.edge_alias, c.edge_src = _p(odds), _p(alias), _p(src)
        if hub_counts is not None:
            hodds, halias = alias_build(node_sampler_weights(np.asarray(hub_counts)))
            self._keep += [hodds, halias]
            c.hub_odds, c.hub_alias = _p(hodds), _p(halias)
        self.c = c

    def plan(self, s, it):
        nodes = np.zeros(7, np.uint32)
        w = C.c_float(0)
        rc = lib().orc_ce_plan(C.byref(self.c), C.c_uint64(s), C.c_uint32(it), _p(nodes), C.byref(w))
        assert rc == 0
        return nodes, w.value

    def ce(self):
        return lib().orc_ce_compute(C.byref(self.c))

    def gradient_iteration(self, nb_sample, grad_step, it, s_begin=0):
        rc = lib().orc_gradient_iteration(C.byref(self.c), C.c_uint64(s_begin), C.c_uint64(nb_sample),
                                          C.c_double(grad_step), C.c_uint32(it))
        assert rc == 0, rc

    def gradient_iteration_hogwild(self, nb_sample, grad_step, it, nthreads=0):
        rc = lib().orc_gradient_iteration_hogwild(C.byref(self.c), C.c_uint64(nb_sample), C.c_double(grad_step),
                                                  C.c_uint32(it), C.c_int(nthreads))
        assert rc == 0, rc


def max_threads():
    return lib().orc_max_threads()


def entropy_optimize(indptr, nbr, proba, scale, y0, nb_grad_batch, nb_sampling_by_edge=10, grad_step=2.0, b=1.0,
                     seed=OMEGA_SEED, sampler=0, hub_counts=None, hogwild_threads=None):
    """entropy_optimize, src/embedder.rs:794-904."""
    eo = EntropyOptim(indptr, nbr, proba, scale, y0, b=b, seed=seed, sampler=sampler, hub_counts=hub_counts)
    ce0 = eo.ce()
    nb_sample = nb_sampling_by_edge * len(nbr)  # :858
    for it in range(1, nb_grad_batch + 1):      # :873
        step = grad_step * (1.0 - it / nb_grad_batch)  # :875
        if hogwild_threads is None:
            eo.gradient_iteration(nb_sample, step, it)
        else:
            eo.gradient_iteration_hogwild(nb_sample, step, it, hogwild_threads)
    return eo.y, ce0, eo.ce()


# ------------------------------------------------------------------------------------------------
# randomized SVD (numpy / scipy-LAPACK restatement)
# ------------------------------------------------------------------------------------------------
def _do_qr(y):
    """do_qr, svdapprox.rs:998-1013: Householder QR (geqrf + orgqr), thin Q, same dtype."""
    import scipy.linalg as sla
    q, _ = sla.qr(y, mode="economic", check_finite=False)
    return np.ascontiguousarray(q.astype(y.dtype, copy=False))


class CsrMat:
    def __init__(self, indptr, indices, values, shape):
        self.indptr = np.ascontiguousarray(indptr, np.uint64)
        self.indices = np.ascontiguousarray(indices, np.uint32)
        self.values = np.ascontiguousarray(values)
        self.shape = shape

    def dot(self, rhs):  # csr_mulacc_dense_rowmaj, svdapprox.rs:366
        m, n = self.shape
        if self.values.dtype == np.float32:
            rhs = np.ascontiguousarray(rhs, np.float32)
            out = np.zeros((m, rhs.shape[1]), np.float32)
            lib().orc_csr_mul_dense(C.c_uint64(m), _p(self.indptr), _p(self.indices), _p(self.values), _p(rhs),
                                    C.c_uint64(rhs.shape[1]), _p(out))
            return out
        return self.to_scipy() @ rhs

    def tdot(self, rhs):  # csc_mulacc_dense_rowmaj on transpose_view, svdapprox.rs:379
        m, n = self.shape
        if self.values.dtype == np.float32:
            rhs = np.ascontiguousarray(rhs, np.float32)
            out = np.zeros((n, rhs.shape[1]), np.float32)
            lib().orc_csr_t_mul_dense(C.c_uint64(m), _p(self.indptr), _p(self.indices), _p(self.values), _p(rhs),
                                      C.c_uint64(rhs.shape[1]), _p(out))
            return out
        return self.to_scipy().T @ rhs

    def to_scipy(self):
        import scipy.sparse as sp
        return sp.csr_matrix((self.values, self.indices.astype(np.int64), self.indptr.astype(np.int64)), shape=self.shape)


def gaussian_matrix(rows, cols, dtype=np.float32):
    """RandomGaussianMatrix::new(dims), svdapprox.rs:69-76 (build's own N(0,1) stream, seed 4664397)."""
    return gaussian_fill(rows * cols, OMEGA_SEED, TAG_OMEGA).reshape(rows, cols).astype(dtype)


def subspace_iteration(mat, rank, nbiter, omega=None):
    """subspace_iteration_full (svdapprox.rs:285-333) / subspace_iteration_csr (:343-408)."""
    m, n = mat.shape
    l = min(m, n, rank)  # :294 / :358
    is_csr = isinstance(mat, CsrMat)
    dtype = mat.values.dtype if is_csr else mat.dtype
    if omega is None:
        omega = gaussian_matrix(n, l, dtype)  # :299 / :363
    dot = (lambda x: mat.dot(x)) if is_csr else (lambda x: mat @ x)
    tdot = (lambda x: mat.tdot(x)) if is_csr else (lambda x: mat.T @ x)
    y = _do_qr(dot(omega))  # :300,:307 / :366,:374
    for _ in range(1, nbiter):  # :308 / :375
        yn = _do_qr(tdot(y))    # :311-319 / :379-387
        y = _do_qr(dot(yn))     # :321-329 / :390-398
    return y


def direct_svd(mat, rank, nbiter, omega=None):
    """SvdApprox::direct_svd with RangeApproxMode::RANK, svdapprox.rs:721-799."""
    import scipy.linalg as sla
    q = subspace_iteration(mat, rank, nbiter, omega)
    if isinstance(mat, CsrMat):
        b = np.ascontiguousarray(mat.tdot(q).T)  # transpose_dense_mult_csr, :116-139, :741
    else:
        b = q.T @ mat                            # :738
    ub, s, vt = sla.svd(b, full_matrices=False, lapack_driver="gesdd", check_finite=False)  # :758
    u = q @ ub                                   # :781
    return s, u, vt


def adaptative_range_finder(mat, epsil, r, max_rank, rng=None):
    """adaptative_range_finder_matrep, svdapprox.rs:444-597 (Halko-Martinsson-Tropp 4.2), line by line.  `rng`
    supplies the N(0,1) draws (the reference's Xoshiro stream is unpinned)."""
    rng = rng if rng is not None else np.random.default_rng(OMEGA_SEED)
    dot = (lambda v: mat.dot(v.reshape(-1, 1)).reshape(-1)) if isinstance(mat, CsrMat) else (lambda v: mat @ v)
    dtype = mat.values.dtype if isinstance(mat, CsrMat) else mat.dtype
    m, n = mat.shape
    q_mat = []
    stop_rel = epsil / (10.0 * np.sqrt(2.0 / (1.0 / np.pi)))                   # :465
    omega = rng.standard_normal((n, r)).astype(dtype) * dtype.type(1.0 / np.sqrt(n))  # :479-482
    y_vec = [dot(np.ascontiguousarray(omega[:, j])) for j in range(r)]         # :485-491
    norms = np.array([np.linalg.norm(y) for y in y_vec])
    if np.isnan(norms).any():
        raise FloatingPointError("adaptative_range_finder: NaN norms")          # :505-509
    norm_sup = norms.max()
    stop_val = norm_sup * stop_rel                                             # :515
    j = nb_iter = 0
    max_iter = min(m, n)

    def orth(qs, y):                                                           # orthogonalize_with_q, :975-992
        if not qs:
            return y
        proj = np.zeros_like(y)
        for it in qs:
            proj += it * it.dot(y)
        return y - proj

    while norm_sup > stop_val and nb_iter <= max_iter and len(q_mat) < max_rank:   # :517
        y_vec[j] = orth(q_mat, y_vec[j])                                       # :519-521
        n_j = np.linalg.norm(y_vec[j])
        if n_j < np.sqrt(np.finfo(dtype).eps):                                 # :524-532
            break
        q_j = y_vec[j] / n_j
        q_mat.append(q_j.copy())                                               # :535
        w = rng.standard_normal(n).astype(dtype) * dtype.type(1.0 / np.sqrt(n))    # :537-538
        y_vec[j] = orth(q_mat, dot(w))                                         # :539-544
        for k in range(r):                                                     # :546-553
            if k != j:
                y_vec[k] = y_vec[k] - q_j * q_j.dot(y_vec[k])
        norms = np.array([np.linalg.norm(y) for y in y_vec])                   # :555-561
        norm_sup = norms.max()
        j = (j + 1) % r
        nb_iter += 1
    return np.ascontiguousarray(np.stack(q_mat, 1)) if q_mat else np.zeros((m, 0), dtype)  # :586-594


def direct_svd_epsil(mat, epsil, step, max_rank, rng=None):
    """SvdApprox::direct_svd with RangeApproxMode::EPSIL(RangePrecision), svdapprox.rs:721-799"""
    import scipy.linalg as sla
    step = 2 if step <= 1 else step                                            # RangePrecision::new, :167-179
    q = adaptative_range_finder(mat, epsil, step, max_rank, rng)
    b = np.ascontiguousarray(mat.tdot(q).T) if isinstance(mat, CsrMat) else q.T @ mat
    ub, s, vt = sla.svd(b, full_matrices=False, lapack_driver="gesdd", check_finite=False)
    return s, q @ ub, vt


def svd_full(mat):
    """svd_f32, graphlaplace.rs:296-344: gesdd JobSvd::Some -> s, u."""
    import scipy.linalg as sla
    u, s, _ = sla.svd(mat, full_matrices=False, lapack_driver="gesdd", check_finite=False)
    return s, u


def estimate_first_singular_value(mat):
    """estimate_first_singular_value_fullmat, svdapprox.rs:891-945 (power iteration on A A^T)."""
    mat = np.asarray(mat, np.float64)
    a2 = mat @ mat.T if mat.shape[0] <= mat.shape[1] else mat.T @ mat
    v1 = np.full(a2.shape[0], 1.0 / np.sqrt(a2.shape[0]))
    lam = 0.0
    for _ in range(1000):
        v2 = a2 @ v1
        lam = np.sqrt(v2 @ v2)
        if lam <= np.finfo(np.float64).eps:
            break
        v2 = v2 / lam
        delta = np.sqrt((v1 - v2) @ (v1 - v2))
        v1 = v2
        if delta < 1e-8:
            break
    return np.sqrt(lam)


def estimate_first_singular_value_csmat(mat):
    """estimate_first_singular_value_csmat, svdapprox.rs:844-887: densify, power iteration on A A^T (or A^T A), stop at
    |v1 - v2| < 1e-10 or 1000 iterations"""
    dense = np.zeros(mat.shape, np.float64)
    rows = np.repeat(np.arange(mat.shape[0]), np.diff(mat.indptr.astype(np.int64)))
    dense[rows, mat.indices.astype(np.int64)] = mat.values
    a2 = dense @ dense.T if dense.shape[0] <= dense.shape[1] else dense.T @ dense
    v1 = np.full(a2.shape[0], 1.0 / np.sqrt(a2.shape[0]))
    lam = 0.0
    for _ in range(1000):
        v2 = a2 @ v1
        lam = np.sqrt(v2 @ v2)
        v2 = v2 * 1.0 / lam
        w = v1 - v2
        if np.sqrt(w @ w) < 1e-10:
            break
        v1 = v2
    return np.sqrt(lam)


# ------------------------------------------------------------------------------------------------
# diffusion-map drivers
# ------------------------------------------------------------------------------------------------
class DiffusionParams:
    """DiffusionParams::new, src/diffmaps.rs:95-105."""

    def __init__(self, asked_dim=2, t=None, gnbn=None, alfa=0.5, beta=-0.1, epsil=2.0):
        self.asked_dim, self.t, self.gnbn = asked_dim, t, gnbn
        self.alfa, self.beta, self.epsil = np.float32(alfa), np.float32(beta), np.float32(epsil)


def _kernel_dense(kindptr, kcols, kvals, n):
    p = np.zeros((n, n), np.float32)
    rows = np.repeat(np.arange(n), np.diff(kindptr).astype(np.int64))
    p[rows, kcols] = kvals  # later duplicates overwrite, as transition_proba[[i, edge.node]] = w  (:455,:875)
    return p


def dmap_laplacian(indptr, nbr, dist, max_nbng, dp, force_repr=0):
    """laplacian_from_kgraph (diffmaps.rs:397-422) = compute_dmap_nodeparams + compute_laplacian."""
    n = len(indptr) - 1
    nbng = min(dp.gnbn, max_nbng) if dp.gnbn is not None else max_nbng  # :414-418
    local, normed, mean_scale = dmap_local_scales(indptr, dist, min(max_nbng, nbng))  # :777, :784-822
    use_dense = (n <= FULL_MAT_REPR) if force_repr == 0 else (force_repr == 1)
    if dp.beta > 0:
        return 9, None
    q = None
    beta_scales = None
    if dp.beta < 0:  # :837-843
        rc, kp, kc, kv, _ = dmap_kernel(indptr, nbr, dist, local, dp.epsil)
        if rc:
            return rc, None
        if use_dense:  # kernel0_to_density dense branch :865-892
            p = _kernel_dense(kp, kc, kv, n)
            sym = (p + p.T) * np.float32(0.5)
            q = sym.sum(axis=1, dtype=np.float32) / np.float32(max_nbng)
            q = (q / (q.sum(dtype=np.float32) / np.float32(n))).astype(np.float32)
            beta_scales = (np.power(q, dp.beta, dtype=np.float32) * mean_scale).astype(np.float32)  # :938-942
        else:
            q, beta_scales = dmap_density_csr(kp, kc, kv, max_nbng, dp.beta, mean_scale)
        rc, kp, kc, kv, _ = dmap_kernel(indptr, nbr, dist, beta_scales, dp.epsil)  # :841
    else:  # :844-848
        rc, kp, kc, kv, _ = dmap_kernel(indptr, nbr, dist, np.full(n, mean_scale, np.float32), dp.epsil)
    if rc:
        return rc, None
    out = dict(normed_scales=normed, mean_scale=mean_scale, q=q, beta_scales=beta_scales, n=n)
    if use_dense:  # compute_laplacian dense branch :445-508
        p = _kernel_dense(kp, kc, kv, n)
        sym = (p + p.T) * np.float32(0.5)  # :460
        qq = sym.sum(axis=1, dtype=np.float32)  # :468
        qq = qq / (qq.sum(dtype=np.float32) / np.float32(max_nbng))  # :469-471
        sym = sym / np.power(np.outer(qq, qq), dp.alfa, dtype=np.float32)  # :476
        deg = sym.sum(axis=1, dtype=np.float32)  # :478
        sw = np.sqrt(deg)  # :482
        sym = (sym / np.outer(sw, sw)).astype(np.float32)  # :486
        out.update(is_csr=False, dense=sym, normalizer=sw.astype(np.float32))
    else:
        li, lc, lv, norm = dmap_laplacian_csr(kp, kc, kv, max_nbng, dp.alfa)
        out.update(is_csr=True, csr=CsrMat(li, lc, lv, (n, n)), normalizer=norm)
    return 0, out


def laplacian_do_svd(lap):
    """GraphLaplacian::do_svd, graphlaplace.rs:127-134."""
    if not lap["is_csr"] and lap["n"] <= FULL_SVD_SIZE_LIMIT:
        return svd_full(lap["dense"])  # :82-94
    mat = lap["csr"] if lap["is_csr"] else lap["dense"]
    s, u, _ = direct_svd(mat, 20, 5)  # :111-116
    return s, u


def dmap_embed_from_kgraph(indptr, nbr, dist, max_nbng, dp, force_repr=0):
    """DiffusionMaps::embed_from_kgraph, diffmaps.rs:1047-1075."""
    rc, lap = dmap_laplacian(indptr, nbr, dist, max_nbng, dp, force_repr)
    if rc:
        return rc, None, None
    s, u = laplacian_do_svd(lap)
    rc, y0 = embed_from_svd(s, u, lap["normalizer"], lap["normed_scales"], dp.asked_dim, dp.t)
    return rc, y0, dict(lap=lap, s=s, u=u)


# ------------------------------------------------------------------------------------------------
# Embedder drivers
# ------------------------------------------------------------------------------------------------
class EmbedderParams:
    """EmbedderParams::default, src/embedparams.rs:107-132."""

    def __init__(self, **kw):
        self.asked_dim = 2
        self.dmap_init = True
        self.beta = 1.0
        self.b = 1.0
        self.scale_rho = 1.0
        self.grad_step = 2.0
        self.nb_sampling_by_edge = 10
        self.nb_grad_batch = 20
        self.grad_factor = 4
        self.hierarchy_layer = 0
        self.hubness_weighting = False
        self.seed = OMEGA_SEED
        self.sampler = 0
        for k, v in kw.items():
            assert hasattr(self, k), k
            setattr(self, k, v)


def one_step_embed(indptr, nbr, dist, max_nbng, params, hogwild_threads=None):
    """Embedder::one_step_embed, src/embedder.rs:298-371 (with the build decisions B1, B2 of SURVEY app. B)."""
    n = len(indptr) - 1
    if params.dmap_init:
        dp = DiffusionParams(params.asked_dim, 5.0, 12)  # :317-321 (B1: asked_dim instead of the hard-wired 2)
        rc, y0, _ = dmap_embed_from_kgraph(indptr, nbr, dist, max_nbng, dp)
        if rc:
            return rc, None
        y0 = set_data_box(y0, 10.0)  # :345
    else:
        y0 = random_init(n, params.asked_dim, 1.0, params.seed)  # :348 (B2)
    rc, proba, scale = to_proba_edges(indptr, nbr, dist, params.scale_rho, params.beta)  # :351
    if rc:
        return rc, None
    hub = hubness(indptr, nbr) if params.hubness_weighting else None
    y, ce0, ce1 = entropy_optimize(indptr, nbr, proba, scale, y0, params.nb_grad_batch, params.nb_sampling_by_edge,
                                   params.grad_step, params.b, params.seed, params.sampler, hub, hogwild_threads)
    return 0, dict(y=y, y0=y0, ce_before=ce0, ce_after=ce1, hubness=hub)


def h_embed(small, large, proj_node, proj_dist, params, hogwild_threads=None):
    """Embedder::h_embed, src/embedder.rs:194-295.  small/large: (indptr, nbr, dist, max_nbng)."""
    import copy
    first = copy.copy(params)
    first.nb_grad_batch = params.grad_factor * params.nb_grad_batch  # :204-205
    first.grad_step = 1.0      # :207
    first.hierarchy_layer = 0  # :208
    rc, res1 = one_step_embed(*small, first, hogwild_threads)  # :213
    if rc:
        return rc, None
    indptr, nbr, dist, max_nbng = large
    rc, proba, scale = to_proba_edges(indptr, nbr, dist, params.scale_rho, params.beta)  # :226
    if rc:
        return rc, None
    n_small = len(small[0]) - 1
    n_large = len(indptr) - 1
    pd = np.asarray(proj_dist, np.float32)[n_small:]
    median = np.float32(np.sort(pd)[(len(pd) - 1) // 2]) if len(pd) else np.float32(1)  # exact lower median for :255
    y0 = projection_init(res1["y"], n_large, proj_node, proj_dist, median, params.seed)  # :245-269
    hub = hubness(indptr, nbr) if params.hubness_weighting else None
    y, ce0, ce1 = entropy_optimize(indptr, nbr, proba, scale, y0, params.nb_grad_batch, params.nb_sampling_by_edge,
                                   params.grad_step, params.b, params.seed, params.sampler, hub, hogwild_threads)
    return 0, dict(y=y, y0=y0, ce_before=ce0, ce_after=ce1, hubness=hub, first=res1)


# ------------------------------------------------------------------------------------------------
# quality estimate
# ------------------------------------------------------------------------------------------------
QUALITY_PROBAS = (0.05, 0.25, 0.5, 0.75, 0.85, 0.95)


def _distl2_f32(a, b):
    """distl2, src/embedder.rs:54-65: f32 sum of squares in coordinate order, sqrt"""
    s = np.zeros(a.shape[0], np.float32)
    for t in range(a.shape[1]):
        df = (a[:, t] - b[:, t]).astype(np.float32)
        s = (s + df * df).astype(np.float32)
    return np.sqrt(s).astype(np.float32)


def quality_estimate(indptr, nbr, y, nbng):
    """get_quality_estimate_from_edge_length, src/embedder.rs:620-753, with get_transformed_kgraph (:478-522: running
    minimum of the embedded edge lengths in neighbour order, then sorted) and the radius of
    get_max_edge_length_embedded_kgraph (:527-554) taken from the EXACT nbng-nearest-neighbour graph of the embedded
    points (the reference's hnsw_rs graph is approximate: parity unpinned); exact order statistics at rank
    floor(q * count) stand for the CKMS(0.01) sketches."""
    from scipy.spatial import cKDTree
    y = np.ascontiguousarray(y, np.float32)
    n = y.shape[0]
    indptr = np.asarray(indptr, np.int64)
    nbr = np.asarray(nbr, np.int64)
    src = np.repeat(np.arange(n), np.diff(indptr))
    d = _distl2_f32(y[src], y[nbr])
    tw = np.empty_like(d)
    for i in range(n):  # :499-512
        b, e = indptr[i], indptr[i + 1]
        tw[b:e] = np.minimum.accumulate(d[b:e])[::-1]
    # candidates from a k-d tree, distances recomputed with the f32 arithmetic of the device path
    _, idx = cKDTree(y.astype(np.float64)).query(y.astype(np.float64), k=nbng + 3)
    radius = np.zeros(n, np.float64)
    for c0 in range(0, n, 4096):
        c1 = min(n, c0 + 4096)
        cand = idx[c0:c1]
        dd = np.stack([_distl2_f32(y[c0:c1], y[cand[:, t]]) for t in range(cand.shape[1])], 1)
        dd[cand == np.arange(c0, c1)[:, None]] = np.inf
        radius[c0:c1] = np.sort(dd, axis=1)[:, nbng - 1].astype(np.float64)
    ratio = tw.astype(np.float64) / radius[src]
    match = np.add.reduceat((tw.astype(np.float64) <= radius[src]).astype(np.int64), indptr[:-1])
    nb_without = int((match == 0).sum())
    q = lambda v: np.array([np.sort(v)[min(len(v) - 1, int(p * len(v)))] for p in QUALITY_PROBAS])
    return {
        "nb_without_match": nb_without, "mean_nbmatch": match.sum() / max(1, n - nb_without),
        "radii_quantiles": q(radius), "ratio_quantiles": q(ratio), "median_ratio": q(ratio)[2], "mean_ratio": ratio.mean(),
        "ratio_by_node": np.add.reduceat(ratio, indptr[:-1]) / np.maximum(1, np.diff(indptr)), "first_dist": tw[indptr[:-1]].astype(np.float64),
        "radius": radius,
    }
