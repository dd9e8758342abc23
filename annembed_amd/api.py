"""Host-side mirror of the reference's Rust API for the embedding hot path, over the C ABI.

Names, argument meaning and error behaviour follow the reference (file:line in each docstring);
everything computes on the GPU through libannembed_hip.so -- there is no CPU path here.
"""
import ctypes as C

import numpy as np

from . import _lib as L
from ._lib import AnnembedError, check, ptr  # noqa: F401


def _f32(a):
    return np.ascontiguousarray(a, np.float32)


def _u32(a):
    return np.ascontiguousarray(a, np.uint32)


def _u64(a):
    return np.ascontiguousarray(a, np.uint64)


class EmbedderParams:
    """EmbedderParams, src/embedparams.rs:77-132 (same fields, same defaults) + seed / ce_mode / ce_sampler."""

    FIELDS = [f[0] for f in L.CEmbedderParams._fields_]

    def __init__(self, **kw):
        c = L.CEmbedderParams()
        check(L.load().ae_embedder_params_default(C.byref(c)))
        for f in self.FIELDS:
            setattr(self, f, getattr(c, f))
        self.dmap_init = bool(self.dmap_init)
        self.hubness_weighting = bool(self.hubness_weighting)
        for k, v in kw.items():
            if k not in self.FIELDS:
                raise AttributeError(k)
            setattr(self, k, v)

    @staticmethod
    def default():
        return EmbedderParams()

    # setters of src/embedparams.rs:150-184
    def set_dim(self, dim):
        self.asked_dim = dim

    def get_dimension(self):
        return self.asked_dim

    def set_hierarchy_layer(self, layer):
        self.hierarchy_layer = layer

    def get_hierarchy_layer(self):
        return self.hierarchy_layer

    def set_nb_gradient_batch(self, nb):
        self.nb_grad_batch = nb

    def set_nb_edge_sampling(self, nb):
        self.nb_sampling_by_edge = nb

    def c(self):
        c = L.CEmbedderParams()
        for f in self.FIELDS:
            setattr(c, f, int(getattr(self, f)) if f in ("dmap_init", "hubness_weighting") else getattr(self, f))
        return c


class DiffusionParams:
    """DiffusionParams, src/diffmaps.rs:72-222."""

    def __init__(self, asked_dim, t_opt=None, g_opt=None):
        self._c = L.CDiffusionParams()
        check(L.load().ae_diffusion_params_new(C.byref(self._c), asked_dim, 0.0 if t_opt is None else t_opt,
                                               0 if t_opt is None else 1, 0 if g_opt is None else g_opt,
                                               0 if g_opt is None else 1))

    def set_alfa(self, alfa):
        check(L.load().ae_diffusion_params_set_alfa(C.byref(self._c), alfa))

    def set_beta(self, beta):
        check(L.load().ae_diffusion_params_set_beta(C.byref(self._c), beta))

    def set_epsil(self, epsil):
        check(L.load().ae_diffusion_params_set_epsil(C.byref(self._c), epsil))

    def get_alfa(self):
        return self._c.alfa

    def get_beta(self):
        return self._c.beta

    def get_epsil(self):
        return self._c.epsil

    def get_time(self):
        return self._c.t if self._c.has_t else None

    def get_gnbn(self):
        return self._c.gnbn if self._c.has_gnbn else None

    def get_data_dim(self):
        return self._c.asked_dim

    def c(self):
        return self._c


class _Handle:
    _destroy = None

    def __init__(self, h):
        self._h = h

    @property
    def handle(self):
        return self._h

    def close(self):
        if getattr(self, "_h", None):
            getattr(L.load(), self._destroy)(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _partition_report(rep):
    return {"components": int(rep.components), "splits": int(rep.splits), "cross_mass": float(rep.cross_mass),
            "cross_mass_worst_rank": float(rep.cross_mass_worst_rank), "imbalance": float(rep.imbalance)}


def set_summation_order(tree):
    """ae_set_summation_order: True (the library's default) = f64 tree reductions in the stage-level entry points, False = the reference's
    sequential f32 order (single-lane chains: what bit parity with the oracle needs).  Embedder.embed / EntropyOptim choose by their CE mode."""
    check(L.load().ae_set_summation_order(1 if tree else 0))


class KGraph(_Handle):
    """KGraph<f32>, src/fromhnsw/kgraph.rs:109-120, resident on the GPU as a CSR."""

    _destroy = "ae_kgraph_destroy"

    def __init__(self, indptr, nbr, dist, max_nbng=None):
        indptr, nbr, dist = _u64(indptr), _u32(nbr), _f32(dist)
        n = len(indptr) - 1
        if max_nbng is None:
            max_nbng = int(np.diff(indptr.astype(np.int64)).max()) if n else 0
        h = C.c_void_p()
        check(L.load().ae_kgraph_create(ptr(indptr), ptr(nbr), ptr(dist), n, max_nbng, C.byref(h)))
        super().__init__(h)
        self.data_ids = None

    @classmethod
    def _wrap(cls, h):
        o = cls.__new__(cls)
        _Handle.__init__(o, h)
        o.data_ids = None
        return o

    @classmethod
    def load(cls, path):
        """KGraph from a `.kgraph` file (annembed_amd/io.py); `.data_ids` keeps the DataIds"""
        from . import io as _io
        d = _io.read_kgraph(path)
        g = cls(d["indptr"], d["nbr"], d["dist"], d["max_nbng"])
        g.data_ids = d["data_ids"]
        return g

    def save(self, path):
        from . import io as _io
        indptr, nbr, dist = self.get_neighbours()
        _io.write_kgraph(path, indptr, nbr, dist, self.get_max_nbng(), getattr(self, "data_ids", None))

    @classmethod
    def from_ragged(cls, point_id, row_ptr, nbr_data_id, nbr_dist, nbng):
        """Tail of kgraph_from_hnsw_all (kgraph.rs:486-546): flatten, reindex, sort, truncate."""
        point_id, row_ptr, nbr_data_id, nbr_dist = _u64(point_id), _u64(row_ptr), _u64(nbr_data_id), _f32(nbr_dist)
        ids = np.zeros(len(point_id), np.uint64)
        h = C.c_void_p()
        check(L.load().ae_kgraph_from_ragged(ptr(point_id), ptr(row_ptr), ptr(nbr_data_id), ptr(nbr_dist), len(point_id),
                                             nbng, C.byref(h), ptr(ids)))
        g = cls._wrap(h)
        g.data_ids = ids
        return g

    @classmethod
    def bruteforce_l2(cls, x, nbng):
        x = _f32(x)
        h = C.c_void_p()
        check(L.load().ae_kgraph_bruteforce_l2(ptr(x), x.shape[0], x.shape[1], nbng, C.byref(h)))
        return cls._wrap(h)

    @classmethod
    def bruteforce_l2_grouped(cls, x, nbng, bounds):
        """the exact global kNN graph of points sorted into groups (ae_kgraph_bruteforce_l2_grouped); `.knn_stats` = (fallback rows,
        pairs of the pruned phase, pairs inside the groups)"""
        x, bounds = _f32(x), _u64(bounds)
        h = C.c_void_p()
        st = np.zeros(3, np.uint64)
        check(L.load().ae_kgraph_bruteforce_l2_grouped(ptr(x), x.shape[0], x.shape[1], nbng, ptr(bounds), len(bounds) - 1, C.byref(h), ptr(st)))
        g = cls._wrap(h)
        g.knn_stats = tuple(int(v) for v in st)
        return g

    def partition(self, world, y=None, node_params=None):
        """Locality partition into `world` contiguous ranges (ae_kgraph_partition; no reference counterpart: SURVEY 8e "after locality
        reordering"): connected components packed whole, a component that must be cut is cut by coordinate bisection of y (n x dim;
        None: id order).  -> order (order[pos] = this graph's node at position pos), ranges [(lo, hi)] per rank, report dict"""
        n = self.get_nb_nodes()
        order = np.zeros(n, np.uint32)
        ranges = np.zeros(2 * world, np.uint64)
        rep = L.CPartitionReport()
        dim = 0
        if y is not None:
            y = _f32(y)
            if y.ndim != 2 or y.shape[0] != n:
                raise ValueError("y must have one row per node")
            dim = y.shape[1]
        check(L.load().ae_kgraph_partition(self._h, node_params._h if node_params is not None else None, ptr(y), dim, world, ptr(order), ptr(ranges), C.byref(rep)))
        return order, [(int(ranges[2 * r]), int(ranges[2 * r + 1])) for r in range(world)], _partition_report(rep)

    def permuted(self, order):
        """the same graph with the node at position p = node order[p] (ae_kgraph_permuted); coordinates come back as y_here[order] = y_there"""
        order = _u32(order)
        h = C.c_void_p()
        check(L.load().ae_kgraph_permuted(self._h, ptr(order), C.byref(h)))
        return KGraph._wrap(h)

    def get_nb_nodes(self):
        v = C.c_uint64()
        check(L.load().ae_kgraph_get_nb_nodes(self._h, C.byref(v)))
        return v.value

    def get_max_nbng(self):
        v = C.c_uint32()
        check(L.load().ae_kgraph_get_max_nbng(self._h, C.byref(v)))
        return v.value

    def get_nb_edges(self):
        v = C.c_uint64()
        check(L.load().ae_kgraph_get_nb_edges(self._h, C.byref(v)))
        return v.value

    def get_neighbours(self):
        n, nnz = self.get_nb_nodes(), self.get_nb_edges()
        indptr, nbr, dist = np.zeros(n + 1, np.uint64), np.zeros(nnz, np.uint32), np.zeros(nnz, np.float32)
        check(L.load().ae_kgraph_get_neighbours(self._h, ptr(indptr), ptr(nbr), ptr(dist)))
        return indptr, nbr, dist

    def fill_l2_distances(self, x):
        x = _f32(x)
        check(L.load().ae_kgraph_fill_l2_distances(self._h, ptr(x), x.shape[1]))

    def hubness(self):
        """Hubness::new(...).get_counts(), src/fromhnsw/hubness.rs:39-80."""
        c = np.zeros(self.get_nb_nodes(), np.uint32)
        check(L.load().ae_kgraph_hubness(self._h, ptr(c)))
        return c


class KGraphProjection(_Handle):
    """KGraphProjection accessors, src/fromhnsw/kgproj.rs:376-410."""

    _destroy = "ae_kgraph_projection_destroy"

    def __init__(self, small, large, proj_node, proj_dist):
        self.small, self.large = small, large
        proj_node, proj_dist = _u32(proj_node), _f32(proj_dist)
        h = C.c_void_p()
        check(L.load().ae_kgraph_projection_create(small.handle, large.handle, ptr(proj_node), ptr(proj_dist), C.byref(h)))
        super().__init__(h)

    def get_small_graph(self):
        return self.small

    def projection_init(self, y_small, seed=4664397):
        """the initial embedding h_embed gives the large graph (src/embedder.rs:245-269)"""
        y_small = _f32(y_small)
        if y_small.ndim != 2 or y_small.shape[0] != self.small.get_nb_nodes():
            raise ValueError("y_small must have one row per node of the small graph (%d), got shape %s" % (self.small.get_nb_nodes(), y_small.shape))
        n_large = self.large.get_nb_nodes()
        y0 = np.zeros((n_large, y_small.shape[1]), np.float32)
        check(L.load().ae_projection_init(self._h, ptr(y_small), y_small.shape[0], y_small.shape[1], seed, ptr(y0)))
        return y0

    def get_large_graph(self):
        return self.large


class NodeParams(_Handle):
    """NodeParams, src/tools/nodeparam.rs:111-114 (device CSR aligned with the KGraph)."""

    _destroy = "ae_node_params_destroy"

    def __init__(self, h, kgraph):
        super().__init__(h)
        self.kgraph = kgraph

    @classmethod
    def from_host(cls, kgraph, proba, scale):
        """NodeParams::new, src/tools/nodeparam.rs:117-119"""
        proba, scale = _f32(proba), _f32(scale)
        h = C.c_void_p()
        check(L.load().ae_node_params_from_host(kgraph.handle, ptr(proba), ptr(scale), C.byref(h)))
        return cls(h, kgraph)

    def get(self):
        proba = np.zeros(self.kgraph.get_nb_edges(), np.float32)
        scale = np.zeros(self.kgraph.get_nb_nodes(), np.float32)
        check(L.load().ae_node_params_get(self._h, ptr(proba), ptr(scale)))
        return proba, scale

    def get_perplexity(self):
        p = np.zeros(self.kgraph.get_nb_nodes(), np.float32)
        check(L.load().ae_node_params_perplexity(self._h, ptr(p)))
        return p


def to_proba_edges(kgraph, scale_rho, beta):
    """to_proba_edges, src/tools/kdumap.rs:26-116."""
    h = C.c_void_p()
    check(L.load().ae_to_proba_edges(kgraph.handle, scale_rho, beta, C.byref(h)))
    return NodeParams(h, kgraph)


def set_data_box(y, box_size=10.0):
    """set_data_box, src/embedder.rs:1376-1408."""
    y = _f32(y).copy()
    check(L.load().ae_set_data_box(ptr(y), y.shape[0], y.shape[1], box_size))
    return y


class EntropyOptim(_Handle):
    """EntropyOptim, src/embedder.rs:936-1315."""

    _destroy = "ae_entropy_optim_destroy"

    def __init__(self, kgraph, node_params, params, y0, hub_counts=None, node_lo=0, node_hi=None):
        self.kgraph, self.node_params, self.params = kgraph, node_params, params
        y0 = _f32(y0)
        self.n, self.dim = y0.shape
        cp = params.c()
        hub = None if hub_counts is None else _u32(hub_counts)
        h = C.c_void_p()
        check(L.load().ae_entropy_optim_create(kgraph.handle, node_params.handle, C.byref(cp), ptr(y0), ptr(hub), node_lo,
                                               self.n if node_hi is None else node_hi, C.byref(h)))
        super().__init__(h)

    def get_nb_edges(self):
        v = C.c_uint64()
        check(L.load().ae_entropy_optim_get_nb_edges(self._h, C.byref(v)))
        return v.value

    def get_ce_mode(self):
        v = C.c_uint32()
        check(L.load().ae_entropy_optim_get_ce_mode(self._h, C.byref(v)))
        return v.value

    def slice_info(self):
        """AE_CE_SLICED: (colour classes run as matchings, overflow share of the probability mass, colouring rounds, slices of the last batch)"""
        a, b, c, d = C.c_uint32(), C.c_double(), C.c_uint32(), C.c_uint32()
        check(L.load().ae_entropy_optim_slice_info(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(d)))
        return a.value, b.value, c.value, d.value

    def comm_bytes(self):
        """bytes of coordinate rows received through the exchanges of this handle's batches (multi-GPU)"""
        v = C.c_uint64()
        check(L.load().ae_entropy_optim_comm_bytes(self._h, C.byref(v)))
        return v.value

    def slice_hub_info(self):
        """AE_CE_SLICED: (largest in-degree of the graph, expected length of the longest chain of a step)"""
        a, b = C.c_uint32(), C.c_double()
        check(L.load().ae_entropy_optim_slice_hub_info(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def slice_form(self):
        """AE_CE_SLICED: the launch form of the last batch -- 0 none yet, 1 one launch per class, 2 the same on node lines, 3 merged slices without
        the class window (debug knob), 4 optimistic passes only, 5 merged slices with the class window -- what merged slices run as
        (ae_entropy_optim_slice_form: how old the negatives' rows are)"""
        v = C.c_uint32()
        check(L.load().ae_entropy_optim_slice_form(self._h, C.byref(v)))
        return v.value

    def ce_compute_threaded(self):
        v = C.c_double()
        check(L.load().ae_entropy_optim_ce(self._h, C.byref(v)))
        return v.value

    def gradient_iteration_threaded(self, nb_sample, grad_step, it):
        check(L.load().ae_entropy_optim_gradient_iteration(self._h, nb_sample, grad_step, it))

    @staticmethod
    def gradient_iteration_lockstep(shards, nb_samples, grad_step, it, exchanges_per_batch=1):
        """one batch of the sharded protocol on one device (ae_entropy_optim_gradient_iteration_lockstep): `shards` are
        rounds-mode handles whose node ranges tile [0, n) in order"""
        hs = (C.c_void_p * len(shards))(*[sh._h for sh in shards])
        ns = (C.c_uint64 * len(shards))(*[int(x) for x in nb_samples])
        check(L.load().ae_entropy_optim_gradient_iteration_lockstep(hs, len(shards), ns, grad_step, it, exchanges_per_batch))

    def plan(self, s_begin, count, it):
        nodes = np.zeros((count, 7), np.uint32)
        w = np.zeros(count, np.float32)
        check(L.load().ae_entropy_optim_plan(self._h, s_begin, count, it, ptr(nodes), ptr(w)))
        return nodes, w

    def samples_drawn(self):
        a, r = C.c_uint64(), C.c_uint32()
        check(L.load().ae_entropy_optim_samples_drawn(self._h, C.byref(a), C.byref(r)))
        return a.value, r.value

    def get_embedded_scales(self):
        s = np.zeros(self.n, np.float32)
        check(L.load().ae_entropy_optim_get_scales(self._h, ptr(s)))
        return s

    def get_embedded(self):
        y = np.zeros((self.n, self.dim), np.float32)
        check(L.load().ae_entropy_optim_get_embedded(self._h, ptr(y)))
        return y

    def device_coords(self):
        p, n, d = C.c_void_p(), C.c_uint64(), C.c_uint64()
        check(L.load().ae_entropy_optim_device_coords(self._h, C.byref(p), C.byref(n), C.byref(d)))
        return p.value, n.value, d.value

    def dataflow_time(self):
        ms, cnt = C.c_double(), C.c_uint64()
        check(L.load().ae_entropy_optim_dataflow_time(self._h, C.byref(ms), C.byref(cnt)))
        return ms.value, cnt.value

    def kernel_time(self):
        ms, cnt = C.c_double(), C.c_uint64()
        check(L.load().ae_entropy_optim_kernel_time(self._h, C.byref(ms), C.byref(cnt)))
        return ms.value, cnt.value


def entropy_optimize(kgraph, node_params, params, y0):
    """entropy_optimize, src/embedder.rs:794-904 -> (embedding, ce_before, ce_after)."""
    y0 = _f32(y0)
    y = np.zeros_like(y0)
    a, b = C.c_double(), C.c_double()
    cp = params.c()
    check(L.load().ae_entropy_optimize(kgraph.handle, node_params.handle, C.byref(cp), ptr(y0), ptr(y), C.byref(a), C.byref(b)))
    return y, a.value, b.value


class MatRepr(_Handle):
    """MatRepr<f32>, src/tools/matrepr.rs:23-32."""

    _destroy = "ae_matrepr_destroy"

    @classmethod
    def from_csrmat(cls, indptr, indices, values, shape):
        indptr, indices, values = _u64(indptr), _u32(indices), _f32(values)
        h = C.c_void_p()
        check(L.load().ae_matrepr_from_csr(ptr(indptr), ptr(indices), ptr(values), shape[0], shape[1], C.byref(h)))
        o = cls(h)
        o.shape = tuple(shape)
        return o

    @classmethod
    def from_array2(cls, mat):
        mat = _f32(mat)
        h = C.c_void_p()
        check(L.load().ae_matrepr_from_dense(ptr(mat), mat.shape[0], mat.shape[1], C.byref(h)))
        o = cls(h)
        o.shape = mat.shape
        return o


class RangeRank:
    """RangeRank, src/tools/svdapprox.rs:186-198."""

    def __init__(self, rank, nbiter):
        self.rank, self.nbiter = rank, nbiter


class RangePrecision:
    """RangePrecision, src/tools/svdapprox.rs:155-180 (step <= 1 is reset to 2)."""

    def __init__(self, epsil, step, max_rank):
        self.epsil, self.step, self.max_rank = float(epsil), (2 if step <= 1 else int(step)), int(max_rank)


def adaptative_range_finder_matrep(mat, epsil, r, max_rank):
    """adaptative_range_finder_matrep, src/tools/svdapprox.rs:444-597 -> Q (m, l), l <= max_rank"""
    m, n = mat.shape
    cap = min(int(max_rank), 4096)
    q = np.zeros((m, cap), np.float32)
    lo = C.c_uint64()
    check(L.load().ae_adaptative_range_finder(mat.handle, float(epsil), int(r), int(max_rank), ptr(q), C.byref(lo)))
    return np.ascontiguousarray(q.reshape(-1)[:m * lo.value].reshape(m, lo.value))


class RangeApprox:
    """RangeApprox, src/tools/svdapprox.rs:211-266: get_approximator() dispatches on the mode"""

    def __init__(self, mat, mode):
        self.mat, self.mode = mat, mode

    def get_approximator(self):
        if isinstance(self.mode, RangePrecision):
            return adaptative_range_finder_matrep(self.mat, self.mode.epsil, self.mode.step, self.mode.max_rank)
        return subspace_iteration(self.mat, self.mode.rank, self.mode.nbiter)


def subspace_iteration(mat, rank, nbiter):
    """subspace_iteration_full / _csr, src/tools/svdapprox.rs:285-408."""
    m, n = mat.shape
    l = min(m, n, rank)
    q = np.zeros((m, l), np.float32)
    lo = C.c_uint64()
    check(L.load().ae_subspace_iteration(mat.handle, rank, nbiter, ptr(q), C.byref(lo)))
    assert lo.value == l
    return q


class SvdResult:
    """SvdResult, src/tools/svdapprox.rs:653-661."""

    def __init__(self, s, u, vt):
        self.s, self.u, self.vt = s, u, vt

    def get_sigma(self):
        return self.s

    def get_u(self):
        return self.u

    def get_vt(self):
        return self.vt


class SvdApprox:
    """SvdApprox, src/tools/svdapprox.rs:698-800."""

    def __init__(self, data):
        self.data = data

    def direct_svd(self, mode, want_u=True, want_vt=True):
        """-> SvdResult(s, u, vt).  want_u / want_vt False (RANK mode): that factor is not copied to the host -- None in the result; Vt is
        then not computed at all (ae_svd_approx_rank with null pointers) -- for timing the device path on tall matrices."""
        m, n = self.data.shape
        if isinstance(mode, RangePrecision):  # RangeApproxMode::EPSIL
            cap = min(mode.max_rank, 64)
            s, u, vt = np.zeros(cap, np.float32), np.zeros(m * cap, np.float32), np.zeros(cap * n, np.float32)
            lo = C.c_uint64()
            check(L.load().ae_svd_approx_epsil(self.data.handle, mode.epsil, mode.step, mode.max_rank, ptr(s), ptr(u), ptr(vt), C.byref(lo)))
            l = lo.value
            return SvdResult(s[:l].copy(), u[:m * l].reshape(m, l).copy(), vt[:l * n].reshape(l, n).copy())
        l = min(m, n, mode.rank)
        s = np.zeros(l, np.float32)
        u = np.zeros((m, l), np.float32) if want_u else None
        vt = np.zeros((l, n), np.float32) if want_vt else None
        lo = C.c_uint64()
        check(L.load().ae_svd_approx_rank(self.data.handle, mode.rank, mode.nbiter, ptr(s), ptr(u) if want_u else None, ptr(vt) if want_vt else None, C.byref(lo)))
        return SvdResult(s, u, vt)


def transpose_dense_mult_csr(qmat, mat):
    """transpose_dense_mult_csr, src/tools/svdapprox.rs:116-139."""
    qmat = _f32(qmat)
    b = np.zeros((qmat.shape[1], mat.shape[1]), np.float32)
    check(L.load().ae_transpose_dense_mult(mat.handle, ptr(qmat), qmat.shape[1], ptr(b)))
    return b


class GraphLaplacian(_Handle):
    """GraphLaplacian, src/graphlaplace.rs:21-35."""

    _destroy = "ae_laplacian_destroy"

    def info(self):
        a, n, z = C.c_int32(), C.c_uint64(), C.c_uint64()
        check(L.load().ae_laplacian_info(self._h, C.byref(a), C.byref(n), C.byref(z)))
        return bool(a.value), n.value, z.value

    def is_csr(self):
        return self.info()[0]

    def get_sym_kernel(self):
        is_csr, n, nnz = self.info()
        if is_csr:
            indptr, ind, val = np.zeros(n + 1, np.uint64), np.zeros(nnz, np.uint32), np.zeros(nnz, np.float32)
            check(L.load().ae_laplacian_get_kernel(self._h, ptr(indptr), ptr(ind), ptr(val)))
            return indptr, ind, val
        val = np.zeros((n, n), np.float32)
        check(L.load().ae_laplacian_get_kernel(self._h, None, None, ptr(val)))
        return val

    def get_vectors(self):
        _, n, _ = self.info()
        out = [np.zeros(n, np.float32) for _ in range(4)]
        ms = C.c_float()
        check(L.load().ae_laplacian_get_vectors(self._h, ptr(out[0]), ptr(out[1]), ptr(out[2]), ptr(out[3]), C.byref(ms)))
        return dict(normalizer=out[0], normed_scales=out[1], q_density=out[2], beta_scales=out[3], mean_scale=ms.value)

    def do_svd(self, asked_dim=0, want_u=True):
        """GraphLaplacian::do_svd, src/graphlaplace.rs:127-134.  want_u=False leaves U on the device (as the embedder's
        own call does) and returns the spectrum only."""
        _, n, _ = self.info()
        s = np.zeros(20, np.float32)
        u = np.zeros((n, 20), np.float32) if want_u else None
        r = C.c_uint64()
        check(L.load().ae_laplacian_do_svd(self._h, ptr(s), ptr(u) if want_u else None, C.byref(r)))
        r = r.value
        return SvdResult(s[:r].copy(), np.ascontiguousarray(u.reshape(-1)[: n * r].reshape(n, r)) if want_u else None, None)


class DiffusionMaps:
    """DiffusionMaps, src/diffmaps.rs:254-271."""

    def __init__(self, params):
        self.params = params

    def laplacian_from_kgraph(self, kgraph, force_repr=0):
        """src/diffmaps.rs:397-422"""
        h = C.c_void_p()
        check(L.load().ae_dmap_laplacian_from_kgraph(kgraph.handle, C.byref(self.params.c()), force_repr, C.byref(h)))
        return GraphLaplacian(h)

    def embed_from_kgraph(self, kgraph, dparams=None):
        """src/diffmaps.rs:1047-1075"""
        dp = self.params if dparams is None else dparams
        n = kgraph.get_nb_nodes()
        y0 = np.zeros((n, dp.get_data_dim()), np.float32)
        rd = C.c_uint64()
        check(L.load().ae_dmap_embed_from_kgraph(kgraph.handle, C.byref(dp.c()), ptr(y0), C.byref(rd)))
        rd = rd.value
        return np.ascontiguousarray(y0.reshape(-1)[: n * rd].reshape(n, rd))


class Embedder(_Handle):
    """Embedder, src/embedder.rs:84-453."""

    _destroy = "ae_embedder_destroy"

    def __init__(self, kgraph, parameters):
        """Embedder::new, src/embedder.rs:107"""
        self.kgraph, self.hkgraph, self.parameters = kgraph, None, parameters
        h = C.c_void_p()
        check(L.load().ae_embedder_new(kgraph.handle, C.byref(parameters.c()), C.byref(h)))
        super().__init__(h)

    def set_comm(self, comm, exchanges_per_batch=4):
        """multi-GPU embedding: this process is one rank of `comm` (annembed_amd.dist.LibraryComm / HostMemComm); embed() then
        shards the CE loop over the ranks (ae_embedder_set_comm): the graph may be in any node order -- embed() partitions it by
        locality (get_partition_report) and returns the embedding in the caller's order.  parameters.ce_mode: AE_CE_AUTO / AE_CE_SLICED
        (the faithful time-sliced mode) or AE_CE_HOGWILD (the approximate rounds mode, by name)."""
        self._comm = comm  # keep the communicator alive
        check(L.load().ae_embedder_set_comm(self._h, comm._h if comm is not None else None, exchanges_per_batch))

    def get_partition_report(self):
        """the locality partition the last multi-GPU embed() applied (ae_embedder_get_partition_report)"""
        rep = L.CPartitionReport()
        check(L.load().ae_embedder_get_partition_report(self._h, C.byref(rep)))
        return _partition_report(rep)

    @classmethod
    def from_hkgraph(cls, graph_projection, parameters):
        """Embedder::from_hkgraph, src/embedder.rs:120"""
        o = cls.__new__(cls)
        o.kgraph, o.hkgraph, o.parameters = None, graph_projection, parameters
        h = C.c_void_p()
        check(L.load().ae_embedder_from_hkgraph(graph_projection.handle, C.byref(parameters.c()), C.byref(h)))
        _Handle.__init__(o, h)
        return o

    def embed(self):
        """Embedder::embed, src/embedder.rs:183 -- returns 1 (Ok(1)) or raises AnnembedError (Err(1))."""
        check(L.load().ae_embedder_embed(self._h))
        return 1

    def get_asked_dimension(self):
        return self.parameters.asked_dim

    def get_nb_nodes(self):
        v = C.c_uint64()
        check(L.load().ae_embedder_get_nb_nodes(self._h, C.byref(v)))
        return v.value

    def _out(self):
        return np.zeros((self.get_nb_nodes(), self.parameters.asked_dim), np.float32)

    def get_embedded(self):
        y = self._out()
        check(L.load().ae_embedder_get_embedded(self._h, ptr(y)))
        return y

    def get_embedded_reindexed(self, data_ids=None):
        y = self._out()
        ids = None if data_ids is None else _u64(data_ids)
        check(L.load().ae_embedder_get_embedded_reindexed(self._h, ptr(ids), ptr(y)))
        return y

    def get_initial_embedding(self):
        y = self._out()
        check(L.load().ae_embedder_get_initial_embedding(self._h, ptr(y)))
        return y

    def get_hubness(self):
        c = np.zeros(self.get_nb_nodes(), np.uint32)
        check(L.load().ae_embedder_get_hubness(self._h, ptr(c)))
        return c

    def get_cross_entropy(self):
        a, b = C.c_double(), C.c_double()
        check(L.load().ae_embedder_get_cross_entropy(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def get_quality_estimate_from_edge_length(self, nbng):
        """Embedder::get_quality_estimate_from_edge_length, src/embedder.rs:620-753 -> QualityReport"""
        n = self.get_nb_nodes()
        rep = L.CQualityReport()
        ratio, first = np.zeros(n, np.float64), np.zeros(n, np.float64)
        check(L.load().ae_embedder_get_quality_estimate_from_edge_length(self._h, int(nbng), C.byref(rep), ptr(ratio), ptr(first)))
        return QualityReport(rep, ratio, first)


QUALITY_PROBAS = (0.05, 0.25, 0.5, 0.75, 0.85, 0.95)


class QualityReport:
    """the figures get_quality_estimate_from_edge_length logs / prints (src/embedder.rs:690-731) and the two per-node
    vectors it dumps to continuity_ratio.csv / first_dist.csv (:735-747)"""

    def __init__(self, rep, ratio_by_node, first_dist):
        self.nb_nodes, self.nb_edges = int(rep.nb_nodes), int(rep.nb_edges)
        self.kgraph_nbng, self.nbng = int(rep.kgraph_nbng), int(rep.nbng)
        self.nb_without_match, self.mean_nbmatch = int(rep.nb_without_match), float(rep.mean_nbmatch)
        self.radii_quantiles = np.array(list(rep.radii_quantiles))
        self.ratio_quantiles = np.array(list(rep.ratio_quantiles))
        self.median_ratio, self.mean_ratio, self.quality = float(rep.median_ratio), float(rep.mean_ratio), float(rep.quality)
        self.ratio_by_node, self.first_dist = ratio_by_node, first_dist

    def __str__(self):  # the reference's text, :695-731, numbers in Rust's {:.2e} / {:.3e} form
        from .io import format_e
        fq = lambda q: " , ".join("%s : %s" % (p, format_e(v, 2)) for p, v in zip(QUALITY_PROBAS, q))
        return ("\n a guess at quality\n  neighbourhood size used in embedding : %d\n  nb neighbourhoods without a match : %d,  "
                "mean number of neighbours conserved when match : %s\n  embedded radii quantiles at %s\n\n statistics on "
                "conservation of neighborhood (of size nbng)\n  neighbourhood size used in target space : %d\n  quantiles at %s\n"
                "  neighborhood are conserved in radius multiplied by median  : %s, mean %s" %
                (self.kgraph_nbng, self.nb_without_match, format_e(self.mean_nbmatch, 3), fq(self.radii_quantiles), self.nbng,
                 fq(self.ratio_quantiles), format_e(self.median_ratio, 2), format_e(self.mean_ratio, 2)))


def quality_estimate_from_edge_length(kgraph, y, nbng):
    """stage-level entry: the same estimate for any embedding y (n x d, node order of the graph)"""
    y = _f32(y)
    n = y.shape[0]
    rep = L.CQualityReport()
    ratio, first = np.zeros(n, np.float64), np.zeros(n, np.float64)
    check(L.load().ae_quality_estimate_from_edge_length(kgraph.handle, ptr(y), y.shape[1], int(nbng), C.byref(rep), ptr(ratio), ptr(first)))
    return QualityReport(rep, ratio, first)
