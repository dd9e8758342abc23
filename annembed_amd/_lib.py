"""ctypes binding of libannembed_hip.so (the C ABI declared in include/annembed_hip.h).

torch is imported first on purpose: torch bundles its own libamdhip64.so; loading it before our
library makes both share one HIP runtime in the process (same SONAME), which the RCCL path needs.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libannembed_hip.so")

AE_OK = 0
AE_ERR_INVALID_ARG = 1
AE_ERR_NO_DEVICE = 2
AE_ERR_ISOLATED_NODE = 3
AE_ERR_PROBA_RANGE = 4
AE_ERR_SVD = 5
AE_ERR_SPECTRUM = 6
AE_ERR_EMBED = 7
AE_ERR_STATE = 8
AE_ERR_BETA = 9
AE_ERR_OOM = 10

AE_CE_HOGWILD = 0
AE_CE_SEQUENTIAL = 1
AE_CE_SAMPLE_RACY = 2
AE_CE_EVENT = 3
AE_CE_AUTO = 4
AE_CE_SLICED = 5
AE_CE_ORDERED = 6
AE_SAMPLER_ROWCDF = 0
AE_SAMPLER_ALIAS = 1
AE_PRECISION_F64 = 0
AE_PRECISION_F32 = 1


class AnnembedError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("annembed_hip error %d: %s" % (code, msg))
        self.code = code


class CEmbedderParams(C.Structure):
    _fields_ = [
        ("asked_dim", C.c_uint64), ("dmap_init", C.c_uint8), ("beta", C.c_double), ("b", C.c_double),
        ("scale_rho", C.c_double), ("grad_step", C.c_double), ("nb_sampling_by_edge", C.c_uint64),
        ("nb_grad_batch", C.c_uint64), ("grad_factor", C.c_uint64), ("hierarchy_layer", C.c_uint64),
        ("hubness_weighting", C.c_uint8), ("seed", C.c_uint64), ("ce_mode", C.c_uint32), ("ce_sampler", C.c_uint32),
        ("ce_precision", C.c_uint32),
    ]


class CQualityReport(C.Structure):
    _fields_ = [
        ("nb_nodes", C.c_uint64), ("nb_edges", C.c_uint64), ("kgraph_nbng", C.c_uint32), ("nbng", C.c_uint32),
        ("nb_without_match", C.c_uint64), ("mean_nbmatch", C.c_double), ("radii_quantiles", C.c_double * 6),
        ("ratio_quantiles", C.c_double * 6), ("median_ratio", C.c_double), ("mean_ratio", C.c_double), ("quality", C.c_double),
    ]


class CPartitionReport(C.Structure):
    _fields_ = [("components", C.c_uint64), ("splits", C.c_uint64), ("cross_mass", C.c_double), ("cross_mass_worst_rank", C.c_double),
                ("imbalance", C.c_double)]


class CDiffusionParams(C.Structure):
    _fields_ = [
        ("asked_dim", C.c_uint64), ("alfa", C.c_float), ("beta", C.c_float), ("epsil", C.c_float), ("t", C.c_float),
        ("has_t", C.c_uint8), ("gnbn", C.c_uint64), ("has_gnbn", C.c_uint8),
    ]


_vp = C.c_void_p
_u64 = C.c_uint64
_u32 = C.c_uint32
_i32 = C.c_int32
_f32 = C.c_float
_f64 = C.c_double
_u8 = C.c_uint8
_P = C.POINTER

# name -> argtypes ; every function returns int32 except the two string getters.
SIGNATURES = {
    "ae_device_count": [_P(_i32)],
    "ae_set_device": [_i32],
    "ae_synchronize": [],
    "ae_get_stream": [_P(_vp)],
    "ae_set_summation_order": [_u32],
    "ae_embedder_params_default": [_P(CEmbedderParams)],
    "ae_diffusion_params_new": [_P(CDiffusionParams), _u64, _f32, _u8, _u64, _u8],
    "ae_diffusion_params_set_alfa": [_P(CDiffusionParams), _f32],
    "ae_diffusion_params_set_beta": [_P(CDiffusionParams), _f32],
    "ae_diffusion_params_set_epsil": [_P(CDiffusionParams), _f32],
    "ae_kgraph_create": [_vp, _vp, _vp, _u64, _u32, _P(_vp)],
    "ae_kgraph_from_ragged": [_vp, _vp, _vp, _vp, _u64, _u32, _P(_vp), _vp],
    "ae_kgraph_destroy": [_vp],
    "ae_kgraph_get_nb_nodes": [_vp, _P(_u64)],
    "ae_kgraph_get_max_nbng": [_vp, _P(_u32)],
    "ae_kgraph_get_nb_edges": [_vp, _P(_u64)],
    "ae_kgraph_get_neighbours": [_vp, _vp, _vp, _vp],
    "ae_kgraph_fill_l2_distances": [_vp, _vp, _u64],
    "ae_kgraph_bruteforce_l2": [_vp, _u64, _u64, _u32, _P(_vp)],
    "ae_kgraph_bruteforce_l2_grouped": [_vp, _u64, _u64, _u32, _vp, _u32, _P(_vp), _vp],
    "ae_kgraph_hubness": [_vp, _vp],
    "ae_kgraph_projection_create": [_vp, _vp, _vp, _vp, _P(_vp)],
    "ae_kgraph_projection_destroy": [_vp],
    "ae_to_proba_edges": [_vp, _f32, _f32, _P(_vp)],
    "ae_node_params_from_host": [_vp, _vp, _vp, _P(_vp)],
    "ae_node_params_destroy": [_vp],
    "ae_node_params_get": [_vp, _vp, _vp],
    "ae_node_params_perplexity": [_vp, _vp],
    "ae_dmap_laplacian_from_kgraph": [_vp, _P(CDiffusionParams), _i32, _P(_vp)],
    "ae_laplacian_destroy": [_vp],
    "ae_laplacian_info": [_vp, _P(_i32), _P(_u64), _P(_u64)],
    "ae_laplacian_get_kernel": [_vp, _vp, _vp, _vp],
    "ae_laplacian_get_vectors": [_vp, _vp, _vp, _vp, _vp, _P(_f32)],
    "ae_laplacian_do_svd": [_vp, _vp, _vp, _P(_u64)],
    "ae_dmap_embed_from_kgraph": [_vp, _P(CDiffusionParams), _vp, _P(_u64)],
    "ae_matrepr_from_csr": [_vp, _vp, _vp, _u64, _u64, _P(_vp)],
    "ae_matrepr_from_dense": [_vp, _u64, _u64, _P(_vp)],
    "ae_matrepr_destroy": [_vp],
    "ae_subspace_iteration": [_vp, _u64, _u64, _vp, _P(_u64)],
    "ae_svd_approx_rank": [_vp, _u64, _u64, _vp, _vp, _vp, _P(_u64)],
    "ae_adaptative_range_finder": [_vp, _f64, _u64, _u64, _vp, _P(_u64)],
    "ae_svd_approx_epsil": [_vp, _f64, _u64, _u64, _vp, _vp, _vp, _P(_u64)],
    "ae_transpose_dense_mult": [_vp, _vp, _u64, _vp],
    "ae_set_data_box": [_vp, _u64, _u64, _f32],
    "ae_entropy_optim_create": [_vp, _vp, _P(CEmbedderParams), _vp, _vp, _u64, _u64, _P(_vp)],
    "ae_entropy_optim_destroy": [_vp],
    "ae_entropy_optim_get_nb_edges": [_vp, _P(_u64)],
    "ae_entropy_optim_get_ce_mode": [_vp, _P(C.c_uint32)],
    "ae_entropy_optim_dataflow_time": [_vp, _P(C.c_double), _P(_u64)],
    "ae_projection_init": [_vp, _vp, _u64, _u64, _u64, _vp],
    "ae_comm_unique_id": [_vp],
    "ae_comm_init": [C.c_int32, C.c_int32, _vp, _P(_vp)],
    "ae_comm_init_hostmem": [C.c_int32, C.c_int32, C.c_char_p, _u64, _P(_vp)],
    "ae_comm_destroy": [_vp],
    "ae_comm_all_reduce_sum": [_vp, _P(C.c_double)],
    "ae_entropy_optim_set_comm": [_vp, _vp, C.c_uint32],
    "ae_entropy_optim_comm_bytes": [_vp, _P(_u64)],
    "ae_entropy_optim_slice_info": [_vp, _P(C.c_uint32), _P(C.c_double), _P(C.c_uint32), _P(C.c_uint32)],
    "ae_entropy_optim_slice_hub_info": [_vp, _P(C.c_uint32), _P(_f64)],
    "ae_entropy_optim_slice_form": [_vp, _P(C.c_uint32)],
    "ae_entropy_optim_ce": [_vp, _P(_f64)],
    "ae_entropy_optim_gradient_iteration": [_vp, _u64, _f64, _u64],
    "ae_entropy_optim_gradient_iteration_lockstep": [_vp, C.c_uint32, _vp, _f64, _u64, C.c_uint32],
    "ae_entropy_optim_plan": [_vp, _u64, _u64, _u64, _vp, _vp],
    "ae_entropy_optim_samples_drawn": [_vp, _P(_u64), _P(_u32)],
    "ae_entropy_optim_get_scales": [_vp, _vp],
    "ae_entropy_optim_get_embedded": [_vp, _vp],
    "ae_entropy_optim_device_coords": [_vp, _P(_vp), _P(_u64), _P(_u64)],
    "ae_entropy_optim_kernel_time": [_vp, _P(_f64), _P(_u64)],
    "ae_entropy_optimize": [_vp, _vp, _P(CEmbedderParams), _vp, _vp, _P(_f64), _P(_f64)],
    "ae_embedder_new": [_vp, _P(CEmbedderParams), _P(_vp)],
    "ae_embedder_from_hkgraph": [_vp, _P(CEmbedderParams), _P(_vp)],
    "ae_embedder_destroy": [_vp],
    "ae_embedder_set_comm": [_vp, _vp, C.c_uint32],
    "ae_embedder_embed": [_vp],
    "ae_kgraph_partition": [_vp, _vp, _vp, _u64, _u32, _vp, _vp, _P(CPartitionReport)],
    "ae_kgraph_permuted": [_vp, _vp, _P(_vp)],
    "ae_embedder_get_partition_report": [_vp, _P(CPartitionReport)],
    "ae_embedder_get_nb_nodes": [_vp, _P(_u64)],
    "ae_embedder_get_embedded": [_vp, _vp],
    "ae_embedder_get_embedded_reindexed": [_vp, _vp, _vp],
    "ae_embedder_get_initial_embedding": [_vp, _vp],
    "ae_embedder_get_hubness": [_vp, _vp],
    "ae_embedder_get_cross_entropy": [_vp, _P(_f64), _P(_f64)],
    "ae_quality_estimate_from_edge_length": [_vp, _vp, _u32, _u32, _P(CQualityReport), _vp, _vp],
    "ae_embedder_get_quality_estimate_from_edge_length": [_vp, _u32, _P(CQualityReport), _vp, _vp],
}
STRING_GETTERS = ["ae_last_error_message", "ae_last_warning_message", "ae_version"]

_lib = None


def load():
    """Loads the shared library; raises if it has not been built (there is no fallback path)."""
    global _lib
    if _lib is not None:
        return _lib
    try:
        import torch  # noqa: F401  (shares the HIP runtime, see module docstring)
    except Exception:  # pragma: no cover - torch is optional for pure ctypes use
        pass
    if not os.path.exists(LIB_PATH):
        raise ImportError("libannembed_hip.so is not built: run `python -m annembed_amd.build` (needs hipcc); "
                          "annembed_amd has no CPU fallback")
    lib = C.CDLL(LIB_PATH)
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = _i32
    for name in STRING_GETTERS:
        getattr(lib, name).restype = C.c_char_p
        getattr(lib, name).argtypes = []
    _lib = lib
    return lib


def check(rc):
    if rc != AE_OK:
        raise AnnembedError(rc, load().ae_last_error_message().decode())
    w = load().ae_last_warning_message()
    if w:
        import warnings
        warnings.warn("annembed_hip: " + w.decode(), RuntimeWarning, stacklevel=3)


def ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)
