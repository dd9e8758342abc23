"""annembed_amd -- MI355X (gfx950) implementation of annembed's embedding hot path.

The package is a thin host-side mirror of the reference's Rust API (KGraph / EmbedderParams /
Embedder / DiffusionMaps / SvdApprox) over the C ABI of libannembed_hip.so.  All compute runs on the GPU
(hand-written HIP kernels; rocPRIM sort / scan as primitives beside them); there is no CPU fallback: importing the API without the built library, or
calling it without a GPU, fails loudly.
"""
from ._lib import (AE_CE_AUTO, AE_CE_EVENT, AE_CE_HOGWILD, AE_CE_ORDERED, AE_CE_SAMPLE_RACY, AE_CE_SEQUENTIAL, AE_CE_SLICED, AE_PRECISION_F32, AE_PRECISION_F64, AE_SAMPLER_ALIAS, AE_SAMPLER_ROWCDF, AnnembedError, LIB_PATH,  # noqa: F401
                   load)
from .api import (DiffusionMaps, DiffusionParams, Embedder, EmbedderParams, EntropyOptim, GraphLaplacian, KGraph,  # noqa: F401
                  KGraphProjection, MatRepr, NodeParams, QualityReport, RangeApprox, RangePrecision, RangeRank, SvdApprox, SvdResult, entropy_optimize,
                  adaptative_range_finder_matrep, quality_estimate_from_edge_length, set_data_box, subspace_iteration, to_proba_edges, transpose_dense_mult_csr, set_summation_order)

__all__ = [
    "KGraph", "KGraphProjection", "NodeParams", "EmbedderParams", "DiffusionParams", "Embedder", "EntropyOptim",
    "DiffusionMaps", "GraphLaplacian", "MatRepr", "RangeRank", "SvdApprox", "SvdResult", "to_proba_edges", "set_data_box",
    "entropy_optimize", "subspace_iteration", "transpose_dense_mult_csr", "QualityReport", "quality_estimate_from_edge_length", "RangePrecision", "RangeApprox",
    "adaptative_range_finder_matrep",
    "set_summation_order", "AnnembedError", "load",
]
