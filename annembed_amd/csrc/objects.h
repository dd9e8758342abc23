// objects.h -- the opaque handles of the C ABI (device-resident state).
#pragma once
#include <memory>

#include "common.h"

// KGraph, src/fromhnsw/kgraph.rs:109-120, as a device CSR: rows sorted by increasing distance.
struct ae_kgraph {
    uint64_t n = 0;
    uint32_t max_nbng = 0;
    uint64_t nnz = 0;
    uint32_t uniform_k = 0;  // k when every row has exactly k entries, else 0
    ae::DevBuf<uint64_t> indptr;
    ae::DevBuf<uint32_t> nbr;
    ae::DevBuf<float> dist;
};

// KGraphProjection, src/fromhnsw/kgproj.rs:35-44 (accessors :376-410)
struct ae_kgraph_projection {
    const ae_kgraph* small_graph = nullptr;
    const ae_kgraph* large_graph = nullptr;
    ae::DevBuf<uint32_t> proj_node;
    ae::DevBuf<float> proj_dist;
    float median_dist = 1.f;  // 0.5 quantile of proj_dist over nodes >= n_small (kgproj.rs:403-410)
};

// NodeParams, src/tools/nodeparam.rs:111-114: proba aligned with the graph's nbr[], scale per node
struct ae_node_params {
    const ae_kgraph* g = nullptr;
    ae::DevBuf<float> proba;
    ae::DevBuf<float> scale;
};

// MatRepr, src/tools/matrepr.rs:23-32
struct ae_matrepr {
    bool is_csr = false;
    uint64_t nrows = 0, ncols = 0, nnz = 0;
    ae::DevBuf<uint64_t> indptr;  // CSR
    ae::DevBuf<uint32_t> indices;
    ae::DevBuf<float> values;     // CSR values or dense row-major
    // lazily built transpose (CSR of A^T) for A^T * Y products
    std::unique_ptr<ae_matrepr> transpose;
    bool symmetric = false;       // A == A^T known by construction (graph laplacian)
};

// GraphLaplacian, src/graphlaplace.rs:21-35
struct ae_laplacian {
    uint64_t n = 0;
    ae_matrepr sym_kernel;
    ae::DevBuf<float> normalizer;     // sqrt(degrees)
    ae::DevBuf<float> normed_scales;  // local scale / mean
    ae::DevBuf<float> q_density;      // may be empty (beta == 0)
    ae::DevBuf<float> beta_scales;    // may be empty
    float mean_scale = 0.f;
};
