// core.hip -- library plumbing (errors, stream, parameter PODs) and the KGraph object (a1).
#include <algorithm>
#include <mutex>
#include <unordered_map>

#include "objects.h"
#include "linalg.h"

namespace ae {

static thread_local std::string g_last_error, g_last_warning;
static thread_local int g_api_depth = 0;
void set_last_error(const std::string& s) { g_last_error = s; }
// (entry points nest -- ae_embedder_embed calls others --: only an OUTERMOST call clears what an earlier one said)
void set_last_warning(const std::string& s) { g_last_warning = s; }
int& api_depth() { return g_api_depth; }

static std::mutex g_stream_mu;
static hipStream_t g_streams[64] = {};

void require_device() {
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        fail(AE_ERR_NO_DEVICE, "no HIP device available (libannembed_hip is GPU-only, there is no CPU fallback): %s",
             e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
}

static thread_local hipStream_t t_stream_override = nullptr;
StreamScope::StreamScope(hipStream_t s) : prev(t_stream_override) { t_stream_override = s; }
StreamScope::~StreamScope() { t_stream_override = prev; }

hipStream_t stream() {
    if (t_stream_override) return t_stream_override;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) fail(AE_ERR_NO_DEVICE, "hipGetDevice failed (no HIP device?)");
    std::lock_guard<std::mutex> lk(g_stream_mu);
    if (dev < 0 || dev >= 64) fail(AE_ERR_INVALID_ARG, "device index %d out of range", dev);
    if (!g_streams[dev]) AE_HIP(hipStreamCreateWithFlags(&g_streams[dev], hipStreamNonBlocking));
    return g_streams[dev];
}

void* pool_alloc(size_t bytes) {
    static bool ready[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) fail(AE_ERR_NO_DEVICE, "hipGetDevice failed (no HIP device?)");
    if (!ready[dev]) {  // keep freed blocks in the pool instead of returning them to the driver at every synchronisation
        hipMemPool_t pool;
        AE_HIP(hipDeviceGetDefaultMemPool(&pool, dev));
        uint64_t threshold = ~0ull;
        AE_HIP(hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &threshold));
        ready[dev] = true;
    }
    void* p = nullptr;
    AE_HIP(hipMallocAsync(&p, bytes, stream()));
    return p;
}
void pool_free(void* p) {
    try {
        (void)hipFreeAsync(p, stream());
    } catch (...) {  // process teardown: the device is gone
    }
}

}  // namespace ae

using namespace ae;

// ---------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------
// validates the KGraph invariants: non-empty rows (kgraph.rs:520-537), ascending distances
// (kgraph.rs:508-509), indices < n and != the row's own node (kgraph.rs:501), row length <= max_nbng.  err[0] = code, err[1] = node.
__global__ void kgraph_validate_kernel(uint64_t n, const uint64_t* __restrict__ indptr, const uint32_t* __restrict__ nbr,
                                       const float* __restrict__ dist, uint32_t max_nbng, unsigned long long* err,
                                       unsigned int* nonuniform, uint32_t k0) {
    uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t b = indptr[i], e = indptr[i + 1];
    unsigned long long code = 0;
    if (e < b) code = AE_ERR_INVALID_ARG;
    else if (e == b) code = AE_ERR_ISOLATED_NODE;
    else if (e - b > max_nbng) code = AE_ERR_INVALID_ARG;
    else {
        for (uint64_t x = b; x < e; x++) {
            if (nbr[x] >= n || nbr[x] == i) code = AE_ERR_INVALID_ARG;  // (a node is not its own neighbour: assert of kgraph.rs:501)
            if (x > b && dist[x] < dist[x - 1]) code = AE_ERR_INVALID_ARG;
            if (!(dist[x] >= 0.f)) code = AE_ERR_INVALID_ARG;  // also rejects NaN
        }
    }
    if (e - b != k0) atomicOr(nonuniform, 1u);
    if (code) atomicMin(err, (code << 48) | i);  // keep the smallest (code, node)
}

// per row: keep the nbng smallest of a ragged candidate list, ordered by (distance, list position)
// -- sort_unstable_by + truncate of kgraph.rs:508,:539 with ties resolved by list order.
__global__ void kgraph_select_kernel(uint64_t n, const uint64_t* __restrict__ row_ptr, const uint32_t* __restrict__ cand_idx,
                                     const float* __restrict__ cand_dist, const uint32_t* __restrict__ slot_of_row,
                                     uint32_t nbng, const uint64_t* __restrict__ out_indptr, uint32_t* __restrict__ out_nbr,
                                     float* __restrict__ out_dist) {
    uint64_t p = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (p >= n) return;
    uint64_t b = row_ptr[p], len = row_ptr[p + 1] - b;
    uint64_t o = out_indptr[slot_of_row[p]];
    uint32_t keep = len < nbng ? (uint32_t)len : nbng;
    // insertion into the sorted prefix held in the output row (stable: strict < moves only)
    uint32_t cnt = 0;
    for (uint64_t t = 0; t < len; t++) {
        float d = cand_dist[b + t];
        uint32_t id = cand_idx[b + t];
        if (cnt == keep && !(d < out_dist[o + cnt - 1])) continue;
        uint32_t pos = cnt < keep ? cnt : keep - 1;
        while (pos > 0 && d < out_dist[o + pos - 1]) {
            out_dist[o + pos] = out_dist[o + pos - 1];
            out_nbr[o + pos] = out_nbr[o + pos - 1];
            pos--;
        }
        out_dist[o + pos] = d;
        out_nbr[o + pos] = id;
        if (cnt < keep) cnt++;
    }
}

// Hubness::new, src/fromhnsw/hubness.rs:51-67: in-degree histogram
__global__ void hubness_kernel(uint64_t nnz, const uint32_t* __restrict__ nbr, uint32_t* __restrict__ counts) {
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t e = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; e < nnz; e += stride) atomicAdd(&counts[nbr[e]], 1u);
}

// dist[e] = ||x_i - x_nbr[e]||_2 ; one wave per source node, lanes stride the coordinates
__global__ void __launch_bounds__(256) l2_edges_kernel(uint64_t n, const uint64_t* __restrict__ indptr,
                                                       const uint32_t* __restrict__ nbr, const float* __restrict__ x,
                                                       uint64_t dim, float* __restrict__ dist) {
    const int lane = threadIdx.x & 63;
    uint64_t wave = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 6;
    uint64_t nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t i = wave; i < n; i += nwaves) {
        const float* xi = x + i * dim;
        for (uint64_t e = indptr[i]; e < indptr[i + 1]; e++) {
            const float* xj = x + (uint64_t)nbr[e] * dim;
            float acc = 0.f;
            for (uint64_t c = lane; c < dim; c += 64) {
                float d = xi[c] - xj[c];
                acc += d * d;
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
            if (lane == 0) dist[e] = sqrtf(acc);
        }
    }
}

// re-sort each (short) row by distance, stable
__global__ void row_sort_kernel(uint64_t n, const uint64_t* __restrict__ indptr, uint32_t* __restrict__ nbr,
                                float* __restrict__ dist) {
    uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t b = indptr[i], e = indptr[i + 1];
    for (uint64_t x = b + 1; x < e; x++) {
        float d = dist[x];
        uint32_t id = nbr[x];
        uint64_t y = x;
        while (y > b && d < dist[y - 1]) {
            dist[y] = dist[y - 1];
            nbr[y] = nbr[y - 1];
            y--;
        }
        dist[y] = d;
        nbr[y] = id;
    }
}

// exact kNN by brute force: one thread per query, candidates staged through LDS in tiles.  Defines the result of the
// kNN producer (knn.hip reaches the same rows on the matrix cores and uses this kernel for the rows it cannot certify).
// `rows` (optional) lists the queries; blockIdx.y selects a chunk of the points, the per-chunk lists are merged by
// knn_merge_chunks_kernel (chunks ascending + strict comparisons: among equal distances the smaller index wins).
// Rectangular form (knn.hip's grouped producer): query t is place p = rows[t] (or t) of a list whose entry p is point qrows[p] (or
// q_begin + p); the points are the range [p_begin, n).
template <int TILE>
__global__ void __launch_bounds__(256) bruteforce_knn_kernel(const float* __restrict__ x, uint64_t p_begin, uint64_t n, uint64_t dim, uint32_t k,
                                                             const uint32_t* __restrict__ rows, const uint32_t* __restrict__ qrows, uint64_t q_begin,
                                                             uint64_t nrows, uint64_t chunk_len,
                                                             uint32_t* __restrict__ out_nbr, float* __restrict__ out_d2, const uint32_t* __restrict__ orig) {
    extern __shared__ float tile[];  // TILE x dimchunk
    constexpr int DC = 32;           // coordinates per chunk
    const uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    const bool active = t < nrows;
    const uint64_t place = active ? (rows ? rows[t] : t) : 0;
    const uint64_t q = active ? (qrows ? (uint64_t)qrows[place] : q_begin + place) : p_begin;
    const uint64_t c_begin = p_begin + blockIdx.y * chunk_len, c_end = c_begin + chunk_len < n ? c_begin + chunk_len : n;
    float* best_d = out_d2 + (t * gridDim.y + blockIdx.y) * k;  // kept sorted ascending in global (L2 resident), k small
    uint32_t* best_i = out_nbr + (t * gridDim.y + blockIdx.y) * k;
    if (active)
        for (uint32_t s = 0; s < k; s++) { best_d[s] = INFINITY; best_i[s] = 0xFFFFFFFFu; }
    for (uint64_t c0 = c_begin; c0 < c_end; c0 += TILE) {
        float acc[TILE];
#pragma unroll
        for (int s = 0; s < TILE; s++) acc[s] = 0.f;
        for (uint64_t d0 = 0; d0 < dim; d0 += DC) {
            __syncthreads();
            for (int idx = threadIdx.x; idx < TILE * DC; idx += blockDim.x) {
                int s = idx / DC, d = idx % DC;
                uint64_t c = c0 + s;
                tile[idx] = (c < c_end && d0 + d < dim) ? x[c * dim + d0 + d] : 0.f;
            }
            __syncthreads();
            if (active) {
                float xq[DC];
#pragma unroll
                for (int d = 0; d < DC; d++) xq[d] = (d0 + d < dim) ? x[q * dim + d0 + d] : 0.f;
#pragma unroll
                for (int s = 0; s < TILE; s++) {
                    float a = acc[s];
#pragma unroll
                    for (int d = 0; d < DC; d++) {
                        float df = xq[d] - tile[s * DC + d];
                        a += df * df;
                    }
                    acc[s] = a;
                }
            }
        }
        if (active) {
#pragma unroll
            for (int s = 0; s < TILE; s++) {
                uint64_t c = c0 + s;
                if (c >= c_end || c == q) continue;
                float d = acc[s];
                if (!orig) {   // points in the caller's order: scanning ascending, a strict comparison keeps the smaller index among equals
                    if (!(d < best_d[k - 1])) continue;
                    uint32_t pos = k - 1;
                    while (pos > 0 && d < best_d[pos - 1]) {
                        best_d[pos] = best_d[pos - 1];
                        best_i[pos] = best_i[pos - 1];
                        pos--;
                    }
                    best_d[pos] = d;
                    best_i[pos] = (uint32_t)c;
                } else {       // points reordered internally (knn.hip's grouped producer): ties by the caller's ids
                    const uint32_t oc = orig[c];
                    auto before = [&](uint32_t slot) { return d < best_d[slot] || (d == best_d[slot] && best_i[slot] != 0xFFFFFFFFu && oc < orig[best_i[slot]]); };
                    if (!before(k - 1)) continue;
                    uint32_t pos = k - 1;
                    while (pos > 0 && before(pos - 1)) {
                        best_d[pos] = best_d[pos - 1];
                        best_i[pos] = best_i[pos - 1];
                        pos--;
                    }
                    best_d[pos] = d;
                    best_i[pos] = (uint32_t)c;
                }
            }
        }
    }
}
// thread per query: the k smallest (d2, index) of its per-chunk lists, written as row `q` of the graph (sqrt applied)
__global__ void knn_merge_chunks_kernel(const uint32_t* __restrict__ rows, const uint32_t* __restrict__ qrows, uint64_t q_begin, uint64_t nrows, uint32_t chunks,
                                        uint32_t k, const uint32_t* __restrict__ part_i, const float* __restrict__ part_d,
                                        uint32_t* __restrict__ nbr, float* __restrict__ dist, int out_by_list, int raw, const uint32_t* __restrict__ orig) {
    const uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (t >= nrows) return;
    const uint64_t place = rows ? rows[t] : t;
    const uint64_t q = out_by_list ? place : (qrows ? (uint64_t)qrows[place] : q_begin + place);
    float* bd = dist + q * k;
    uint32_t* bi = nbr + q * k;
    for (uint32_t s = 0; s < k; s++) { bd[s] = INFINITY; bi[s] = 0xFFFFFFFFu; }
    for (uint64_t e = t * chunks * k; e < (t + 1) * chunks * k; e++) {
        const float d = part_d[e];
        const uint32_t pi = part_i[e];
        if (pi == 0xFFFFFFFFu) continue;
        auto before = [&](uint32_t slot) {
            return d < bd[slot] || (orig && d == bd[slot] && bi[slot] != 0xFFFFFFFFu && orig[pi] < orig[bi[slot]]);
        };
        if (!before(k - 1)) continue;
        uint32_t pos = k - 1;
        while (pos > 0 && before(pos - 1)) {
            bd[pos] = bd[pos - 1];
            bi[pos] = bi[pos - 1];
            pos--;
        }
        bd[pos] = d;
        bi[pos] = pi;
    }
    if (!raw)
        for (uint32_t s = 0; s < k; s++) bd[s] = sqrtf(bd[s]);
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
static void finish_kgraph(ae_kgraph* g) {
    // validate + uniformity
    DevBuf<unsigned long long> err(1);
    DevBuf<unsigned int> nonuni(1);
    unsigned long long init = ~0ull;
    err.upload(&init, 1);
    nonuni.zero();
    uint32_t k0 = g->n ? (uint32_t)(g->nnz / g->n) : 0;
    hipLaunchKernelGGL(kgraph_validate_kernel, dim3(blocks_for(g->n, 256)), dim3(256), 0, stream(), g->n, g->indptr.p,
                       g->nbr.p, g->dist.p, g->max_nbng, err.p, nonuni.p, k0);
    check_launch("kgraph_validate");
    unsigned long long herr;
    unsigned int hnon;
    err.download(&herr, 1);
    nonuni.download(&hnon, 1);
    if (herr != ~0ull) {
        int32_t code = (int32_t)(herr >> 48);
        uint64_t node = herr & ((1ull << 48) - 1);
        if (code == AE_ERR_ISOLATED_NODE)
            fail(code, "node rank %llu has no neighbour (graph would not be connected; kgraph.rs:520-537)",
                 (unsigned long long)node);
        fail(code, "invalid KGraph row %llu: rows must be non-empty, sorted by increasing distance, <= max_nbng long, indices < n and never the row's own node",
             (unsigned long long)node);
    }
    g->uniform_k = (hnon == 0 && g->n && g->nnz == (uint64_t)k0 * g->n) ? k0 : 0;
}

namespace ae {
// exact brute force for the listed places of a query list against the points [p_begin, p_end) (see bruteforce_knn_kernel); few rows
// are spread over many point chunks
void bruteforce_knn_rect(const float* d_x, uint64_t dim, uint32_t k, const uint32_t* d_qrows, uint64_t q_begin, const uint32_t* d_places, uint64_t nrows,
                         uint64_t p_begin, uint64_t p_end, uint32_t* d_nbr, float* d_dist, bool out_by_list, bool raw, const uint32_t* d_orig) {
    if (nrows == 0) return;
    constexpr int TILE = 32;
    const uint64_t np = p_end - p_begin;
    const uint64_t row_blocks = blocks_for(nrows, 256);
    uint32_t chunks = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(1, 2048 / row_blocks), std::max<uint64_t>(1, np / 128));
    chunks = std::min<uint32_t>(chunks, 1024);
    const uint64_t chunk_len = (np + chunks - 1) / chunks;
    DevBuf<uint32_t> part_i;
    DevBuf<float> part_d;
    part_i.alloc_pooled(nrows * chunks * k);
    part_d.alloc_pooled(nrows * chunks * k);
    hipLaunchKernelGGL((bruteforce_knn_kernel<TILE>), dim3((unsigned)row_blocks, chunks), dim3(256), TILE * 32 * sizeof(float), stream(), d_x, p_begin, p_end,
                       dim, k, d_places, d_qrows, q_begin, nrows, chunk_len, part_i.p, part_d.p, d_orig);
    check_launch("bruteforce_knn");
    hipLaunchKernelGGL(knn_merge_chunks_kernel, dim3((unsigned)row_blocks), dim3(256), 0, stream(), d_places, d_qrows, q_begin, nrows, chunks, k,
                       (const uint32_t*)part_i.p, (const float*)part_d.p, d_nbr, d_dist, out_by_list ? 1 : 0, raw ? 1 : 0, d_orig);
    check_launch("knn_merge_chunks");
}
// exact brute force for the listed rows (nullptr: rows 0..nrows-1) against all n points
void bruteforce_knn_rows(const float* d_x, uint64_t n, uint64_t dim, uint32_t k, const uint32_t* d_rows, uint64_t nrows,
                         uint32_t* d_nbr, float* d_dist) {
    bruteforce_knn_rect(d_x, dim, k, nullptr, 0, d_rows, nrows, 0, n, d_nbr, d_dist, false, false, nullptr);
}
}  // namespace ae

extern "C" {

const char* ae_last_error_message(void) { return ae::g_last_error.c_str(); }
const char* ae_last_warning_message(void) { return ae::g_last_warning.c_str(); }
const char* ae_version(void) { return "annembed_hip 0.1.0 gfx950"; }

int32_t ae_device_count(int32_t* count) {
    return guard([&] {
        if (!count) fail(AE_ERR_INVALID_ARG, "count is NULL");
        int c = 0;
        if (hipGetDeviceCount(&c) != hipSuccess) c = 0;
        *count = c;
    });
}
int32_t ae_set_device(int32_t device) {
    return guard([&] {
        require_device();
        AE_HIP(hipSetDevice(device));
    });
}
int32_t ae_synchronize(void) {
    return guard([&] {
        require_device();
        sync();
    });
}
int32_t ae_set_summation_order(uint32_t order) {
    return guard([&] {
        if (order != AE_SUM_REFERENCE_ORDER && order != AE_SUM_TREE) fail(AE_ERR_INVALID_ARG, "unknown summation order %u", order);
        set_tree_sums_default(order == AE_SUM_TREE);
    });
}
int32_t ae_get_stream(void** s) {
    return guard([&] {
        require_device();
        if (!s) fail(AE_ERR_INVALID_ARG, "stream out pointer is NULL");
        *s = (void*)stream();
    });
}

// EmbedderParams::default, src/embedparams.rs:107-132
int32_t ae_embedder_params_default(ae_embedder_params* p) {
    return guard([&] {
        if (!p) fail(AE_ERR_INVALID_ARG, "params is NULL");
        memset(p, 0, sizeof(*p));
        p->asked_dim = 2;
        p->dmap_init = 1;
        p->beta = 1.;
        p->b = 1.;
        p->scale_rho = 1.;
        p->grad_step = 2.;
        p->nb_sampling_by_edge = 10;
        p->nb_grad_batch = 20;
        p->grad_factor = 4;
        p->hierarchy_layer = 0;
        p->hubness_weighting = 0;
        p->seed = kDefaultSeed;
        p->ce_mode = AE_CE_AUTO;
        p->ce_sampler = AE_SAMPLER_ROWCDF;
        p->ce_precision = AE_PRECISION_F64;
    });
}

// DiffusionParams::new, src/diffmaps.rs:95-105
int32_t ae_diffusion_params_new(ae_diffusion_params* p, uint64_t asked_dim, float t, uint8_t has_t, uint64_t gnbn,
                                uint8_t has_gnbn) {
    return guard([&] {
        if (!p) fail(AE_ERR_INVALID_ARG, "params is NULL");
        memset(p, 0, sizeof(*p));
        p->asked_dim = asked_dim;
        p->alfa = 0.5f;
        p->beta = -0.1f;
        p->epsil = 2.0f;
        p->t = t;
        p->has_t = has_t ? 1 : 0;
        p->gnbn = gnbn;
        p->has_gnbn = has_gnbn ? 1 : 0;
    });
}
// set_alfa, src/diffmaps.rs:122-136: clamp to [-2, 1]
int32_t ae_diffusion_params_set_alfa(ae_diffusion_params* p, float alfa) {
    return guard([&] {
        if (!p) fail(AE_ERR_INVALID_ARG, "params is NULL");
        p->alfa = std::min(std::max(alfa, -2.f), 1.f);
    });
}
// set_beta, src/diffmaps.rs:140-148: accepted only inside [-1.01, 0]
int32_t ae_diffusion_params_set_beta(ae_diffusion_params* p, float beta) {
    return guard([&] {
        if (!p) fail(AE_ERR_INVALID_ARG, "params is NULL");
        if (beta >= -1.01f && beta <= 0.f) p->beta = beta;
    });
}
// set_epsil, src/diffmaps.rs:151-160: clamp to [0.5, 4]
int32_t ae_diffusion_params_set_epsil(ae_diffusion_params* p, float epsil) {
    return guard([&] {
        if (!p) fail(AE_ERR_INVALID_ARG, "params is NULL");
        p->epsil = std::max(std::min(epsil, 4.0f), 0.5f);
    });
}

int32_t ae_kgraph_create(const uint64_t* indptr, const uint32_t* nbr, const float* dist, uint64_t n, uint32_t max_nbng,
                         ae_kgraph** out) {
    return guard([&] {
        require_device();
        if (!indptr || !nbr || !dist || !out || n == 0 || max_nbng == 0) fail(AE_ERR_INVALID_ARG, "null argument or empty graph");
        if (n >= 0xFFFFFFFFull) fail(AE_ERR_INVALID_ARG, "n must fit in u32 node indices");
        if (indptr[0] != 0) fail(AE_ERR_INVALID_ARG, "indptr[0] must be 0");
        std::unique_ptr<ae_kgraph> g(new ae_kgraph);
        g->n = n;
        g->max_nbng = max_nbng;
        g->nnz = indptr[n];
        if (g->nnz == 0) fail(AE_ERR_ISOLATED_NODE, "graph has no edge");
        g->indptr.alloc(n + 1);
        g->indptr.upload(indptr, n + 1);
        g->nbr.alloc(g->nnz);
        g->nbr.upload(nbr, g->nnz);
        g->dist.alloc(g->nnz);
        g->dist.upload(dist, g->nnz);
        finish_kgraph(g.get());
        *out = g.release();
    });
}

int32_t ae_kgraph_from_ragged(const uint64_t* point_id, const uint64_t* row_ptr, const uint64_t* nbr_data_id,
                              const float* nbr_dist, uint64_t n, uint32_t nbng, ae_kgraph** out,
                              uint64_t* data_id_of_idx_out) {
    return guard([&] {
        require_device();
        if (!point_id || !row_ptr || !nbr_data_id || !nbr_dist || !out || n == 0 || nbng == 0)
            fail(AE_ERR_INVALID_ARG, "null argument or empty graph");
        if (n >= 0xFFFFFFFFull) fail(AE_ERR_INVALID_ARG, "n must fit in u32 node indices");
        uint64_t total = row_ptr[n];
        // IndexSet<DataId> insertion order (kgraph.rs:489,:500): inherently sequential, done on the host
        std::unordered_map<uint64_t, uint32_t> set;
        set.reserve(n * 2);
        std::vector<uint64_t> order;
        order.reserve(n);
        auto insert_full = [&](uint64_t id) -> uint32_t {
            auto it = set.find(id);
            if (it != set.end()) return it->second;
            uint32_t idx = (uint32_t)order.size();
            set.emplace(id, idx);
            order.push_back(id);
            return idx;
        };
        std::vector<uint32_t> cand_idx(total), slot(n);
        std::vector<uint32_t> row_len(n, 0);
        for (uint64_t p = 0; p < n; p++) {
            uint32_t index = insert_full(point_id[p]);
            if (index >= n) fail(AE_ERR_INVALID_ARG, "more distinct DataIds than points");
            slot[p] = index;
            uint64_t len = row_ptr[p + 1] - row_ptr[p];
            if (len == 0)
                fail(AE_ERR_ISOLATED_NODE, "kgraph_from_hnsw_all: graph will not be connected, isolated point %llu",
                     (unsigned long long)point_id[p]);
            for (uint64_t t = row_ptr[p]; t < row_ptr[p + 1]; t++) {
                cand_idx[t] = insert_full(nbr_data_id[t]);
                if (cand_idx[t] >= n) fail(AE_ERR_INVALID_ARG, "more distinct DataIds than points");
                if (cand_idx[t] == index) fail(AE_ERR_INVALID_ARG, "self edge (kgraph.rs:502 asserts index != neighbour)");
            }
            row_len[index] = (uint32_t)std::min<uint64_t>(len, nbng);
        }
        std::unique_ptr<ae_kgraph> g(new ae_kgraph);
        g->n = n;
        g->max_nbng = nbng;
        std::vector<uint64_t> indptr(n + 1, 0);
        for (uint64_t i = 0; i < n; i++) {
            if (row_len[i] == 0) fail(AE_ERR_ISOLATED_NODE, "node %llu has no neighbour list", (unsigned long long)i);
            indptr[i + 1] = indptr[i] + row_len[i];
        }
        g->nnz = indptr[n];
        g->indptr.alloc(n + 1);
        g->indptr.upload(indptr.data(), n + 1);
        g->nbr.alloc(g->nnz);
        g->dist.alloc(g->nnz);
        DevBuf<uint64_t> d_row_ptr;
        d_row_ptr.alloc(n + 1);
        d_row_ptr.upload(row_ptr, n + 1);
        DevBuf<uint32_t> d_cand, d_slot;
        d_cand.alloc(total);
        d_cand.upload(cand_idx.data(), total);
        d_slot.alloc(n);
        d_slot.upload(slot.data(), n);
        DevBuf<float> d_cdist;
        d_cdist.alloc(total);
        d_cdist.upload(nbr_dist, total);
        hipLaunchKernelGGL(kgraph_select_kernel, dim3(blocks_for(n, 128)), dim3(128), 0, stream(), n, d_row_ptr.p, d_cand.p,
                           d_cdist.p, d_slot.p, nbng, g->indptr.p, g->nbr.p, g->dist.p);
        check_launch("kgraph_select");
        finish_kgraph(g.get());
        if (data_id_of_idx_out) memcpy(data_id_of_idx_out, order.data(), sizeof(uint64_t) * order.size());
        *out = g.release();
    });
}

int32_t ae_kgraph_destroy(ae_kgraph* g) {
    return guard([&] { delete g; });
}
int32_t ae_kgraph_get_nb_nodes(const ae_kgraph* g, uint64_t* n) {
    return guard([&] {
        if (!g || !n) fail(AE_ERR_INVALID_ARG, "null argument");
        *n = g->n;
    });
}
int32_t ae_kgraph_get_max_nbng(const ae_kgraph* g, uint32_t* k) {
    return guard([&] {
        if (!g || !k) fail(AE_ERR_INVALID_ARG, "null argument");
        *k = g->max_nbng;
    });
}
int32_t ae_kgraph_get_nb_edges(const ae_kgraph* g, uint64_t* nnz) {
    return guard([&] {
        if (!g || !nnz) fail(AE_ERR_INVALID_ARG, "null argument");
        *nnz = g->nnz;
    });
}
int32_t ae_kgraph_get_neighbours(const ae_kgraph* g, uint64_t* indptr, uint32_t* nbr, float* dist) {
    return guard([&] {
        if (!g) fail(AE_ERR_INVALID_ARG, "null graph");
        if (indptr) g->indptr.download(indptr, g->n + 1);
        if (nbr) g->nbr.download(nbr, g->nnz);
        if (dist) g->dist.download(dist, g->nnz);
    });
}

int32_t ae_kgraph_fill_l2_distances(ae_kgraph* g, const float* x, uint64_t dim) {
    return guard([&] {
        require_device();
        if (!g || !x || dim == 0) fail(AE_ERR_INVALID_ARG, "null argument");
        DevBuf<float> dx;
        dx.alloc(g->n * dim);
        dx.upload(x, g->n * dim);
        hipLaunchKernelGGL(l2_edges_kernel, dim3(grid_cap(g->n * 64, 256)), dim3(256), 0, stream(), g->n, g->indptr.p, g->nbr.p,
                           dx.p, dim, g->dist.p);
        check_launch("l2_edges");
        hipLaunchKernelGGL(row_sort_kernel, dim3(blocks_for(g->n, 128)), dim3(128), 0, stream(), g->n, g->indptr.p, g->nbr.p,
                           g->dist.p);
        check_launch("row_sort");
        sync();
    });
}

int32_t ae_kgraph_bruteforce_l2(const float* x, uint64_t n, uint64_t dim, uint32_t nbng, ae_kgraph** out) {
    return guard([&] {
        require_device();
        if (!x || !out || n < 2 || dim == 0 || nbng == 0 || nbng >= n) fail(AE_ERR_INVALID_ARG, "bad arguments");
        DevBuf<float> dx;
        dx.alloc(n * dim);
        dx.upload(x, n * dim);
        std::unique_ptr<ae_kgraph> g(new ae_kgraph);
        g->n = n;
        g->max_nbng = nbng;
        g->nnz = n * nbng;
        std::vector<uint64_t> indptr(n + 1);
        for (uint64_t i = 0; i <= n; i++) indptr[i] = i * nbng;
        g->indptr.alloc(n + 1);
        g->indptr.upload(indptr.data(), n + 1);
        g->nbr.alloc(g->nnz);
        g->dist.alloc(g->nnz);
        // matrix-core path (knn.hip) for the usual neighbourhood sizes; AE_KNN_LEGACY=1 keeps the plain kernel (A/B)
        if (nbng + 8 <= 64 && !debug_knob("AE_KNN_LEGACY")) {
            const uint64_t fell_back = knn_mfma(dx.p, n, dim, nbng, g->nbr.p, g->dist.p);
            if (debug_knob("AE_CE_PROF")) fprintf(stderr, "KNN rows recomputed by the brute-force fallback: %llu of %llu\n",
                                              (unsigned long long)fell_back, (unsigned long long)n);
        } else {
            bruteforce_knn_rows(dx.p, n, dim, nbng, nullptr, n, g->nbr.p, g->dist.p);
        }
        finish_kgraph(g.get());
        *out = g.release();
    });
}

int32_t ae_kgraph_bruteforce_l2_grouped(const float* x, uint64_t n, uint64_t dim, uint32_t nbng, const uint64_t* bounds, uint32_t groups, ae_kgraph** out,
                                        uint64_t* stats3) {
    return guard([&] {
        require_device();
        if (!x || !out || !bounds || n < 2 || dim == 0 || nbng == 0 || nbng >= n || nbng + 8 > 64) fail(AE_ERR_INVALID_ARG, "bad arguments (nbng <= 56)");
        DevBuf<float> dx;
        dx.alloc(n * dim);
        dx.upload(x, n * dim);
        std::unique_ptr<ae_kgraph> g(new ae_kgraph);
        g->n = n;
        g->max_nbng = nbng;
        g->nnz = n * nbng;
        std::vector<uint64_t> indptr(n + 1);
        for (uint64_t i = 0; i <= n; i++) indptr[i] = i * nbng;
        g->indptr.alloc(n + 1);
        g->indptr.upload(indptr.data(), n + 1);
        g->nbr.alloc(g->nnz);
        g->dist.alloc(g->nnz);
        uint64_t st[3] = {0, 0, 0};
        knn_grouped(dx.p, n, dim, nbng, bounds, groups, g->nbr.p, g->dist.p, st);
        if (stats3) { stats3[0] = st[0]; stats3[1] = st[1]; stats3[2] = st[2]; }
        finish_kgraph(g.get());
        *out = g.release();
    });
}

int32_t ae_kgraph_hubness(const ae_kgraph* g, uint32_t* counts) {
    return guard([&] {
        require_device();
        if (!g || !counts) fail(AE_ERR_INVALID_ARG, "null argument");
        DevBuf<uint32_t> c(g->n);
        c.zero();
        hipLaunchKernelGGL(hubness_kernel, dim3(grid_cap(g->nnz, 256)), dim3(256), 0, stream(), g->nnz, g->nbr.p, c.p);
        check_launch("hubness");
        c.download(counts, g->n);
    });
}

int32_t ae_kgraph_projection_create(const ae_kgraph* small, const ae_kgraph* large, const uint32_t* proj_node,
                                    const float* proj_dist, ae_kgraph_projection** out) {
    return guard([&] {
        require_device();
        if (!small || !large || !proj_node || !proj_dist || !out) fail(AE_ERR_INVALID_ARG, "null argument");
        if (small->n > large->n) fail(AE_ERR_INVALID_ARG, "small graph larger than large graph");
        for (uint64_t i = small->n; i < large->n; i++)
            if (proj_node[i] >= small->n) fail(AE_ERR_INVALID_ARG, "proj_node[%llu] outside the small graph", (unsigned long long)i);
        std::unique_ptr<ae_kgraph_projection> p(new ae_kgraph_projection);
        p->small_graph = small;
        p->large_graph = large;
        p->proj_node.alloc(large->n);
        p->proj_node.upload(proj_node, large->n);
        p->proj_dist.alloc(large->n);
        p->proj_dist.upload(proj_dist, large->n);
        // get_projection_distance_quant().query(0.5) (kgproj.rs:403-410, embedder.rs:255): exact lower median
        std::vector<float> pd(proj_dist + small->n, proj_dist + large->n);
        if (!pd.empty()) {
            size_t mid = (pd.size() - 1) / 2;
            std::nth_element(pd.begin(), pd.begin() + mid, pd.end());
            p->median_dist = pd[mid];
        }
        sync();
        *out = p.release();
    });
}
int32_t ae_kgraph_projection_destroy(ae_kgraph_projection* p) {
    return guard([&] { delete p; });
}

}  // extern "C"
