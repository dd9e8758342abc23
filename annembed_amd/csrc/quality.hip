// quality.hip -- Embedder::get_quality_estimate_from_edge_length, src/embedder.rs:620-753, with
// get_transformed_kgraph (:478-522) and get_max_edge_length_embedded_kgraph (:527-554), on the device.
//
// What the reference computes, per node i of the original graph:
//   * transformed neighbourhood (:495-516): for the neighbours in their stored order a RUNNING MINIMUM of the
//     embedded L2 distances (`node_edge_length = distl2(..).min(node_edge_length)`, pushed for every edge), then
//     sorted ascending -- kept as is (the multiset is what the statistics below read);
//   * radius_i (:527-554, kgraph.rs:167-183): largest edge of node i in a kNN graph of size nbng built on the
//     EMBEDDED points.  The reference builds that graph with hnsw_rs (approximate, un-vendored: parity unpinned);
//     here it is the exact nbng-th neighbour distance (brute force on the device);
//   * matches_i = #{e : w_e <= radius_i}, ratio_e = w_e / radius_i (f64), node_ratio_i = mean_e ratio_e.
// Summary: nodes without a match, mean matches of the others, quantiles of the radii and of the ratios at
// 0.05 .25 .5 .75 .85 .95 (reference: CKMS sketches with eps = 0.01, i.e. any element within 1 % of the rank;
// here the exact order statistic at rank floor(q * count)), median and mean ratio.
#include "internal.h"
#include "linalg.h"

#include <rocprim/rocprim.hpp>

using namespace ae;

namespace ae {
void sort_pairs_u32_u32(uint32_t* d_keys_in, uint32_t* d_keys_out, uint32_t* d_vals_in, uint32_t* d_vals_out, uint64_t count, unsigned end_bit);  // svd.hip
}

namespace {

// thread per node: running minimum of the embedded edge lengths (embedder.rs:499-512), stored ascending
__global__ void transformed_edges_kernel(uint64_t n, const uint64_t* __restrict__ indptr, const uint32_t* __restrict__ nbr,
                                         const float* __restrict__ y, uint32_t dim, float* __restrict__ tw) {
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t b = indptr[i], e = indptr[i + 1];
    float running = 3.402823466e+38f;  // F::max_value()
    for (uint64_t x = b; x < e; x++) {
        const uint32_t j = nbr[x];
        float s = 0.f;  // distl2, embedder.rs:54-65: sum of squares in coordinate order, then sqrt
        for (uint32_t t = 0; t < dim; t++) {
            const float df = y[i * dim + t] - y[(uint64_t)j * dim + t];
            s += df * df;
        }
        running = fminf(sqrtf(s), running);
        tw[e - 1 - (x - b)] = running;  // non-increasing sequence written backwards = sorted ascending
    }
}

// thread per node: matches, ratios (f64), per-node mean ratio, first (smallest) transformed length
__global__ void quality_node_kernel(uint64_t n, const uint64_t* __restrict__ indptr, const float* __restrict__ tw,
                                    const uint64_t* __restrict__ e_indptr, const float* __restrict__ e_dist, const float* __restrict__ kth,
                                    double* __restrict__ radius,
                                    double* __restrict__ ratio, double* __restrict__ node_ratio, double* __restrict__ first_dist,
                                    unsigned long long* __restrict__ acc_u, double* __restrict__ acc_ratio) {
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    unsigned long long nomatch = 0, matches = 0;
    double sum = 0.;
    if (i < n) {
        double r = 0.;  // compute_max_edge, kgraph.rs:167-183
        if (kth) r = (double)kth[i];  // (grid path: the nbng-th neighbour distance itself)
        else
            for (uint64_t x = e_indptr[i]; x < e_indptr[i + 1]; x++) r = fmax(r, (double)e_dist[x]);
        radius[i] = r;
        const uint64_t b = indptr[i], e = indptr[i + 1];
        unsigned m = 0;
        double nr = 0.;
        for (uint64_t x = b; x < e; x++) {
            const double w = (double)tw[x];
            if (w <= r) m++;                 // :671-673
            const double q = w / r;          // :674-676
            ratio[x] = q;
            nr += q;
        }
        sum = nr;
        node_ratio[i] = nr / fmax(1.0, (double)(e - b));  // :679
        first_dist[i] = (double)tw[b];                     // :680
        nomatch = m == 0 ? 1ull : 0ull;
        matches = m;
    }
    // block reduction -> one atomic per block and quantity
    __shared__ unsigned long long s_a[256], s_b[256];
    __shared__ double s_c[256];
    s_a[threadIdx.x] = nomatch; s_b[threadIdx.x] = matches; s_c[threadIdx.x] = sum;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            s_a[threadIdx.x] += s_a[threadIdx.x + off];
            s_b[threadIdx.x] += s_b[threadIdx.x + off];
            s_c[threadIdx.x] += s_c[threadIdx.x + off];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        atomicAdd(&acc_u[0], s_a[0]);
        atomicAdd(&acc_u[1], s_b[0]);
        atomicAdd(acc_ratio, s_c[0]);
    }
}

// ---- exact distance to the nbng-th nearest neighbour in 2 / 3 embedded dimensions through a uniform grid: O(n) instead of
// the O(n^2) brute force (1.65 M points: seconds -> tens of milliseconds).  The reference rebuilds an hnsw_rs index on the
// embedded points and reads each node's largest edge (embedder.rs:527-554, kgraph.rs:167-183); that index is approximate and
// un-vendored -- here the radius is exact, and bit-identical to the brute-force producer's: F = f32 sum over the coordinates,
// in order, of (y_i[t] - y_j[t])^2 without fma, radius = sqrtf of the nbng-th smallest F over j != i (knn.hip).
// Points are binned into G^dim cells and sorted by cell (rocPRIM); one wave per query scans the block of cells within r
// cells of its own (r = 1, 2, ... until the block holds nbng other points AND the nbng-th distance is no larger than the
// distance from the query to the nearest face of the block that is not a face of the whole grid -- then no point outside
// the block can be nearer), and selects the nbng-th smallest F by a radix select over the float bits.
struct GridArgs {
    const float* ys;            // coordinates, cell-sorted order
    const uint32_t* perm;       // sorted position -> node
    const uint32_t* cell_start; // ncells + 1
    uint64_t n;
    uint32_t dim, k, G;
    float lo[3], w[3], inv_w[3], slack;
    float* kth;                 // [node]
    unsigned int* fallback_count;
};
constexpr int kGridCap = 1024;  // candidate distances staged in LDS per wave

__global__ void grid_bbox_kernel(const float* __restrict__ y, uint64_t n, uint32_t dim, float* __restrict__ part /* [blocks][6] */) {
    __shared__ float smin[256][3], smax[256][3];
    float mn[3] = {3.4e38f, 3.4e38f, 3.4e38f}, mx[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        for (uint32_t t = 0; t < dim; t++) { const float v = y[i * dim + t]; mn[t] = fminf(mn[t], v); mx[t] = fmaxf(mx[t], v); }
    for (int t = 0; t < 3; t++) { smin[threadIdx.x][t] = mn[t]; smax[threadIdx.x][t] = mx[t]; }
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off)
            for (int t = 0; t < 3; t++) {
                smin[threadIdx.x][t] = fminf(smin[threadIdx.x][t], smin[threadIdx.x + off][t]);
                smax[threadIdx.x][t] = fmaxf(smax[threadIdx.x][t], smax[threadIdx.x + off][t]);
            }
        __syncthreads();
    }
    if (threadIdx.x == 0)
        for (int t = 0; t < 3; t++) { part[blockIdx.x * 6 + t] = smin[0][t]; part[blockIdx.x * 6 + 3 + t] = smax[0][t]; }
}
__device__ __forceinline__ uint32_t grid_coord(float v, float lo, float inv_w, uint32_t G) {
    const float f = (v - lo) * inv_w;
    const int c = (int)f;
    return (uint32_t)(c < 0 ? 0 : (c >= (int)G ? (int)G - 1 : c));
}
__global__ void grid_keys_kernel(const float* __restrict__ y, GridArgs a, uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    uint32_t key = 0;
    for (int t = (int)a.dim - 1; t >= 0; t--) key = key * a.G + grid_coord(y[i * a.dim + t], a.lo[t], a.inv_w[t], a.G);  // x fastest
    keys[i] = key;
    vals[i] = (uint32_t)i;
}
__global__ void grid_cell_start_kernel(const uint32_t* __restrict__ keys, uint64_t n, uint32_t ncells, uint32_t* __restrict__ cell_start) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c > ncells) return;
    uint64_t lo = 0, hi = n;
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if (keys[mid] < c) lo = mid + 1;
        else hi = mid;
    }
    cell_start[c] = (uint32_t)lo;
}
__global__ void grid_gather_kernel(const float* __restrict__ y, const uint32_t* __restrict__ perm, uint64_t n, uint32_t dim, float* __restrict__ ys) {
    const uint64_t p = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (p >= n) return;
    for (uint32_t t = 0; t < dim; t++) ys[p * dim + t] = y[(uint64_t)perm[p] * dim + t];
}
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
// F of the definition: sequential f32 sum of squares, no fma
__device__ __forceinline__ float grid_f(const float* q, const float* p, uint32_t dim) {
    float f = 0.f;
    for (uint32_t t = 0; t < dim; t++) { const float df = __fsub_rn(q[t], p[t]); f = __fadd_rn(f, __fmul_rn(df, df)); }
    return f;
}
__global__ void __launch_bounds__(256) grid_kth_kernel(GridArgs a) {
    __shared__ uint32_t s_f[4][kGridCap];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint64_t p = blockIdx.x * 4ull + wv;  // query = sorted position p (wave-uniform)
    if (p >= a.n) return;
    float q[3] = {0.f, 0.f, 0.f};
    uint32_t cq[3] = {0, 0, 0};
    for (uint32_t t = 0; t < a.dim; t++) { q[t] = a.ys[p * a.dim + t]; cq[t] = grid_coord(q[t], a.lo[t], a.inv_w[t], a.G); }
    const uint32_t G = a.G, k = a.k;
    float result = 0.f;
    for (uint32_t r = 1;; r++) {
        // block of cells, clipped; margin = distance to the nearest face that is not a face of the whole grid
        uint32_t c0[3] = {0, 0, 0}, c1[3] = {0, 0, 0};
        float margin = 3.4e38f;
        bool whole = true;
        for (uint32_t t = 0; t < a.dim; t++) {
            c0[t] = cq[t] >= r ? cq[t] - r : 0u;
            c1[t] = cq[t] + r < G ? cq[t] + r : G - 1u;
            if (c0[t] > 0u) { margin = fminf(margin, q[t] - (a.lo[t] + (float)c0[t] * a.w[t])); whole = false; }
            if (c1[t] < G - 1u) { margin = fminf(margin, (a.lo[t] + (float)(c1[t] + 1u) * a.w[t]) - q[t]); whole = false; }
        }
        margin -= a.slack;
        // segments: for every (y, z) row of the block the cells c0[0] .. c1[0] are consecutive cell ids
        const uint32_t ny = a.dim >= 2 ? c1[1] - c0[1] + 1u : 1u, nz = a.dim >= 3 ? c1[2] - c0[2] + 1u : 1u;
        const uint32_t nseg = ny * nz;
        uint32_t count = 0;
        for (uint32_t sgi = (uint32_t)lane; sgi < nseg; sgi += 64u) {
            const uint32_t cy = c0[1] + sgi % ny, cz = c0[2] + sgi / ny;
            const uint32_t base = (cz * G + cy) * G;
            count += a.cell_start[base + c1[0] + 1u] - a.cell_start[base + c0[0]];
        }
        count = wave_sum_u32(count);
        if (count < k + 1u && !whole) continue;  // (the query itself is in the block)
        // radix select of the k-th smallest F over the block, the query excluded (rank k - 1 among the others)
        const bool staged = count <= (uint32_t)kGridCap;
        auto for_each_candidate = [&](auto&& fn) {  // fn(slot, F bits): lanes share the points of a segment
            uint32_t slot0 = 0;
            for (uint32_t sgi = 0; sgi < nseg; sgi++) {
                const uint32_t cy = c0[1] + sgi % ny, cz = c0[2] + sgi / ny;
                const uint32_t base = (cz * G + cy) * G;
                const uint32_t b = a.cell_start[base + c0[0]], e = a.cell_start[base + c1[0] + 1u];
                for (uint32_t x = b + (uint32_t)lane; x < e; x += 64u) {
                    const float f = x == (uint32_t)p ? __uint_as_float(0x7F800000u) : grid_f(q, a.ys + (uint64_t)x * a.dim, a.dim);
                    fn(slot0 + (x - b), __float_as_uint(f));
                }
                slot0 += e - b;
            }
        };
        if (staged) {
            for_each_candidate([&](uint32_t slot, uint32_t bits) { s_f[wv][slot] = bits; });
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        uint32_t prefix = 0, want = k - 1u;  // rank (0-based) still wanted among the values matching the prefix
        for (int bit = 31; bit >= 0; bit--) {
            const uint32_t mask = bit == 31 ? 0u : (0xFFFFFFFFu << (bit + 1));
            uint32_t zeros = 0;
            if (staged) {
                for (uint32_t x = (uint32_t)lane; x < count; x += 64u) {
                    const uint32_t v = s_f[wv][x];
                    zeros += ((v & mask) == prefix && ((v >> bit) & 1u) == 0u) ? 1u : 0u;
                }
            } else {
                for_each_candidate([&](uint32_t, uint32_t v) { zeros += ((v & mask) == prefix && ((v >> bit) & 1u) == 0u) ? 1u : 0u; });
            }
            zeros = wave_sum_u32(zeros);
            if (want >= zeros) { want -= zeros; prefix |= 1u << bit; }
        }
        if (!staged && lane == 0) atomicAdd(a.fallback_count, 1u);
        const float kth = sqrtf(__uint_as_float(prefix));
        __builtin_amdgcn_wave_barrier();
        if (whole || kth <= margin) { result = kth; break; }
    }
    if (lane == 0) a.kth[a.perm[p]] = result;
}

// d_kth[i] = distance from point i to its k-th nearest other point (dim 2 or 3); returns false when the grid path does not apply
bool grid_kth_distance(const float* d_y, uint64_t n, uint32_t dim, uint32_t k, float* d_kth) {
    if ((dim != 2 && dim != 3) || n < 4096 || n >= (1ull << 31) || debug_knob("AE_QUALITY_BRUTE")) return false;
    DevBuf<float> part(256 * 6);
    hipLaunchKernelGGL(grid_bbox_kernel, dim3(256), dim3(256), 0, stream(), d_y, n, dim, part.p);
    check_launch("grid_bbox");
    std::vector<float> hp = part.to_host();
    GridArgs a;
    memset(&a, 0, sizeof(a));
    float ext_max = 0.f;
    const double occ = std::max(4.0, 0.75 * (double)k);  // points per cell on average: the 3^dim block holds ~7-20 k
    uint32_t G = dim == 2 ? (uint32_t)std::ceil(std::sqrt((double)n / occ)) : (uint32_t)std::ceil(std::cbrt((double)n / occ));
    G = std::max(1u, std::min(G, dim == 2 ? 8192u : 400u));
    for (uint32_t t = 0; t < dim; t++) {
        float lo = 3.4e38f, hi = -3.4e38f;
        for (int b = 0; b < 256; b++) { lo = std::min(lo, hp[b * 6 + t]); hi = std::max(hi, hp[b * 6 + 3 + t]); }
        if (!(hi > lo)) hi = lo + 1.f;
        a.lo[t] = lo;
        a.w[t] = (hi - lo) / (float)G * (1.f + 1e-6f);
        a.inv_w[t] = 1.f / a.w[t];
        ext_max = std::max(ext_max, std::max(std::fabs(lo), std::fabs(hi)));
    }
    a.slack = 8e-7f * ext_max * 4.f;  // rounding of a cell face, of the binning and of F against the real distance
    a.n = n; a.dim = dim; a.k = k; a.G = G;
    const uint64_t ncells = dim == 2 ? (uint64_t)G * G : (uint64_t)G * G * G;
    DevBuf<uint32_t> k0(n), k1(n), v0(n), v1(n), cell_start(ncells + 1), fb(1);
    DevBuf<float> ys(n * dim);
    fb.zero();
    hipLaunchKernelGGL(grid_keys_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, stream(), d_y, a, k0.p, v0.p);
    unsigned bits = 1;
    while (bits < 32 && (ncells >> bits)) bits++;
    sort_pairs_u32_u32(k0.p, k1.p, v0.p, v1.p, n, bits);
    hipLaunchKernelGGL(grid_cell_start_kernel, dim3(blocks_for(ncells + 1, 256)), dim3(256), 0, stream(), (const uint32_t*)k1.p, n, (uint32_t)ncells, cell_start.p);
    hipLaunchKernelGGL(grid_gather_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, stream(), d_y, (const uint32_t*)v1.p, n, dim, ys.p);
    a.ys = ys.p; a.perm = v1.p; a.cell_start = cell_start.p; a.kth = d_kth; a.fallback_count = reinterpret_cast<unsigned int*>(fb.p);
    hipLaunchKernelGGL(grid_kth_kernel, dim3(blocks_for(n, 4)), dim3(256), 0, stream(), a);
    check_launch("grid_kth");
    sync();
    return true;
}

const double kQuantiles[6] = {0.05, 0.25, 0.5, 0.75, 0.85, 0.95};

// exact order statistics of a device array (sorted copy by rocPRIM radix sort)
void quantiles_of(const double* d_vals, uint64_t count, double* out6) {
    DevBuf<double> sorted(count);
    size_t tmp_bytes = 0;
    AE_HIP(rocprim::radix_sort_keys(nullptr, tmp_bytes, d_vals, sorted.p, count, 0, 64, stream()));
    DevBuf<char> tmp(tmp_bytes ? tmp_bytes : 1);
    AE_HIP(rocprim::radix_sort_keys(tmp.p, tmp_bytes, d_vals, sorted.p, count, 0, 64, stream()));
    for (int q = 0; q < 6; q++) {
        uint64_t rank = (uint64_t)(kQuantiles[q] * (double)count);
        if (rank >= count) rank = count - 1;
        AE_HIP(hipMemcpyAsync(&out6[q], sorted.p + rank, sizeof(double), hipMemcpyDeviceToHost, stream()));
    }
    sync();
}

}  // namespace

extern "C" {

int32_t ae_quality_estimate_from_edge_length(const ae_kgraph* g, const float* y, uint32_t dim, uint32_t nbng, ae_quality_report* rep,
                                             double* ratio_by_node, double* first_dist) {
    return guard([&] {
        require_device();
        if (!g || !y || !rep || dim == 0) fail(AE_ERR_INVALID_ARG, "null argument");
        if (nbng == 0 || nbng >= g->n) fail(AE_ERR_INVALID_ARG, "nbng must be in 1 .. nb_nodes - 1");
        const uint64_t n = g->n;
        // radius_i = distance to the nbng-th nearest embedded point (exact; the reference's is an hnsw_rs approximation):
        // uniform grid in 2 / 3 dimensions, brute-force kNN graph of the embedded points otherwise
        DevBuf<float> dy;
        dy.alloc(n * dim);
        dy.upload(y, n * dim);
        DevBuf<float> kth(n);
        std::unique_ptr<ae_kgraph> eg;
        const bool grid = grid_kth_distance(dy.p, n, dim, nbng, kth.p);
        if (!grid) {
            ae_kgraph* eg_raw = nullptr;
            int32_t rc = ae_kgraph_bruteforce_l2(y, n, dim, nbng, &eg_raw);
            if (rc != AE_OK) throw Error(rc, ae_last_error_message());
            eg.reset(eg_raw);
        }
        DevBuf<float> tw(g->nnz);
        hipLaunchKernelGGL(transformed_edges_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, stream(), n, g->indptr.p, g->nbr.p, dy.p, dim, tw.p);
        check_launch("transformed_edges");
        DevBuf<double> radius(n), ratio(g->nnz), node_ratio(n), first(n), acc_ratio(1);
        DevBuf<unsigned long long> acc_u(2);
        acc_u.zero();
        acc_ratio.zero();
        hipLaunchKernelGGL(quality_node_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, stream(), n, g->indptr.p, tw.p,
                           grid ? (const uint64_t*)nullptr : (const uint64_t*)eg->indptr.p, grid ? (const float*)nullptr : (const float*)eg->dist.p,
                           grid ? (const float*)kth.p : (const float*)nullptr, radius.p, ratio.p, node_ratio.p, first.p, acc_u.p, acc_ratio.p);
        check_launch("quality_node");
        unsigned long long hu[2];
        double hr = 0.;
        acc_u.download(hu, 2);
        acc_ratio.download(&hr, 1);
        memset(rep, 0, sizeof(*rep));
        rep->nb_nodes = n;
        rep->nb_edges = g->nnz;
        rep->kgraph_nbng = g->max_nbng;
        rep->nbng = nbng;
        rep->nb_without_match = hu[0];
        rep->mean_nbmatch = (n > hu[0]) ? (double)hu[1] / (double)(n - hu[0]) : 0.;  // :688-689
        quantiles_of(radius.p, n, rep->radii_quantiles);
        quantiles_of(ratio.p, g->nnz, rep->ratio_quantiles);
        rep->median_ratio = rep->ratio_quantiles[2];
        rep->mean_ratio = hr / (double)g->nnz;  // :677-678, :730
        rep->quality = 0.;                      // the reference returns Some(0.) (:630, :751)
        if (ratio_by_node) node_ratio.download(ratio_by_node, n);
        if (first_dist) first.download(first_dist, n);
    });
}

}  // extern "C"
