// dmap.hip -- a3-a6, a9, a10: diffusion-map initialisation (src/diffmaps.rs, src/graphlaplace.rs).
//
//   compute_dmap_nodeparams   diffmaps.rs:752-849  (get_dist_l2_from_node :1020-1043,
//                                                    scales_to_nodeparams :691-745, build_node_param :590-675)
//   kernel0_to_density        diffmaps.rs:855-952
//   compute_laplacian         diffmaps.rs:427-587
//   do_svd                    graphlaplace.rs:127-134
//   embed_from_laplacian      diffmaps.rs:1145-1243
//   set_data_box              embedder.rs:1376-1408
//
// The reference symmetrises through a serial HashMap<(i,j),w> (:513-544, :898-929).  Here the
// (k+1) N stored keys emit 2 triplets each, the triplet keys (row<<32|col) are radix-sorted ONCE
// (the structure is the same for the density pass and the laplacian pass), and every later step
// is a segmented pass over the sorted order: deterministic, no atomics on floats.

#include <cmath>

#include "linalg.h"
// (after <cstring>: rocprim's texture iterator calls the host memset)
#include <rocprim/rocprim.hpp>

using namespace ae;

namespace ae {
void sort_pairs_u64_u32(uint64_t* d_keys_in, uint64_t* d_keys_out, uint32_t* d_vals_in, uint32_t* d_vals_out, uint64_t count);
void rowptr_from_sorted_keys(const uint64_t* d_keys, uint64_t nnz, uint64_t nrows, uint64_t* d_rowptr);
void full_svd_leading(ae_matrepr& a, uint32_t rank, std::vector<float>& s, DevBuf<float>& u);
void direct_svd_rank(ae_matrepr& a, uint64_t rank, uint64_t nbiter, std::vector<float>& s, DevBuf<float>& u);
}  // namespace ae

namespace {

// get_dist_l2_from_node, diffmaps.rs:1020-1043
__global__ void dmap_local_scale_kernel(uint64_t n, const uint64_t* __restrict__ indptr, const float* __restrict__ dist, uint32_t nbng,
                                        float* __restrict__ local) {
    uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t b = indptr[i], len = indptr[i + 1] - b;
    float d2 = 0.f;
    for (uint64_t m = 0; m < len && m < nbng; m++) d2 += dist[b + m] * dist[b + m];  // :1032-1036
    local[i] = len ? sqrtf(d2 / (float)len) : 0.f;                                  // :1038-1042
}
// diffmaps.rs:806-816: zero scales -> mean, normed = local / mean
__global__ void dmap_fix_scales_kernel(uint64_t n, float* __restrict__ local, float mean, float* __restrict__ normed) {
    uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = local[i];
    if (v <= 0.f) { v = mean; local[i] = v; }
    normed[i] = v / mean;
}

// build_node_param, diffmaps.rs:590-675 with remap_weight of :831-834.  Row i of the kernel is
// [(i, kself[i])] ++ [(nbr[e], kval[e])].  scales == nullptr means the constant `cscale` (beta == 0, :846).
__global__ void __launch_bounds__(256) dmap_kernel_rows_kernel(uint64_t n, const uint64_t* __restrict__ indptr,
                                                               const uint32_t* __restrict__ nbr, const float* __restrict__ dist,
                                                               const float* __restrict__ scales, float cscale, float epsil,
                                                               float* __restrict__ kself, float* __restrict__ kval,
                                                               unsigned long long* low_count, unsigned long long* err) {
    uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t b = indptr[i], len = indptr[i + 1] - b;
    if (len == 0) {  // :611-615
        atomicMin(err, ((unsigned long long)AE_ERR_ISOLATED_NODE << 48) | i);
        return;
    }
    bool all_equal = false;
    long last = -1;
    for (long m = (long)len - 1; m >= 0; m--)  // :618-620
        if (dist[b + m] > 0.f) { last = m; break; }
    if (last >= 0) { if (dist[b + last] <= dist[b]) all_equal = true; } else all_equal = true;  // :622-630
    if (all_equal) {
        const float p = 1.0f / (float)(len + 1);  // :642
        kself[i] = p;
        for (uint64_t m = 0; m < len; m++) kval[b + m] = p;
        return;
    }
    const float from_scale = scales ? scales[i] : cscale;
    unsigned low = 0;
    for (uint64_t m = 0; m < len; m++) {
        const float to_scale = scales ? scales[nbr[b + m]] : cscale;
        const float local_scale = sqrtf(to_scale * from_scale);  // :656
        const float x = (dist[b + m] - 0.f) / (epsil * local_scale);
        float w = expf(-(x * x));  // powf(x, 2.0f) :832 ; exp :833
        if (w < kProbaMin) { w = kProbaMin; low++; }  // :658-661
        kval[b + m] = w;
    }
    kself[i] = 1.0f;  // :667-668
    if (low) atomicAdd(low_count, (unsigned long long)low);
}

__global__ void edge_src_kernel(uint64_t n, const uint64_t* __restrict__ indptr, uint32_t* __restrict__ src) {
    uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    for (uint64_t e = indptr[i]; e < indptr[i + 1]; e++) src[e] = (uint32_t)i;
}

// triplet keys for source key t in [0, nnz + n): t < nnz is edge t, else the self edge of node t - nnz.
// Triplet 2t = (row i, col j), 2t+1 = (row j, col i)  (diffmaps.rs:535-543).  A stored edge that is
// shadowed by a later duplicate of the same (i,j) in its row is dead (HashMap::insert overwrites).
__global__ void triplet_keys_kernel(uint64_t n, uint64_t nnz, const uint64_t* __restrict__ indptr, const uint32_t* __restrict__ nbr,
                                    const uint32_t* __restrict__ src, uint64_t* __restrict__ keys, uint32_t* __restrict__ payload) {
    uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (t >= nnz + n) return;
    uint64_t i, j;
    bool dead = false;
    if (t < nnz) {
        i = src[t];
        j = nbr[t];
        for (uint64_t e = t + 1; e < indptr[i + 1]; e++)
            if (nbr[e] == j) { dead = true; break; }
        if (j == i) dead = true;  // would collide with the self key; excluded by kgraph.rs:502
    } else {
        i = j = t - nnz;
    }
    keys[2 * t] = dead ? ~0ull : ((i << 32) | j);
    keys[2 * t + 1] = dead ? ~0ull : ((j << 32) | i);
    payload[2 * t] = (uint32_t)(2 * t);
    payload[2 * t + 1] = (uint32_t)(2 * t + 1);
}

// sym value of every source key: max(w_ij, w_ji) if the reverse key exists else w_ij (:527-534)
__global__ void sym_value_kernel(uint64_t n, uint64_t nnz, const uint64_t* __restrict__ indptr, const uint32_t* __restrict__ nbr,
                                 const uint32_t* __restrict__ src, const float* __restrict__ kself, const float* __restrict__ kval,
                                 float* __restrict__ sym) {
    uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (t >= nnz + n) return;
    if (t >= nnz) { sym[t] = kself[t - nnz]; return; }  // max(w, w)
    const uint32_t i = src[t], j = nbr[t];
    const float w = kval[t];
    float s = w;
    for (uint64_t e = indptr[j + 1]; e > indptr[j]; e--)  // last occurrence wins
        if (nbr[e - 1] == i) { s = fmaxf(w, kval[e - 1]); break; }
    sym[t] = s;
}

// one wave per row of the sorted triplets: out[r] = sum_p val(p)
template <bool FROM_SYM>
__global__ void __launch_bounds__(256) sorted_row_sum_kernel(uint64_t n, const uint64_t* __restrict__ rowptr,
                                                             const uint32_t* __restrict__ payload, const float* __restrict__ vals,
                                                             float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const uint64_t wave = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 6;
    const uint64_t nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t r = wave; r < n; r += nwaves) {
        float acc = 0.f;
        for (uint64_t p = rowptr[r] + lane; p < rowptr[r + 1]; p += 64) acc += FROM_SYM ? vals[payload[p] >> 1] : vals[p];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
        if (lane == 0) out[r] = acc;
    }
}

__global__ void scale_vec_kernel(uint64_t n, float* __restrict__ x, float divisor) {
    uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i < n) x[i] /= divisor;
}
__global__ void beta_scales_kernel(uint64_t n, const float* __restrict__ q, float beta, float mean_scale, float* __restrict__ out) {
    uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i < n) out[i] = powf(q[i], beta) * mean_scale;  // :938-942
}
__global__ void sqrt_vec_kernel(uint64_t n, const float* __restrict__ x, float* __restrict__ out) {
    uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i < n) out[i] = sqrtf(x[i]);
}

// v[p] = sym / (q_row q_col)^alfa   (:553-557)
__global__ void trip_density_weight_kernel(uint64_t cnt, const uint64_t* __restrict__ keys, const uint32_t* __restrict__ payload,
                                           const float* __restrict__ sym, const float* __restrict__ q, float alfa, float* __restrict__ v) {
    uint64_t p = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (p >= cnt) return;
    const uint64_t k = keys[p];
    v[p] = sym[payload[p] >> 1] / powf(q[k >> 32] * q[k & 0xFFFFFFFFull], alfa);
}
// v[p] /= sw_row * sw_col   (:566-570)
__global__ void trip_sym_norm_kernel(uint64_t cnt, const uint64_t* __restrict__ keys, const float* __restrict__ sw, float* __restrict__ v) {
    uint64_t p = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (p >= cnt) return;
    const uint64_t k = keys[p];
    v[p] /= sw[k >> 32] * sw[k & 0xFFFFFFFFull];
}
__global__ void head_flags_kernel(uint64_t cnt, const uint64_t* __restrict__ keys, uint32_t* __restrict__ flags) {
    uint64_t p = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (p >= cnt) return;
    flags[p] = (p == 0 || keys[p] != keys[p - 1]) ? 1u : 0u;
}
// TriMat::to_csr (:572-578): duplicates summed
__global__ void merge_duplicates_kernel(uint64_t cnt, const uint64_t* __restrict__ keys, const uint32_t* __restrict__ flags,
                                        const uint32_t* __restrict__ pos, const float* __restrict__ v, uint64_t* __restrict__ mkeys,
                                        uint32_t* __restrict__ cols, float* __restrict__ vals) {
    uint64_t p = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (p >= cnt || !flags[p]) return;
    float s = v[p];
    for (uint64_t x = p + 1; x < cnt && keys[x] == keys[p]; x++) s += v[x];
    const uint32_t o = pos[p];
    mkeys[o] = keys[p];
    cols[o] = (uint32_t)(keys[p] & 0xFFFFFFFFull);
    vals[o] = s;
}

// ---- dense branch (N <= 5000), diffmaps.rs:445-508 / :865-892 ----
__global__ void dense_scatter_kernel(uint64_t n, const uint64_t* __restrict__ indptr, const uint32_t* __restrict__ nbr,
                                     const float* __restrict__ kself, const float* __restrict__ kval, float* __restrict__ p) {
    uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    p[i * n + i] = kself[i];
    for (uint64_t e = indptr[i]; e < indptr[i + 1]; e++) p[i * n + nbr[e]] = kval[e];  // :455 later entries overwrite
}
__global__ void dense_sym_kernel(uint64_t n, const float* __restrict__ p, float* __restrict__ s) {
    uint64_t idx = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (idx >= n * n) return;
    const uint64_t i = idx / n, j = idx % n;
    s[idx] = (p[idx] + p[j * n + i]) * 0.5f;  // :460
}
__global__ void __launch_bounds__(256) dense_row_sum_kernel(uint64_t n, const float* __restrict__ s, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const uint64_t wave = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 6;
    const uint64_t nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t r = wave; r < n; r += nwaves) {
        float acc = 0.f;
        for (uint64_t c = lane; c < n; c += 64) acc += s[r * n + c];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
        if (lane == 0) out[r] = acc;
    }
}
__global__ void dense_density_weight_kernel(uint64_t n, float* __restrict__ s, const float* __restrict__ q, float alfa) {
    uint64_t idx = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (idx >= n * n) return;
    s[idx] /= powf(q[idx / n] * q[idx % n], alfa);  // :476
}
__global__ void dense_sym_norm_kernel(uint64_t n, float* __restrict__ s, const float* __restrict__ sw) {
    uint64_t idx = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (idx >= n * n) return;
    s[idx] /= sw[idx / n] * sw[idx % n];  // :486
}

// embed_from_laplacian rows, diffmaps.rs:1226-1237
__global__ void embed_rows_kernel(uint64_t n, uint32_t r, const float* __restrict__ u, const float* __restrict__ lam_pow,
                                  const float* __restrict__ normalizer, const float* __restrict__ normed, float sum_diag,
                                  uint32_t real_dim, float* __restrict__ y0) {
    uint64_t idx = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (idx >= n * real_dim) return;
    const uint64_t i = idx / real_dim;
    const uint32_t j = (uint32_t)(idx % real_dim);
    const float weight_i = normed[i] * sqrtf(normalizer[i] / sum_diag);  // :1228
    float v = lam_pow[j] * u[i * r + j + 1] / weight_i;                  // :1232
    v = v > 10.0f ? 10.0f : (v < -10.0f ? -10.0f : v);                   // clip, src/tools/clip.rs
    y0[idx] = v;
}

// set_data_box, embedder.rs:1396-1407
__global__ void box_center_kernel(uint64_t n, uint32_t dim, float* __restrict__ y, const float* __restrict__ means,
                                  unsigned int* __restrict__ maxbits) {
    uint64_t idx = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    unsigned int local = 0;
    if (idx < n * dim) {
        const float v = y[idx] - means[idx % dim];
        y[idx] = v;
        local = __float_as_uint(fabsf(v));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        unsigned int o = __shfl_xor(local, off);
        local = o > local ? o : local;
    }
    if ((threadIdx.x & 63) == 0 && local) atomicMax(maxbits, local);
}

}  // namespace

namespace ae {

void set_data_box_device(float* d_y, uint64_t n, uint64_t dim, float box_size) {
    std::vector<float> means(dim);
    if (dim <= 256) {  // :1391-1394, the dim column sums in one pass
        seq_sum_cols_f32(d_y, n, (uint32_t)dim, means.data());
        for (uint64_t j = 0; j < dim; j++) means[j] /= (float)n;
    } else {
        for (uint64_t j = 0; j < dim; j++) means[j] = seq_sum_f32(d_y + j, n, dim) / (float)n;
    }
    DevBuf<float> dm(dim);
    dm.upload(means.data(), dim);
    DevBuf<unsigned int> mb(1);
    mb.zero();
    hipLaunchKernelGGL(box_center_kernel, dim3(blocks_for(n * dim, 256)), dim3(256), 0, stream(), n, (uint32_t)dim, d_y, dm.p, mb.p);
    check_launch("box_center");
    unsigned int bits;
    mb.download(&bits, 1);
    float max_max;
    memcpy(&max_max, &bits, sizeof(float));
    max_max /= box_size / 2.0f;  // :1403
    hipLaunchKernelGGL(scale_vec_kernel, dim3(blocks_for(n * dim, 256)), dim3(256), 0, stream(), n * dim, d_y, max_max);  // :1404-1405
    check_launch("box_scale");
}

// the sorted triplet structure of one graph (shared by the density pass and the laplacian pass)
struct TripletStructure {
    uint64_t n = 0, nnz = 0, nkeys = 0, ntrip = 0, valid = 0;
    DevBuf<uint32_t> src;
    DevBuf<uint64_t> keys;     // sorted
    DevBuf<uint32_t> payload;  // sorted: 2t / 2t+1
    DevBuf<uint64_t> rowptr;   // n + 1
    void build(const ae_kgraph* g) {
        n = g->n; nnz = g->nnz; nkeys = nnz + n; ntrip = 2 * nkeys;
        if (ntrip >= 0xFFFFFFFFull) fail(AE_ERR_INVALID_ARG, "graph too large for u32 triplet payloads");
        src.alloc(nnz);
        hipLaunchKernelGGL(edge_src_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, stream(), n, g->indptr.p, src.p);
        check_launch("edge_src");
        DevBuf<uint64_t> k0(ntrip);
        DevBuf<uint32_t> p0(ntrip);
        keys.alloc(ntrip);
        payload.alloc(ntrip);
        hipLaunchKernelGGL(triplet_keys_kernel, dim3(blocks_for(nkeys, 256)), dim3(256), 0, stream(), n, nnz, g->indptr.p, g->nbr.p, src.p,
                           k0.p, p0.p);
        check_launch("triplet_keys");
        sort_pairs_u64_u32(k0.p, keys.p, p0.p, payload.p, ntrip);
        rowptr.alloc(n + 1);
        rowptr_from_sorted_keys(keys.p, ntrip, n, rowptr.p);
        AE_HIP(hipMemcpyAsync(&valid, rowptr.p + n, sizeof(uint64_t), hipMemcpyDeviceToHost, stream()));
        sync();
    }
};

static void kernel_rows(const ae_kgraph* g, const float* d_scales, float cscale, float epsil, DevBuf<float>& kself, DevBuf<float>& kval) {
    kself.alloc(g->n);
    kval.alloc(g->nnz);
    DevBuf<unsigned long long> aux(2);
    unsigned long long init[2] = {0ull, ~0ull};
    aux.upload(init, 2);
    hipLaunchKernelGGL(dmap_kernel_rows_kernel, dim3(blocks_for(g->n, 256)), dim3(256), 0, stream(), g->n, g->indptr.p, g->nbr.p, g->dist.p,
                       d_scales, cscale, epsil, kself.p, kval.p, aux.p, aux.p + 1);
    check_launch("dmap_kernel_rows");
    unsigned long long h[2];
    aux.download(h, 2);
    if (h[1] != ~0ull) fail(AE_ERR_ISOLATED_NODE, "encountered an isolated point (node %llu)", h[1] & ((1ull << 48) - 1));
}

// laplacian_from_kgraph, diffmaps.rs:397-422
void dmap_laplacian_device(const ae_kgraph* g, const ae_diffusion_params* dp, int force_repr, ae_laplacian* lap) {
    const uint64_t n = g->n;
    const uint32_t max_nbng = g->max_nbng;
    if (dp->beta > 0.f) fail(AE_ERR_BETA, "beta cannot be > 0.");  // :827-830
    const uint32_t nbng = dp->has_gnbn ? (uint32_t)std::min<uint64_t>(dp->gnbn, max_nbng) : max_nbng;  // :414-418
    const bool dense = force_repr == 0 ? (n <= kFullMatRepr) : (force_repr == 1);
    if (dense && n > 46000) fail(AE_ERR_INVALID_ARG, "dense representation asked for %llu nodes", (unsigned long long)n);
    lap->n = n;
    // ---- compute_dmap_nodeparams :784-822 ----
    DevBuf<float> local(n);
    lap->normed_scales.alloc(n);
    hipLaunchKernelGGL(dmap_local_scale_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, stream(), n, g->indptr.p, g->dist.p,
                       std::min(max_nbng, nbng), local.p);
    check_launch("dmap_local_scale");
    const float mean = seq_sum_f32(local.p, n) / (float)n;  // :801-803
    if (!(mean > 0.f)) fail(AE_ERR_INVALID_ARG, "mean local scale is not positive (all distances are 0?)");  // :805
    hipLaunchKernelGGL(dmap_fix_scales_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, stream(), n, local.p, mean, lap->normed_scales.p);
    check_launch("dmap_fix_scales");
    lap->mean_scale = mean;  // :817
    const float epsil = sqrtf(dp->epsil);  // :824
    DevBuf<float> kself, kval, sym, q(n);
    TripletStructure ts;
    DevBuf<float> pdense, sdense;
    if (!dense) {
        ts.build(g);
        sym.alloc(ts.nkeys);
    } else {
        pdense.alloc(n * n);
        sdense.alloc(n * n);
    }
    auto symmetrise_rowsum = [&]() {  // fills q with the row sums of the symmetrised kernel
        if (!dense) {
            hipLaunchKernelGGL(sym_value_kernel, dim3(blocks_for(ts.nkeys, 256)), dim3(256), 0, stream(), n, g->nnz, g->indptr.p, g->nbr.p,
                               ts.src.p, kself.p, kval.p, sym.p);
            check_launch("sym_value");
            hipLaunchKernelGGL((sorted_row_sum_kernel<true>), dim3(grid_cap(n * 64, 256)), dim3(256), 0, stream(), n, ts.rowptr.p,
                               ts.payload.p, sym.p, q.p);
            check_launch("sorted_row_sum");
        } else {
            pdense.zero();
            hipLaunchKernelGGL(dense_scatter_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, stream(), n, g->indptr.p, g->nbr.p, kself.p,
                               kval.p, pdense.p);
            check_launch("dense_scatter");
            hipLaunchKernelGGL(dense_sym_kernel, dim3(blocks_for(n * n, 256)), dim3(256), 0, stream(), n, pdense.p, sdense.p);
            check_launch("dense_sym");
            hipLaunchKernelGGL(dense_row_sum_kernel, dim3(grid_cap(n * 64, 256)), dim3(256), 0, stream(), n, sdense.p, q.p);
            check_launch("dense_row_sum");
        }
    };
    auto scale_vec = [&](float* x, float divisor) {
        hipLaunchKernelGGL(scale_vec_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, stream(), n, x, divisor);
        check_launch("scale_vec");
    };
    if (dp->beta < 0.f) {  // :837-843
        kernel_rows(g, local.p, 0.f, epsil, kself, kval);  // :838
        // kernel0_to_density :855-952
        symmetrise_rowsum();
        scale_vec(q.p, (float)max_nbng);              // :888 / :931
        scale_vec(q.p, ndarray_sum_f32(q.p, n) / (float)n);  // :889-891 / :932-933 (q.sum(): ndarray order)
        lap->q_density.alloc(n);
        AE_HIP(hipMemcpyAsync(lap->q_density.p, q.p, sizeof(float) * n, hipMemcpyDeviceToDevice, stream()));
        lap->beta_scales.alloc(n);
        hipLaunchKernelGGL(beta_scales_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, stream(), n, q.p, dp->beta, mean, lap->beta_scales.p);
        check_launch("beta_scales");
        kernel_rows(g, lap->beta_scales.p, 0.f, epsil, kself, kval);  // :841
    } else {
        kernel_rows(g, nullptr, mean, epsil, kself, kval);  // :844-848
    }
    // ---- compute_laplacian :427-587 ----
    symmetrise_rowsum();                                      // :468 / :538,:543
    scale_vec(q.p, ndarray_sum_f32(q.p, n) / (float)max_nbng);    // :469-471 / :546-548  (sic: / max_nbng; q.sum(): ndarray order)
    DevBuf<float> deg(n);
    lap->normalizer.alloc(n);
    ae_matrepr& k = lap->sym_kernel;
    k.nrows = k.ncols = n;
    k.symmetric = true;
    if (dense) {
        hipLaunchKernelGGL(dense_density_weight_kernel, dim3(blocks_for(n * n, 256)), dim3(256), 0, stream(), n, sdense.p, q.p, dp->alfa);
        check_launch("dense_density_weight");
        hipLaunchKernelGGL(dense_row_sum_kernel, dim3(grid_cap(n * 64, 256)), dim3(256), 0, stream(), n, sdense.p, deg.p);  // :478
        check_launch("dense_row_sum");
        hipLaunchKernelGGL(sqrt_vec_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, stream(), n, deg.p, lap->normalizer.p);  // :482
        check_launch("sqrt_vec");
        hipLaunchKernelGGL(dense_sym_norm_kernel, dim3(blocks_for(n * n, 256)), dim3(256), 0, stream(), n, sdense.p, lap->normalizer.p);
        check_launch("dense_sym_norm");
        k.is_csr = false;
        k.nnz = n * n;
        k.values = std::move(sdense);
    } else {
        const uint64_t cnt = ts.valid;
        DevBuf<float> v(cnt);
        hipLaunchKernelGGL(trip_density_weight_kernel, dim3(blocks_for(cnt, 256)), dim3(256), 0, stream(), cnt, ts.keys.p, ts.payload.p, sym.p,
                           q.p, dp->alfa, v.p);
        check_launch("trip_density_weight");
        hipLaunchKernelGGL((sorted_row_sum_kernel<false>), dim3(grid_cap(n * 64, 256)), dim3(256), 0, stream(), n, ts.rowptr.p, ts.payload.p,
                           v.p, deg.p);  // :561-564
        check_launch("sorted_row_sum");
        hipLaunchKernelGGL(sqrt_vec_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, stream(), n, deg.p, lap->normalizer.p);  // :565
        check_launch("sqrt_vec");
        hipLaunchKernelGGL(trip_sym_norm_kernel, dim3(blocks_for(cnt, 256)), dim3(256), 0, stream(), cnt, ts.keys.p, lap->normalizer.p, v.p);
        check_launch("trip_sym_norm");
        // TriMat::to_csr :572-578
        DevBuf<uint32_t> flags(cnt), pos(cnt);
        hipLaunchKernelGGL(head_flags_kernel, dim3(blocks_for(cnt, 256)), dim3(256), 0, stream(), cnt, ts.keys.p, flags.p);
        check_launch("head_flags");
        size_t tmp_bytes = 0;
        rocprim::exclusive_scan(nullptr, tmp_bytes, flags.p, pos.p, 0u, cnt, rocprim::plus<uint32_t>(), stream());
        DevBuf<char> tmp(tmp_bytes ? tmp_bytes : 1);
        if (rocprim::exclusive_scan(tmp.p, tmp_bytes, flags.p, pos.p, 0u, cnt, rocprim::plus<uint32_t>(), stream()) != hipSuccess)
            fail(AE_ERR_NO_DEVICE, "rocprim exclusive_scan failed");
        uint32_t last_pos = 0, last_flag = 0;
        AE_HIP(hipMemcpyAsync(&last_pos, pos.p + cnt - 1, sizeof(uint32_t), hipMemcpyDeviceToHost, stream()));
        AE_HIP(hipMemcpyAsync(&last_flag, flags.p + cnt - 1, sizeof(uint32_t), hipMemcpyDeviceToHost, stream()));
        sync();
        const uint64_t merged = (uint64_t)last_pos + last_flag;
        DevBuf<uint64_t> mkeys(merged);
        k.is_csr = true;
        k.nnz = merged;
        k.indices.alloc(merged);
        k.values.alloc(merged);
        k.indptr.alloc(n + 1);
        hipLaunchKernelGGL(merge_duplicates_kernel, dim3(blocks_for(cnt, 256)), dim3(256), 0, stream(), cnt, ts.keys.p, flags.p, pos.p, v.p,
                           mkeys.p, k.indices.p, k.values.p);
        check_launch("merge_duplicates");
        rowptr_from_sorted_keys(mkeys.p, merged, n, k.indptr.p);
        sync();
    }
    sync();
}

// GraphLaplacian::do_svd, graphlaplace.rs:127-134
void laplacian_do_svd_device(ae_laplacian* lap, std::vector<float>& s, DevBuf<float>& u) {
    ae_matrepr& k = lap->sym_kernel;
    if (!k.is_csr && lap->n <= kFullSvdSizeLimit) {
        full_svd_leading(k, 20, s, u);  // do_full_svd :82-94 (leading 20 triplets, see svd.hip)
    } else {
        direct_svd_rank(k, 20, 5, s, u);  // do_approx_svd :97-125: RANK(rank = 20, nbiter = 5)
    }
    for (float x : s)
        if (!std::isfinite(x)) fail(AE_ERR_SVD, "svd approximation failed (non finite singular value)");
}

// embed_from_laplacian, diffmaps.rs:1145-1243.  d_y0: n x real_dim
uint32_t embed_from_laplacian_device(ae_laplacian* lap, uint64_t asked_dim, float t, bool has_t, DevBuf<float>& y0,
                                     std::vector<float>* s_out) {
    std::vector<float> s;
    DevBuf<float> u;
    laplacian_do_svd_device(lap, s, u);  // :1164
    const uint32_t r = (uint32_t)s.size();
    if (r < 2) fail(AE_ERR_SVD, "svd returned less than 2 singular values");
    if (r > 2 && s[1] > s[0]) fail(AE_ERR_SPECTRUM, "svd spectrum not decreasing");  // :1176
    const uint32_t real_dim = (uint32_t)std::min<uint64_t>(asked_dim, r - 1);        // :1207
    std::vector<float> nl(r);
    for (uint32_t j = 0; j < r; j++) nl[j] = s[j] / s[0];  // :1213
    if (!has_t && r < 3) fail(AE_ERR_SVD, "the automatic diffusion time needs 3 singular values (the reference indexes out of bounds, diffmaps.rs:1216)");
    const float time = has_t ? t : std::fmin(5.0f, std::log(0.9f) / std::log(nl[2] / nl[1]));  // :1214-1217
    std::vector<float> lam_pow(real_dim);
    for (uint32_t j = 0; j < real_dim; j++) lam_pow[j] = std::pow(nl[j + 1], time);  // :1232
    DevBuf<float> dl(real_dim);
    dl.upload(lam_pow.data(), real_dim);
    const float sum_diag = seq_sum_f32(lap->normalizer.p, lap->n) / (float)lap->n;  // :1223
    y0.alloc(lap->n * real_dim);
    hipLaunchKernelGGL(embed_rows_kernel, dim3(blocks_for(lap->n * real_dim, 256)), dim3(256), 0, stream(), lap->n, r, u.p, dl.p,
                       lap->normalizer.p, lap->normed_scales.p, sum_diag, real_dim, y0.p);
    check_launch("embed_rows");
    sync();
    if (s_out) *s_out = s;
    return real_dim;
}

}  // namespace ae

extern "C" {

int32_t ae_dmap_laplacian_from_kgraph(const ae_kgraph* g, const ae_diffusion_params* dp, int32_t force_repr, ae_laplacian** out) {
    return guard([&] {
        require_device();
        if (!g || !dp || !out) fail(AE_ERR_INVALID_ARG, "null argument");
        std::unique_ptr<ae_laplacian> lap(new ae_laplacian);
        dmap_laplacian_device(g, dp, force_repr, lap.get());
        *out = lap.release();
    });
}
int32_t ae_laplacian_destroy(ae_laplacian* l) {
    return guard([&] { delete l; });
}
int32_t ae_laplacian_info(const ae_laplacian* l, int32_t* is_csr, uint64_t* n, uint64_t* nnz) {
    return guard([&] {
        if (!l) fail(AE_ERR_INVALID_ARG, "null argument");
        if (is_csr) *is_csr = l->sym_kernel.is_csr ? 1 : 0;
        if (n) *n = l->n;
        if (nnz) *nnz = l->sym_kernel.nnz;
    });
}
int32_t ae_laplacian_get_kernel(const ae_laplacian* l, uint64_t* indptr, uint32_t* indices, float* values) {
    return guard([&] {
        if (!l) fail(AE_ERR_INVALID_ARG, "null argument");
        const ae_matrepr& k = l->sym_kernel;
        if (k.is_csr) {
            if (indptr) k.indptr.download(indptr, l->n + 1);
            if (indices) k.indices.download(indices, k.nnz);
        }
        if (values) k.values.download(values, k.nnz);
    });
}
int32_t ae_laplacian_get_vectors(const ae_laplacian* l, float* normalizer, float* normed_scales, float* q_density, float* beta_scales,
                                 float* mean_scale) {
    return guard([&] {
        if (!l) fail(AE_ERR_INVALID_ARG, "null argument");
        if (normalizer) l->normalizer.download(normalizer, l->n);
        if (normed_scales) l->normed_scales.download(normed_scales, l->n);
        if (q_density && l->q_density.n) l->q_density.download(q_density, l->n);
        if (beta_scales && l->beta_scales.n) l->beta_scales.download(beta_scales, l->n);
        if (mean_scale) *mean_scale = l->mean_scale;
    });
}
int32_t ae_laplacian_do_svd(ae_laplacian* l, float* s, float* u, uint64_t* rank_out) {
    return guard([&] {
        require_device();
        if (!l || !s) fail(AE_ERR_INVALID_ARG, "null argument");
        std::vector<float> hs;
        DevBuf<float> du;
        laplacian_do_svd_device(l, hs, du);
        memcpy(s, hs.data(), sizeof(float) * hs.size());
        if (u) du.download(u, l->n * hs.size());
        if (rank_out) *rank_out = hs.size();
    });
}
int32_t ae_dmap_embed_from_kgraph(const ae_kgraph* g, const ae_diffusion_params* dp, float* y0, uint64_t* real_dim) {
    return guard([&] {
        require_device();
        if (!g || !dp || !y0) fail(AE_ERR_INVALID_ARG, "null argument");
        ae_laplacian lap;
        dmap_laplacian_device(g, dp, 0, &lap);
        DevBuf<float> dy;
        const uint32_t rd = embed_from_laplacian_device(&lap, dp->asked_dim, dp->t, dp->has_t != 0, dy, nullptr);
        dy.download(y0, g->n * rd);
        if (real_dim) *real_dim = rd;
    });
}
int32_t ae_set_data_box(float* y, uint64_t n, uint64_t dim, float box_size) {
    return guard([&] {
        require_device();
        if (!y || n == 0 || dim == 0) fail(AE_ERR_INVALID_ARG, "bad argument");
        DevBuf<float> dy(n * dim);
        dy.upload(y, n * dim);
        set_data_box_device(dy.p, n, dim, box_size);
        dy.download(y, n * dim);
    });
}

}  // extern "C"
