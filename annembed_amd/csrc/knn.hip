// knn.hip -- exact L2 kNN-graph producer on the matrix cores (the step BEFORE the path: stands in for
// hnsw_rs when a benchmark has to build its own graph; the reference copies hnsw_rs's distances, kgraph.rs:504).
//
// Definition of the result (the same as bruteforce_knn_kernel in core.hip, which stays as the fallback and the A/B
// reference): F(i, j) = f32 sum over the coordinates, in order, of (x_i[t] - x_j[t])^2 (no fma); row i = the nbng
// points j != i with the smallest (F, j), ascending; stored distance sqrtf(F).
//
// How it gets there at MFMA speed without giving up exactness:
//   1. candidate pass (GEMM-shaped, v_mfma_f32_32x32x2_f32): a workgroup owns 128 query rows and sweeps all points in
//      tiles of 128; A(i, j) = |p_j|^2 - 2 <x_i, p_j> ranks the points of a row like the squared distance does, up to
//      rounding; every row keeps the M = 32 smallest (A, j) in LDS (a threshold test per tile entry, insertion only
//      for the few that pass);
//   2. refine: F for the 32 candidates (thread per pair, the sequential f32 sum of the definition), ranked by (F, j);
//   3. certificate: every point outside the list has A >= a_M (largest A kept).  With E_i a bound on |A - (D - |x_i|^2)|
//      (D the real squared distance) and delta the relative error of F, a point outside the list has
//      F >= (a_M - E_i + |x_i|^2) (1 - delta); if that exceeds the nbng-th smallest F of the list, the list provably
//      contains the answer.  Rows that fail the test (near-ties beyond nbng + 8 points, duplicates) are recomputed by
//      the brute-force kernel -- the result is exact either way, only the speed is data dependent.
#include "internal.h"
#include "linalg.h"

#include <cfloat>
#include <chrono>

using namespace ae;

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int kBQ = 128;  // query rows per workgroup
constexpr int kBP = 128;  // points per tile
constexpr int kKT = 32;   // coordinates per LDS stage
constexpr int kHalf = kBP / 2;  // columns of the ranking-value buffer (fits in the staging buffers: 2 workgroups per CU)
static_assert(kBQ * (kHalf + 1) <= (kBQ + kBP) * (kKT + 1), "ranking buffer must fit in the staging buffers");

// |p_j|^2 in f32 (any fixed order: it only enters the ranking value A, bounded by E) and the largest of them
__global__ void __launch_bounds__(256) knn_norms_kernel(const float* __restrict__ x, uint64_t n, uint64_t dim, float* __restrict__ pn,
                                                        double* __restrict__ pn64, unsigned int* __restrict__ pn_max_bits) {
    const uint64_t j = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (j >= n) return;
    float s = 0.f;
    double s64 = 0.;
    for (uint64_t t = 0; t < dim; t++) {
        const float v = x[j * dim + t];
        s += v * v;
        s64 += (double)v * (double)v;
    }
    pn[j] = s;
    pn64[j] = s64;  // |x_j|^2 for the certificate of row j
    atomicMax(pn_max_bits, __float_as_uint(s));  // non-negative floats order like their bit patterns
}

// candidate pass.  MFMA operand layout (svd.hip): A lane -> A[i = lane & 31][k = lane >> 5], B lane -> B[k = lane >> 5][j = lane & 31],
// C: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).  Wave w owns query rows [32 w, 32 w + 32) of
// the workgroup's 128 and all four 32-column blocks of the point tile: one A fragment feeds four MFMAs.
// Rectangular form (the grouped producer below): the QUERIES are row r of a list (qrows[r], or q_begin + r when qrows is null), r < nq;
// the POINTS the contiguous range [p_begin, p_end) of x.  The whole-set producer is q_begin = 0, nq = n, [0, n).
struct KnnRect {
    const uint32_t* qrows;
    uint64_t q_begin, nq, p_begin, p_end;
};
__device__ __forceinline__ uint64_t knn_qid(const KnnRect& rc, uint64_t r) { return rc.qrows ? (uint64_t)rc.qrows[r] : rc.q_begin + r; }
template <bool VEC4, int kM>  // dim % 4 == 0: 16-byte loads; kM candidates kept per row (32: two workgroups per CU, 64: one)
__global__ void __launch_bounds__(256, kM == 32 ? 2 : 1) knn_candidates_kernel(const float* __restrict__ x, KnnRect rc, uint64_t dim,
                                                             const float* __restrict__ pn, uint32_t* __restrict__ cand_i,
                                                             float* __restrict__ cand_a) {
    constexpr int LDK = kKT + 1;
    extern __shared__ float smem[];
    float* sA = smem;                       // [kBQ][LDK]
    float* sB = sA + kBQ * LDK;             // [kBP][LDK]
    float* tile = smem;                     // [kBQ][kHalf + 1]: half a tile of ranking values, reuses the staging buffers
    float* ld = sB + kBP * LDK;             // [kM][kBQ]  candidate values, ascending in the slot index
    uint32_t* li = reinterpret_cast<uint32_t*>(ld + kM * kBQ);  // [kM][kBQ]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const uint64_t q0 = blockIdx.x * (uint64_t)kBQ;
    const uint64_t n = rc.p_end;   // (end of the point range: the sweep's bound)
    // the four query rows this thread stages (fixed over the sweep) and, for the threads that scan a row, that row's own id
    uint64_t qa_id[4];
#pragma unroll
    for (int q = 0; q < 4; q++) { const uint64_t r = q0 + ((tid + q * 256) >> 3); qa_id[q] = r < rc.nq ? knn_qid(rc, r) : knn_qid(rc, 0); }
    const uint64_t my_q = (tid < kBQ && q0 + tid < rc.nq) ? knn_qid(rc, q0 + tid) : ~0ull;
    for (int idx = tid; idx < kM * kBQ; idx += 256) { ld[idx] = INFINITY; li[idx] = 0xFFFFFFFFu; }
    float thr = INFINITY;  // threads < kBQ: value of the last slot of their row
    // staging: (128 + 128) rows x 32 coordinates = 2048 float4, 8 per thread (4 of the query tile, 4 of the point tile)
    // The loads keep their raw values in registers; the masking (coordinates beyond dim -> 0) happens when the stage is
    // written to LDS, one K-step later -- masking at the load site would put the wait for the loads in front of the MFMAs.
    // Rows beyond n are clamped to row 0: their results are never used (query side) or overwritten by +inf (point side).
    float ra[4][4], rb[4][4];
    auto gload = [&](uint64_t p0, uint64_t k0) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int idx = tid + q * 256;
            const int r = idx >> 3, kc = (idx & 7) * 4;
            const uint64_t kk = k0 + kc;
            const uint64_t qb = p0 + r;
            const float* pa = x + qa_id[q] * dim;
            const float* pb = x + (qb < n ? qb : rc.p_begin) * dim;
            if constexpr (VEC4) {
                const uint64_t ko = kk < dim ? kk : 0;
                const float4 ta = *reinterpret_cast<const float4*>(pa + ko);
                const float4 tb = *reinterpret_cast<const float4*>(pb + ko);
                ra[q][0] = ta.x; ra[q][1] = ta.y; ra[q][2] = ta.z; ra[q][3] = ta.w;
                rb[q][0] = tb.x; rb[q][1] = tb.y; rb[q][2] = tb.z; rb[q][3] = tb.w;
            } else {
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    const uint64_t ko = kk + t < dim ? kk + t : 0;
                    ra[q][t] = pa[ko];
                    rb[q][t] = pb[ko];
                }
            }
        }
    };
    gload(rc.p_begin, 0);
    for (uint64_t p0 = rc.p_begin; p0 < n; p0 += kBP) {
        f32x16 acc[4];
#pragma unroll
        for (int b = 0; b < 4; b++)
#pragma unroll
            for (int q = 0; q < 16; q++) acc[b][q] = 0.f;
        for (uint64_t k0 = 0; k0 < dim; k0 += kKT) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int idx = tid + q * 256;
                const int off = (idx >> 3) * LDK + (idx & 7) * 4;
                const uint64_t kk = k0 + (idx & 7) * 4;
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    const bool kin = kk + t < dim;
                    sA[off + t] = kin ? ra[q][t] : 0.f;
                    sB[off + t] = kin ? rb[q][t] : 0.f;
                }
            }
            __syncthreads();
            // next stage (or the first stage of the next tile) in flight under the MFMAs and the epilogue
            if (k0 + kKT < dim) gload(p0, k0 + kKT);
            else if (p0 + kBP < n) gload(p0 + kBP, 0);
            __builtin_amdgcn_sched_barrier(0);  // the loads are issued here, not sunk below the MFMAs to their first use
            const float* pa = sA + (w * 32 + (lane & 31)) * LDK + (lane >> 5);
            const float* pb = sB + (lane & 31) * LDK + (lane >> 5);
#pragma unroll
            for (int kk = 0; kk < kKT; kk += 2) {
                const float av = pa[kk];
#pragma unroll
                for (int b = 0; b < 4; b++) acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, pb[b * 32 * LDK + kk], acc[b], 0, 0, 0);
            }
            __syncthreads();
        }
        // ranking values, half a tile at a time, into the (now free) staging buffers; every row is then scanned by its
        // own thread: insertion is private to the row, no locks
#pragma unroll
        for (int h = 0; h < 2; h++) {
#pragma unroll
            for (int bb = 0; bb < 2; bb++) {
                const int b = 2 * h + bb;
                const int col = bb * 32 + (lane & 31);
                const uint64_t j = p0 + h * kHalf + col;
                const float nj = j < n ? pn[j] : INFINITY;
#pragma unroll
                for (int q = 0; q < 16; q++) {
                    const int row = w * 32 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
                    tile[row * (kHalf + 1) + col] = j < n ? nj - 2.f * acc[b][q] : INFINITY;
                }
            }
            __syncthreads();
            if (my_q != ~0ull) {
                const uint64_t qi = my_q;
                const float* trow = tile + tid * (kHalf + 1);
                for (int c = 0; c < kHalf; c++) {
                    const float v = trow[c];
                    const uint64_t j = p0 + h * kHalf + c;
                    if (!(v < thr) || j == qi) continue;
                    int pos = kM - 1;
                    while (pos > 0 && v < ld[(pos - 1) * kBQ + tid]) {
                        ld[pos * kBQ + tid] = ld[(pos - 1) * kBQ + tid];
                        li[pos * kBQ + tid] = li[(pos - 1) * kBQ + tid];
                        pos--;
                    }
                    ld[pos * kBQ + tid] = v;
                    li[pos * kBQ + tid] = (uint32_t)j;
                    thr = ld[(kM - 1) * kBQ + tid];
                }
            }
            __syncthreads();  // the buffer is rewritten by the next half / the next tile's staging
        }
    }
    if (my_q != ~0ull) {
        const uint64_t qr = q0 + tid;   // (candidate lists are indexed by the query's place in the list)
        for (int s = 0; s < kM; s++) {
            cand_i[qr * kM + s] = li[s * kBQ + tid];
            cand_a[qr * kM + s] = ld[s * kBQ + tid];
        }
    }
}

// refine + certificate: kM lanes per row, lane s owns candidate s
// out_by_list: the results go to row r of nbr / dist (the query's place in the list) instead of row i (its id); raw: the squared
// distance F is stored instead of sqrtf(F) (the grouped producer merges lists by (F, j) and takes the roots at the end).
// The flagged entries are places in the list.
template <int kM>
__global__ void __launch_bounds__(256) knn_refine_kernel(const float* __restrict__ x, KnnRect rc, uint64_t dim, uint32_t k,
                                                         const uint32_t* __restrict__ cand_i, const float* __restrict__ cand_a,
                                                         const double* __restrict__ pn64,
                                                         const unsigned int* __restrict__ pn_max_bits, uint32_t* __restrict__ nbr,
                                                         float* __restrict__ dist, uint32_t* __restrict__ flagged,
                                                         unsigned int* __restrict__ nflagged, int out_by_list, int raw,
                                                         const uint32_t* __restrict__ orig) {
    const uint64_t gid = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    const uint64_t r = gid / kM;
    const int s = (int)(gid % kM);
    if (r >= rc.nq) return;  // the kM lanes of a row leave together
    const uint64_t i = knn_qid(rc, r);
    const uint64_t orow = out_by_list ? r : i;
    const uint32_t j = cand_i[r * kM + s];
    const bool valid = j != 0xFFFFFFFFu;
    const float* xi = x + i * dim;
    const float* xj = x + (uint64_t)(valid ? j : 0) * dim;
    float f = 0.f;
    for (uint64_t t = 0; t < dim; t++) {
        const float df = xi[t] - xj[t];
        f += df * df;
    }
    if (!valid) f = INFINITY;
    // rank of (f, j) among the 32 candidates of the row (j: the caller's id where the points were reordered internally -- `orig`)
    const uint32_t jt = (orig && valid) ? orig[j] : j;
    int rank = 0;
    for (int o = 0; o < kM; o++) {
        const float fo = __shfl(f, o, kM);
        const uint32_t jo = (uint32_t)__shfl((int)jt, o, kM);
        rank += (fo < f || (fo == f && jo < jt)) ? 1 : 0;
    }
    if ((uint32_t)rank < k) {
        nbr[orow * k + rank] = j;
        dist[orow * k + rank] = raw ? f : sqrtf(f);
    }
    if ((uint32_t)rank == k - 1) {
        // certificate (see the header).  E bounds the rounding of A = |p|^2 - 2 <x, p> in f32 against its real value:
        // (dim + 8) u (|x|^2 + 2 max|p|^2) covers the dim-term dot product and norm sums and the final operations;
        // delta = (dim + 4) u bounds the relative error of the f32 sum of squares F.
        const double u = 1.0 / 16777216.0;
        const double xn = pn64[i];
        const double pmax = (double)__uint_as_float(*pn_max_bits) * (1. + (double)dim * u);
        const double e = ((double)dim + 8.) * u * (xn + 2. * pmax);
        const double a_last = (double)cand_a[r * kM + kM - 1];  // +inf when fewer than M points exist
        const double lower = (a_last - e + xn) * (1. - ((double)dim + 4.) * u);
        if (!(lower > (double)f)) flagged[atomicAdd(nflagged, 1u)] = (uint32_t)r;
    }
}

// the norms every rectangle of one point set shares
struct KnnNorms {
    DevBuf<float> pn;
    DevBuf<double> pn64;
    DevBuf<unsigned int> pn_max;   // [0]: bits of the largest |p|^2
    void compute(const float* d_x, uint64_t n, uint64_t dim) {
        pn.alloc_pooled(n);
        pn64.alloc_pooled(n);
        pn_max.alloc(1);
        pn_max.zero();
        hipLaunchKernelGGL(knn_norms_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, stream(), d_x, n, dim, pn.p, pn64.p, pn_max.p);
        check_launch("knn_norms");
    }
};

// exact k nearest points of [p_begin, p_end) for every query of the rectangle; returns the rows the brute-force fallback recomputed
template <int kM>
static uint64_t knn_rect_m(const float* d_x, const KnnRect& rc, uint64_t dim, uint32_t k, const KnnNorms& nm, uint32_t* d_nbr, float* d_dist, bool out_by_list,
                           bool raw, const uint32_t* d_orig) {
    if (!rc.nq || rc.p_end <= rc.p_begin) return 0;
    DevBuf<float> cand_a;
    DevBuf<uint32_t> cand_i, flagged, flagged_ids;
    DevBuf<unsigned int> counter;
    cand_a.alloc_pooled(rc.nq * kM);
    cand_i.alloc_pooled(rc.nq * kM);
    flagged.alloc_pooled(rc.nq);
    counter.alloc_pooled(1);
    counter.zero();
    const size_t lds = sizeof(float) * ((size_t)(kBQ + kBP) * (kKT + 1) + 2 * (size_t)kM * kBQ);
    static bool attr_set = false;
    if (!attr_set) {
        AE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(knn_candidates_kernel<true, kM>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        AE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(knn_candidates_kernel<false, kM>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    if (dim % 4 == 0)
        hipLaunchKernelGGL((knn_candidates_kernel<true, kM>), dim3(blocks_for(rc.nq, kBQ)), dim3(256), lds, stream(), d_x, rc, dim, (const float*)nm.pn.p,
                           cand_i.p, cand_a.p);
    else
        hipLaunchKernelGGL((knn_candidates_kernel<false, kM>), dim3(blocks_for(rc.nq, kBQ)), dim3(256), lds, stream(), d_x, rc, dim, (const float*)nm.pn.p,
                           cand_i.p, cand_a.p);
    check_launch("knn_candidates");
    hipLaunchKernelGGL(knn_refine_kernel<kM>, dim3(blocks_for(rc.nq * kM, 256)), dim3(256), 0, stream(), d_x, rc, dim, k, (const uint32_t*)cand_i.p,
                       (const float*)cand_a.p, (const double*)nm.pn64.p, (const unsigned int*)nm.pn_max.p, d_nbr, d_dist, flagged.p, counter.p, out_by_list ? 1 : 0,
                       raw ? 1 : 0, d_orig);
    check_launch("knn_refine");
    unsigned int h = 0;
    counter.download(&h, 1);
    if (h) bruteforce_knn_rect(d_x, dim, k, rc.qrows, rc.q_begin, flagged.p, h, rc.p_begin, rc.p_end, d_nbr, d_dist, out_by_list, raw, d_orig);
    return h;
}
static uint64_t knn_rect(const float* d_x, const KnnRect& rc, uint64_t dim, uint32_t k, const KnnNorms& nm, uint32_t* d_nbr, float* d_dist, bool out_by_list, bool raw,
                         const uint32_t* d_orig = nullptr) {
    if (k + 8 <= 32) return knn_rect_m<32>(d_x, rc, dim, k, nm, d_nbr, d_dist, out_by_list, raw, d_orig);
    if (k + 8 <= 64) return knn_rect_m<64>(d_x, rc, dim, k, nm, d_nbr, d_dist, out_by_list, raw, d_orig);
    fail(AE_ERR_INVALID_ARG, "knn_mfma: nbng + 8 exceeds the candidate list");
    return 0;
}

// ---------------------------------------------------------------------------------------------
// grouped producer: exact GLOBAL kNN of points that come sorted into groups (clusters)
// ---------------------------------------------------------------------------------------------
// Phase A: the k nearest points inside the own group (a rectangle per group).  Phase B, group by group: with m_g any fixed point (the
// group's centroid) and rho_y = |y - m_g|, |x - y| >= |x - m_g| - rho_y for every y of g (triangle inequality): a point x of another
// group needs only the SHELL of g beyond radius |x - m_g| - (its current k-th distance).  The points of a group are therefore kept
// sorted by rho, largest first -- a shell is a prefix of the group --, every x finds its prefix length by a binary search, the queries
// of g are bucketed by prefix length (2 048, 8 192, ... points) and every bucket runs as one rectangle against its prefix; the lists
// are merged by (F, j).  Exact by construction: a skipped pair is provably no neighbour.  (In 28-D the clusters of configs[3] overlap
// as balls -- the test with the group's largest radius alone lets 21 % of the pairs through -- but the outer shells are thin.)
// The points are reordered internally (inside their groups); `orig` carries the caller's ids for the tie-breaks of the definition.
__global__ void __launch_bounds__(256) group_sum_kernel(const float* __restrict__ x, uint64_t b, uint64_t e, uint64_t dim, double* __restrict__ out) {
    // column sums in f64 -> atomicAdd (the centroid may be ANY point: its rounding does not matter)
    extern __shared__ double s_sum[];
    for (uint64_t t = threadIdx.x; t < dim; t += 256) s_sum[t] = 0.;
    __syncthreads();
    for (uint64_t idx = b * dim + blockIdx.x * 256ull + threadIdx.x; idx < e * dim; idx += (uint64_t)gridDim.x * 256ull)
        atomicAdd(&s_sum[idx % dim], (double)x[idx]);
    __syncthreads();
    for (uint64_t t = threadIdx.x; t < dim; t += 256) atomicAdd(out + t, s_sum[t]);
}
__device__ __forceinline__ float dist_to(const float* __restrict__ xi, const float* __restrict__ m, uint64_t dim) {
    float s = 0.f;
    for (uint64_t t = 0; t < dim; t++) { const float df = xi[t] - m[t]; s += df * df; }
    return sqrtf(s);
}
// rho_i = |x_i - m| for i in [b, e), the sort key that puts the largest first, the identity
__global__ void __launch_bounds__(256) group_radius_kernel(const float* __restrict__ x, uint64_t b, uint64_t e, uint64_t dim, const float* __restrict__ m,
                                                           uint32_t* __restrict__ key, uint32_t* __restrict__ ident) {
    const uint64_t i = b + blockIdx.x * 256ull + threadIdx.x;
    if (i >= e) return;
    key[i] = 0x7FFFFFFFu - __float_as_uint(dist_to(x + i * dim, m, dim));   // (non-negative floats order like their bit patterns)
    ident[i] = (uint32_t)i;
}
__global__ void __launch_bounds__(256) group_gather_kernel(const float* __restrict__ x, uint64_t n, uint64_t dim, const uint32_t* __restrict__ order,
                                                           const uint32_t* __restrict__ key, float* __restrict__ xp, float* __restrict__ rho) {
    for (uint64_t t = blockIdx.x * 256ull + threadIdx.x; t < n * dim; t += (uint64_t)gridDim.x * 256ull) {
        const uint64_t p = t / dim, q = t % dim;
        xp[t] = x[(uint64_t)order[p] * dim + q];
        if (q == 0) rho[p] = __uint_as_float(0x7FFFFFFFu - key[p]);
    }
}
constexpr int kBuckets = 5;
__host__ __device__ inline uint64_t bucket_len(int b, uint64_t group) {
    const uint64_t l = 2048ull << (2 * b);
    return (b == kBuckets - 1 || l > group) ? group : l;
}
// the queries of group [gb, ge): every point outside it, by the length of the prefix (shell) of the group it cannot exclude
__global__ void __launch_bounds__(256) group_filter_kernel(const float* __restrict__ x, uint64_t n, uint64_t dim, uint32_t k, uint64_t gb, uint64_t ge,
                                                           const float* __restrict__ m, const float* __restrict__ rho, const float* __restrict__ d2,
                                                           uint32_t* __restrict__ qlist, unsigned int* __restrict__ nq) {
    const uint64_t i = blockIdx.x * 256ull + threadIdx.x;
    if (i >= n || (i >= gb && i < ge)) return;
    const float dm = dist_to(x + i * dim, m, dim);
    const float rk = sqrtf(d2[i * k + k - 1] * (1.f + 2e-4f));   // current k-th distance (F is raw: squared), rounded up
    const float tau = dm - rk - 1e-4f * (dm + rho[gb]) - 1e-30f;  // points of the group with rho < tau are provably farther than the k-th
    if (!(rho[gb] >= tau)) return;                                // (rho[gb]: the group's largest)
    uint64_t lo = 0, hi = ge - gb;   // first place whose rho < tau (rho descends)
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if (rho[gb + mid] >= tau) lo = mid + 1; else hi = mid;
    }
    int b = 0;
    while (b < kBuckets - 1 && bucket_len(b, ge - gb) < lo) b++;
    qlist[(uint64_t)b * n + atomicAdd(nq + b, 1u)] = (uint32_t)i;
}
// row i = qlist[r] keeps the k smallest (F, j) of its own k and the k of tmp row r (j compared by the caller's ids)
__global__ void __launch_bounds__(256) group_merge_kernel(const uint32_t* __restrict__ qlist, uint64_t nq, uint32_t k, const uint32_t* __restrict__ t_nbr,
                                                          const float* __restrict__ t_d2, const uint32_t* __restrict__ orig, uint32_t* __restrict__ nbr,
                                                          float* __restrict__ d2) {
    const uint64_t r = blockIdx.x * 256ull + threadIdx.x;
    if (r >= nq) return;
    const uint64_t i = qlist[r];
    uint32_t* bi = nbr + i * k;
    float* bd = d2 + i * k;
    for (uint32_t s = 0; s < k; s++) {
        const float d = t_d2[r * k + s];
        const uint32_t j = t_nbr[r * k + s];
        if (j == 0xFFFFFFFFu) break;
        const uint32_t oj = orig[j];
        if (!(d < bd[k - 1] || (d == bd[k - 1] && oj < orig[bi[k - 1]]))) break;   // (tmp rows ascend: nothing later fits either)
        uint32_t pos = k - 1;
        while (pos > 0 && (d < bd[pos - 1] || (d == bd[pos - 1] && oj < orig[bi[pos - 1]]))) { bd[pos] = bd[pos - 1]; bi[pos] = bi[pos - 1]; pos--; }
        bd[pos] = d;
        bi[pos] = j;
    }
}
// back to the caller's numbering, roots taken: row orig[p] = {orig[j], sqrt(F)} of internal row p
__global__ void __launch_bounds__(256) group_finish_kernel(uint64_t n, uint32_t k, const uint32_t* __restrict__ orig, const uint32_t* __restrict__ nbr_p,
                                                           const float* __restrict__ d2_p, uint32_t* __restrict__ nbr, float* __restrict__ dist) {
    for (uint64_t t = blockIdx.x * 256ull + threadIdx.x; t < n * k; t += (uint64_t)gridDim.x * 256ull) {
        const uint64_t p = t / k, s = t % k;
        const uint64_t o = (uint64_t)orig[p] * k + s;
        nbr[o] = orig[nbr_p[t]];
        dist[o] = sqrtf(d2_p[t]);
    }
}

}  // namespace

namespace ae {
void sort_pairs_u32_u32(uint32_t* d_keys_in, uint32_t* d_keys_out, uint32_t* d_vals_in, uint32_t* d_vals_out, uint64_t count, unsigned end_bit);

// exact kNN rows of all n points into d_nbr / d_dist (n x k); returns the number of rows that needed the fallback
uint64_t knn_mfma(const float* d_x, uint64_t n, uint64_t dim, uint32_t k, uint32_t* d_nbr, float* d_dist) {
    KnnNorms nm;
    nm.compute(d_x, n, dim);
    return knn_rect(d_x, KnnRect{nullptr, 0, n, 0, n}, dim, k, nm, d_nbr, d_dist, false, false);
}

// exact GLOBAL kNN rows of points sorted into groups [bounds[g], bounds[g + 1]) (see above); stats: [0] rows recomputed by the
// brute-force fallback, [1] query-point pairs of phase B, [2] of phase A
void knn_grouped(const float* d_x, uint64_t n, uint64_t dim, uint32_t k, const uint64_t* bounds, uint32_t groups, uint32_t* d_nbr, float* d_dist, uint64_t* stats) {
    if (groups == 0 || bounds[0] != 0 || bounds[groups] != n) fail(AE_ERR_INVALID_ARG, "knn_grouped: the groups must tile the points");
    for (uint32_t g = 0; g < groups; g++)
        if (bounds[g + 1] <= bounds[g] + k) fail(AE_ERR_INVALID_ARG, "knn_grouped: group %u has no more than nbng points", g);
    // centroids; inside every group the points by decreasing distance to it
    DevBuf<double> d_sum;
    DevBuf<float> d_m, xp, rho, d2;
    DevBuf<uint32_t> key, key2, ident, orig, nbr_p;
    d_sum.alloc_pooled((size_t)groups * dim);
    d_sum.zero();
    d_m.alloc_pooled((size_t)groups * dim);
    for (uint32_t g = 0; g < groups; g++)
        hipLaunchKernelGGL(group_sum_kernel, dim3(grid_cap((bounds[g + 1] - bounds[g]) * dim, 256, 256)), dim3(256), sizeof(double) * dim, stream(), d_x, bounds[g],
                           bounds[g + 1], dim, d_sum.p + (size_t)g * dim);
    std::vector<double> hs = d_sum.to_host();
    std::vector<float> hm((size_t)groups * dim);
    for (uint32_t g = 0; g < groups; g++)
        for (uint64_t t = 0; t < dim; t++) hm[(size_t)g * dim + t] = (float)(hs[(size_t)g * dim + t] / (double)(bounds[g + 1] - bounds[g]));
    d_m.upload(hm.data(), hm.size());
    key.alloc_pooled(n); key2.alloc_pooled(n); ident.alloc_pooled(n); orig.alloc_pooled(n);
    for (uint32_t g = 0; g < groups; g++) {
        const uint64_t b = bounds[g], e = bounds[g + 1];
        hipLaunchKernelGGL(group_radius_kernel, dim3(blocks_for(e - b, 256)), dim3(256), 0, stream(), d_x, b, e, dim, (const float*)(d_m.p + (size_t)g * dim), key.p, ident.p);
        sort_pairs_u32_u32(key.p + b, key2.p + b, ident.p + b, orig.p + b, e - b, 31);
    }
    xp.alloc_pooled(n * dim);
    rho.alloc_pooled(n);
    hipLaunchKernelGGL(group_gather_kernel, dim3(grid_cap(n * dim, 256, 1u << 16)), dim3(256), 0, stream(), d_x, n, dim, (const uint32_t*)orig.p, (const uint32_t*)key2.p,
                       xp.p, rho.p);
    check_launch("knn_grouped: order");
    key.release(); ident.release();
    const bool prof = debug_knob("AE_CE_PROF") != nullptr;
    auto wall = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    if (prof) sync();
    const double t_start = wall();
    KnnNorms nm;
    nm.compute(xp.p, n, dim);
    nbr_p.alloc_pooled(n * k);
    d2.alloc_pooled(n * k);
    uint64_t fell = 0, pairs_a = 0, pairs_b = 0;
    for (uint32_t g = 0; g < groups; g++) {   // phase A (raw F)
        const uint64_t b = bounds[g], e = bounds[g + 1];
        fell += knn_rect(xp.p, KnnRect{nullptr, b, e - b, b, e}, dim, k, nm, nbr_p.p, d2.p, false, true, orig.p);
        pairs_a += (e - b) * (e - b);
    }
    if (prof) sync();
    const double t_a = wall();
    if (groups > 1) {
        DevBuf<unsigned int> d_nq;
        DevBuf<uint32_t> qlist, t_nbr;
        DevBuf<float> t_d2;
        d_nq.alloc_pooled(kBuckets);
        qlist.alloc_pooled((size_t)kBuckets * n);
        uint64_t t_cap = 0;
        for (uint32_t g = 0; g < groups; g++) {   // phase B
            const uint64_t gb = bounds[g], ge = bounds[g + 1];
            d_nq.zero();
            hipLaunchKernelGGL(group_filter_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, stream(), (const float*)xp.p, n, dim, k, gb, ge,
                               (const float*)(d_m.p + (size_t)g * dim), (const float*)rho.p, (const float*)d2.p, qlist.p, d_nq.p);
            unsigned int nq[kBuckets];
            d_nq.download(nq, kBuckets);
            for (int b = 0; b < kBuckets; b++) {
                if (!nq[b]) continue;
                if (t_cap < nq[b]) { t_cap = (uint64_t)nq[b] + nq[b] / 4 + 1024; t_nbr.alloc_pooled(t_cap * k); t_d2.alloc_pooled(t_cap * k); }
                const uint64_t len = bucket_len(b, ge - gb);
                const uint32_t* ql = qlist.p + (size_t)b * n;
                fell += knn_rect(xp.p, KnnRect{ql, 0, nq[b], gb, gb + len}, dim, k, nm, t_nbr.p, t_d2.p, true, true, orig.p);
                hipLaunchKernelGGL(group_merge_kernel, dim3(blocks_for(nq[b], 256)), dim3(256), 0, stream(), ql, (uint64_t)nq[b], k, (const uint32_t*)t_nbr.p,
                                   (const float*)t_d2.p, (const uint32_t*)orig.p, nbr_p.p, d2.p);
                check_launch("knn_grouped: merge");
                pairs_b += (uint64_t)nq[b] * len;
            }
        }
    }
    hipLaunchKernelGGL(group_finish_kernel, dim3(grid_cap(n * k, 256)), dim3(256), 0, stream(), n, k, (const uint32_t*)orig.p, (const uint32_t*)nbr_p.p, (const float*)d2.p,
                       d_nbr, d_dist);
    check_launch("knn_grouped: finish");
    sync();
    if (prof) fprintf(stderr, "KNN grouped: phase A %.2f s (%.3g pairs), phase B %.2f s (%.3g pairs)\n", t_a - t_start, (double)pairs_a, wall() - t_a, (double)pairs_b);
    if (stats) { stats[0] = fell; stats[1] = pairs_b; stats[2] = pairs_a; }
}

}  // namespace ae
