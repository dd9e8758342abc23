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
template <bool VEC4, int kM>  // dim % 4 == 0: 16-byte loads; kM candidates kept per row (32: two workgroups per CU, 64: one)
__global__ void __launch_bounds__(256, kM == 32 ? 2 : 1) knn_candidates_kernel(const float* __restrict__ x, uint64_t n, uint64_t dim,
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
            const uint64_t qa = q0 + r, qb = p0 + r;
            const float* pa = x + (qa < n ? qa : 0) * dim;
            const float* pb = x + (qb < n ? qb : 0) * dim;
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
    gload(0, 0);
    for (uint64_t p0 = 0; p0 < n; p0 += kBP) {
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
            if (tid < kBQ && q0 + tid < n) {
                const uint64_t qi = q0 + tid;
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
    if (tid < kBQ && q0 + tid < n) {
        const uint64_t qi = q0 + tid;
        for (int s = 0; s < kM; s++) {
            cand_i[qi * kM + s] = li[s * kBQ + tid];
            cand_a[qi * kM + s] = ld[s * kBQ + tid];
        }
    }
}

// refine + certificate: kM lanes per row, lane s owns candidate s
template <int kM>
__global__ void __launch_bounds__(256) knn_refine_kernel(const float* __restrict__ x, uint64_t n, uint64_t dim, uint32_t k,
                                                         const uint32_t* __restrict__ cand_i, const float* __restrict__ cand_a,
                                                         const double* __restrict__ pn64,
                                                         const unsigned int* __restrict__ pn_max_bits, uint32_t* __restrict__ nbr,
                                                         float* __restrict__ dist, uint32_t* __restrict__ flagged,
                                                         unsigned int* __restrict__ nflagged) {
    const uint64_t gid = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    const uint64_t i = gid / kM;
    const int s = (int)(gid % kM);
    if (i >= n) return;  // the kM lanes of a row leave together
    const uint32_t j = cand_i[i * kM + s];
    const bool valid = j != 0xFFFFFFFFu;
    const float* xi = x + i * dim;
    const float* xj = x + (uint64_t)(valid ? j : 0) * dim;
    float f = 0.f;
    for (uint64_t t = 0; t < dim; t++) {
        const float df = xi[t] - xj[t];
        f += df * df;
    }
    if (!valid) f = INFINITY;
    // rank of (f, j) among the 32 candidates of the row
    int rank = 0;
    for (int o = 0; o < kM; o++) {
        const float fo = __shfl(f, o, kM);
        const uint32_t jo = (uint32_t)__shfl((int)j, o, kM);
        rank += (fo < f || (fo == f && jo < j)) ? 1 : 0;
    }
    if ((uint32_t)rank < k) {
        nbr[i * k + rank] = j;
        dist[i * k + rank] = sqrtf(f);
    }
    if ((uint32_t)rank == k - 1) {
        // certificate (see the header).  E bounds the rounding of A = |p|^2 - 2 <x, p> in f32 against its real value:
        // (dim + 8) u (|x|^2 + 2 max|p|^2) covers the dim-term dot product and norm sums and the final operations;
        // delta = (dim + 4) u bounds the relative error of the f32 sum of squares F.
        const double u = 1.0 / 16777216.0;
        const double xn = pn64[i];
        const double pmax = (double)__uint_as_float(*pn_max_bits) * (1. + (double)dim * u);
        const double e = ((double)dim + 8.) * u * (xn + 2. * pmax);
        const double a_last = (double)cand_a[i * kM + kM - 1];  // +inf when fewer than M points exist
        const double lower = (a_last - e + xn) * (1. - ((double)dim + 4.) * u);
        if (!(lower > (double)f)) flagged[atomicAdd(nflagged, 1u)] = (uint32_t)i;
    }
}

template <int kM>
static uint64_t knn_mfma_m(const float* d_x, uint64_t n, uint64_t dim, uint32_t k, uint32_t* d_nbr, float* d_dist) {
    DevBuf<float> pn, cand_a;
    DevBuf<double> pn64;
    DevBuf<uint32_t> cand_i, flagged;
    DevBuf<unsigned int> counters(2);
    pn.alloc_pooled(n);
    pn64.alloc_pooled(n);
    cand_a.alloc_pooled(n * kM);
    cand_i.alloc_pooled(n * kM);
    flagged.alloc_pooled(n);
    counters.zero();
    hipLaunchKernelGGL(knn_norms_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, stream(), d_x, n, dim, pn.p, pn64.p, counters.p);
    check_launch("knn_norms");
    const size_t lds = sizeof(float) * ((size_t)(kBQ + kBP) * (kKT + 1) + 2 * (size_t)kM * kBQ);
    static bool attr_set = false;
    if (!attr_set) {
        AE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(knn_candidates_kernel<true, kM>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        AE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(knn_candidates_kernel<false, kM>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    if (dim % 4 == 0)
        hipLaunchKernelGGL((knn_candidates_kernel<true, kM>), dim3(blocks_for(n, kBQ)), dim3(256), lds, stream(), d_x, n, dim, (const float*)pn.p,
                           cand_i.p, cand_a.p);
    else
        hipLaunchKernelGGL((knn_candidates_kernel<false, kM>), dim3(blocks_for(n, kBQ)), dim3(256), lds, stream(), d_x, n, dim, (const float*)pn.p,
                           cand_i.p, cand_a.p);
    check_launch("knn_candidates");
    hipLaunchKernelGGL(knn_refine_kernel<kM>, dim3(blocks_for(n * kM, 256)), dim3(256), 0, stream(), d_x, n, dim, k, (const uint32_t*)cand_i.p,
                       (const float*)cand_a.p, (const double*)pn64.p, (const unsigned int*)counters.p, d_nbr, d_dist, flagged.p, counters.p + 1);
    check_launch("knn_refine");
    unsigned int h[2];
    counters.download(h, 2);
    if (h[1]) bruteforce_knn_rows(d_x, n, dim, k, flagged.p, h[1], d_nbr, d_dist);
    return h[1];
}

}  // namespace

namespace ae {

// exact kNN rows of all n points into d_nbr / d_dist (n x k); returns the number of rows that needed the fallback
uint64_t knn_mfma(const float* d_x, uint64_t n, uint64_t dim, uint32_t k, uint32_t* d_nbr, float* d_dist) {
    if (k + 8 <= 32) return knn_mfma_m<32>(d_x, n, dim, k, d_nbr, d_dist);
    if (k + 8 <= 64) return knn_mfma_m<64>(d_x, n, dim, k, d_nbr, d_dist);
    fail(AE_ERR_INVALID_ARG, "knn_mfma: nbng + 8 exceeds the candidate list");
    return 0;
}

}  // namespace ae
