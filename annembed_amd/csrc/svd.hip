// svd.hip -- a6-a8: tools::svdapprox (src/tools/svdapprox.rs) and the linear-algebra kernels under it.
//
//   subspace_iteration_{csr,full}  svdapprox.rs:285-408   Halko-Tropp algorithm 4.4
//   SvdApprox::direct_svd          svdapprox.rs:721-799   algorithm 5.1
//
// Design (MI355X): every operand is a tall-skinny row-major panel (rows x l, l = 20 in the
// embedder); the passes are HBM-bound, so each kernel streams its panel once with
// coalesced row reads.  LAPACK's Householder QR (do_qr :998-1013) is replaced by Gram + small
// symmetric eigendecomposition + scaling, twice (SVQB): the Gram (l x l, f64) is the only
// reduction, the eigenproblem runs on one workgroup, and rank-deficient panels give zero columns
// instead of failing.  The l x n SVD of B (gesdd at :758) is done the same way through B B^T.

#include "linalg.h"
#include <algorithm>
#include <cfloat>
#include <cmath>
// (after <cstring>: rocprim's texture iterator calls the host memset)
#include <rocprim/rocprim.hpp>
#include "philox.h"

using namespace ae;

namespace {

__global__ void gaussian_fill_kernel(float* __restrict__ out, uint64_t count, uint64_t seed, uint32_t tag) {
    uint64_t nblk = (count + 3) / 4;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t b = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; b < nblk; b += stride) {
        uint32_t w[4];
        philox4x32_10((uint32_t)b, (uint32_t)(b >> 32), tag, 0, (uint32_t)seed, (uint32_t)(seed >> 32), w);
        float z[4];
        box_muller(w[0], w[1], z[0], z[1]);
        box_muller(w[2], w[3], z[2], z[3]);
#pragma unroll
        for (int t = 0; t < 4; t++)
            if (4 * b + t < count) out[4 * b + t] = z[t];
    }
}

// Y = A X, A CSR.  One wave per row; the 64 lanes are (64/LP) edge slots x LP panel columns, so a
// wave reads 64/LP neighbour rows of X (l contiguous floats each) per step.
template <int LP>
__global__ void __launch_bounds__(256) spmm_csr_kernel(uint64_t m, const uint64_t* __restrict__ indptr,
                                                       const uint32_t* __restrict__ ind, const float* __restrict__ val,
                                                       const float* __restrict__ x, float* __restrict__ y, uint32_t l) {
    constexpr int G = 64 / LP;
    const int lane = threadIdx.x & 63;
    const int c = lane & (LP - 1);
    const int g = lane / LP;
    const uint64_t wave = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 6;
    const uint64_t nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t row = wave; row < m; row += nwaves) {
        float acc = 0.f;
        const uint64_t e1 = indptr[row + 1];
        uint64_t e = indptr[row] + g;
        for (; e + 3 * G < e1; e += 4 * G) {  // 4 neighbour rows in flight per lane
            const float a0 = val[e], a1 = val[e + G], a2 = val[e + 2 * G], a3 = val[e + 3 * G];
            const uint64_t c0 = ind[e], c1 = ind[e + G], c2 = ind[e + 2 * G], c3 = ind[e + 3 * G];
            if (c < (int)l) {
                const float x0 = x[c0 * l + c], x1 = x[c1 * l + c], x2 = x[c2 * l + c], x3 = x[c3 * l + c];
                acc = fmaf(a0, x0, acc); acc = fmaf(a1, x1, acc); acc = fmaf(a2, x2, acc); acc = fmaf(a3, x3, acc);
            }
        }
        for (; e < e1; e += G) {
            const float a = val[e];
            const uint64_t col = ind[e];
            if (c < (int)l) acc = fmaf(a, x[col * l + c], acc);
        }
#pragma unroll
        for (int off = LP; off < 64; off <<= 1) acc += __shfl_xor(acc, off);
        if (g == 0 && c < (int)l) y[row * l + c] = acc;
    }
}

// Y = A X, A dense m x n row-major: one wave per row of A, lanes stride the contraction index.
template <int LT>
__global__ void __launch_bounds__(256) dense_mul_panel_kernel(const float* __restrict__ a, uint64_t m, uint64_t n,
                                                              const float* __restrict__ x, float* __restrict__ y, uint32_t l) {
    const int lane = threadIdx.x & 63;
    const uint64_t wave = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 6;
    const uint64_t nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t row = wave; row < m; row += nwaves) {
        float acc[LT];
#pragma unroll
        for (int c = 0; c < LT; c++) acc[c] = 0.f;
        const float* ar = a + row * n;
        for (uint64_t k = lane; k < n; k += 64) {
            const float av = ar[k];
            const float* xr = x + k * l;
#pragma unroll
            for (int c = 0; c < LT; c++)
                if (c < (int)l) acc[c] = fmaf(av, xr[c], acc[c]);
        }
#pragma unroll
        for (int c = 0; c < LT; c++) {
            float v = acc[c];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
            if (lane == 0 && c < (int)l) y[row * l + c] = v;
        }
    }
}

// partial[r][k][c] = sum_{i in chunk r} A[i,k] * X[i,c]  (A^T X), lanes own columns k of A (coalesced rows)
template <int LT>
__global__ void __launch_bounds__(64) dense_t_mul_panel_kernel(const float* __restrict__ a, uint64_t m, uint64_t n,
                                                               const float* __restrict__ x, float* __restrict__ partial,
                                                               uint32_t l, uint64_t rows_per_chunk) {
    const uint64_t k = blockIdx.x * 64ull + threadIdx.x;
    const uint64_t r = blockIdx.y;
    const uint64_t i0 = r * rows_per_chunk;
    const uint64_t i1 = i0 + rows_per_chunk < m ? i0 + rows_per_chunk : m;
    float acc[LT];
#pragma unroll
    for (int c = 0; c < LT; c++) acc[c] = 0.f;
    if (k < n) {
        for (uint64_t i = i0; i < i1; i++) {
            const float av = a[i * n + k];
            const float* xr = x + i * l;
#pragma unroll
            for (int c = 0; c < LT; c++)
                if (c < (int)l) acc[c] = fmaf(av, xr[c], acc[c]);
        }
        float* o = partial + (r * n + k) * l;
#pragma unroll
        for (int c = 0; c < LT; c++)
            if (c < (int)l) o[c] = acc[c];
    }
}

// ---- dense tall-skinny products on the matrix cores (subspace_iteration_full, svdapprox.rs:285-333; B = Q^T A :738)
// v_mfma_f32_32x32x2_f32: exact f32 (an fmaf chain), A operand lane -> A[i = lane & 31][k = lane >> 5],
// B operand lane -> B[k = lane >> 5][j = lane & 31], C: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
// The panel width l <= 32 is padded to the 32 MFMA columns.  Both kernels stream A once: HBM-bound (16 flop / B with
// the padded panel, below the ~20 flop/B ridge of 157 TF / 8 TB/s).
using f32x16 = __attribute__((ext_vector_type(16))) float;

// Y[m x l] = A[m x n] * X[n x l].  A is row-major with the contraction index contiguous, so the 32 x KT tile of each
// wave goes through LDS (coalesced 16-byte row segments in, conflict-free column reads out: row stride KT + 1).
constexpr int kMfmaKT = 64;
template <bool VEC4, int BM>
__global__ void __launch_bounds__(BM * 2) dense_mul_panel_mfma_kernel(const float* __restrict__ a, uint64_t m, uint64_t n,
                                                                   const float* __restrict__ x, float* __restrict__ y, uint32_t l) {
    constexpr int KT = kMfmaKT, NT = BM * 2, NQ = BM * KT / 4 / NT, NX = KT * 32 / NT;  // float4 / floats per thread and tile
    __shared__ float sA[BM * (KT + 1)];
    __shared__ float sX[KT * 32];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const uint64_t row_base = blockIdx.x * (uint64_t)BM;
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; q++) acc[q] = 0.f;
    float ra[NQ][4], rx[NX];
    auto gload = [&](uint64_t k0) {  // tile k0 into registers: 16 consecutive threads read 256 contiguous bytes of a row
#pragma unroll
        for (int q = 0; q < NQ; q++) {  // unconditional loads from clamped addresses (all in flight together), then masked
            const int idx = tid + q * NT;
            const int r = idx / (KT / 4), kc = (idx % (KT / 4)) * 4;
            const uint64_t row = row_base + r, kk = k0 + kc;
            const bool rin = row < m;
            const float* p = a + (rin ? row : 0) * n;
            if (VEC4) {  // n % 4 == 0: a float4 is entirely inside or outside the row
                const bool kin = kk < n;
                const float4 t = *reinterpret_cast<const float4*>(p + (kin ? kk : 0));
                const bool use = rin && kin;
                ra[q][0] = use ? t.x : 0.f; ra[q][1] = use ? t.y : 0.f; ra[q][2] = use ? t.z : 0.f; ra[q][3] = use ? t.w : 0.f;
            } else {
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    const bool kin = kk + t < n;
                    const float vv = p[kin ? kk + t : 0];
                    ra[q][t] = (rin && kin) ? vv : 0.f;
                }
            }
        }
#pragma unroll
        for (int q = 0; q < NX; q++) {  // X slab: KT k x 32 j (zero padded beyond l)
            const int idx = tid + q * NT;
            const int kk = idx >> 5, j = idx & 31;
            const bool in = k0 + kk < n && (uint32_t)j < l;
            const float vv = x[in ? (k0 + kk) * l + j : 0];
            rx[q] = in ? vv : 0.f;
        }
    };
    gload(0);
    for (uint64_t k0 = 0; k0 < n; k0 += KT) {
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            const int idx = tid + q * NT;
            float* d = sA + (idx / (KT / 4)) * (KT + 1) + (idx % (KT / 4)) * 4;
            d[0] = ra[q][0]; d[1] = ra[q][1]; d[2] = ra[q][2]; d[3] = ra[q][3];
        }
#pragma unroll
        for (int q = 0; q < NX; q++) sX[tid + q * NT] = rx[q];
        __syncthreads();
        if (k0 + KT < n) gload(k0 + KT);  // the next tile is in flight while the matrix cores work on this one
        const float* pa = sA + (w * 32 + (lane & 31)) * (KT + 1) + (lane >> 5);
        const float* pb = sX + (lane >> 5) * 32 + (lane & 31);
#pragma unroll
        for (int kk = 0; kk < KT; kk += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[kk], pb[kk * 32], acc, 0, 0, 0);
        __syncthreads();
    }
    const int j = lane & 31;
    if ((uint32_t)j < l) {
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const uint64_t row = row_base + w * 32 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
            if (row < m) y[row * l + j] = acc[q];
        }
    }
}

// The same product with NO workgroup barrier in the contraction loop (round 3).  The whole panel X (n x l floats: 63 KB at
// 784 x 20) is staged in LDS once per workgroup; afterwards every wave is on its own: lane (i = lane % 32, h = lane / 32) streams
// the float4 A[row_i][k0 + 4 h .. + 3] of ITS row straight from memory -- 32 bytes per row and instruction, the rest of the 128-byte
// line comes from the caches on the next loads -- and feeds four MFMAs with it: MFMA t contracts k = k0 + 4 h + t, whose X operand
// (lane (j, h) -> X[k0 + 4 h + t][j]) comes from LDS.  Any assignment of the k's to (trip, h, t) is a valid contraction order as
// long as both operands agree.
//  * LDS layout of the panel: [k / 4][j][k % 4] -- the four X operands of a float4 of A are ONE 16-byte LDS read;
//  * two register sets, no copies: while the MFMAs of one trip run, the loads of the next trip are in flight and those of the trip
//    after it are being issued (the first form copied `next` into `current` at the end of a trip, which waits for every load);
//  * the panel's loads are all issued before the first LDS store (one load per loop trip with its wait was ~10 us of 47);
//  * rows beyond m and columns beyond l are computed on clamped addresses and never stored: no select in the loop;
//  * KS = 2: the contraction is split over two waves per 32-row group (wave (g, s) contracts the s-th half of the k range; the halves
//    are added through LDS at the end, in a fixed order) -- 16 waves per CU instead of 8 at the same 63 KB of LDS per workgroup.
// Tried and measured slower: fully coalesced loads (8 lanes per 128-byte line) through a wave-private LDS tile (64 us: 80 KB of LDS per
// workgroup, one per CU); non-temporal loads (63-66 us: the later loads of a row's 128-byte line must come from the caches).
// Requires n % 8 == 0, l % 4 == 0 and n * l * 4 <= 64 KB of LDS; 128 rows per workgroup, two workgroups per CU.
template <int U, int KS>
__global__ void __launch_bounds__(256 * KS) dense_mul_panel_mfma_stream_kernel(const float* __restrict__ a, uint64_t m, uint64_t n,
                                                                               const float* __restrict__ x, float* __restrict__ y, uint32_t l) {
    extern __shared__ __attribute__((aligned(16))) float sXs[];  // [n / 4][l][4] (KS = 2: at least 4 x 64 x 16 floats, the partial sums' exchange)
    const int tid = threadIdx.x, lane = tid & 63, w = (tid >> 6) & 3, ks = tid >> 8;
    const int i = lane & 31, h = lane >> 5;
    // Round 6: a workgroup walks over row blocks blockIdx.x, + gridDim.x, ... (KS = 1; the launcher caps the grid): the panel is staged
    // ONCE per workgroup -- at 128 columns a block of 128 rows is 64 KB of A against 10 KB of panel and a barrier, and a wave lived for
    // four trips: 0.52 of HBM on a 6.25 M x 128 matrix.
    const uint64_t nblk = (m + 127) / 128;
    uint64_t blk = blockIdx.x;
    uint64_t row = blk * 128ull + (uint64_t)w * 32u + (uint64_t)i;
    // this wave's k range: [k_lo, k_hi), multiples of 8
    const uint64_t k_mid = KS == 2 ? (n / 16) * 8 : n;
    const uint64_t k_lo = ks ? k_mid : 0, k_hi = ks ? n : k_mid;
    const float* pa = a + (row < m ? row : 0) * n + 4 * h;
    const uint64_t full = (k_hi - k_lo) / (8 * U);  // whole trips of U loads
    float4 buf0[U], buf1[U];
    auto issue = [&](float4 (&buf)[U], uint64_t k) {
#pragma unroll
        for (int u = 0; u < U; u++) buf[u] = *reinterpret_cast<const float4*>(pa + k + 8 * u);
    };
    if (full) issue(buf0, k_lo);  // the first trip's loads do not wait for the panel
    const int nrem = (int)((k_hi - k_lo - full * 8 * U) / 8);  // float4 loads of the remainder (< U): issued now, used last
    float4 rem[U > 1 ? U - 1 : 1];
    auto issue_rem = [&]() {
#pragma unroll
        for (int u = 0; u < U - 1; u++)  // (u >= nrem: never used -- loaded from the start of the wave's k range, inside the row: with nrem == 0 the remainder's base is one past the row, for the last row of A past the allocation)
            rem[u] = *reinterpret_cast<const float4*>(pa + (u < nrem ? k_lo + full * 8 * U + 8 * u : k_lo));
    };
    issue_rem();
    {
        constexpr int XL = 4096 / (256 * KS);  // at most 64 KB = 4096 float4: XL per thread
        const uint64_t nl = n * l;
        float4 xs[XL];
#pragma unroll
        for (int q = 0; q < XL; q++) {
            const uint64_t idx = ((uint64_t)q * 256 * KS + tid) * 4;
            xs[q] = *reinterpret_cast<const float4*>(x + (idx < nl ? idx : nl - 4));
        }
#pragma unroll
        for (int q = 0; q < XL; q++) asm volatile("" : "+v"(xs[q].x), "+v"(xs[q].y), "+v"(xs[q].z), "+v"(xs[q].w));  // (keeps the loads above the stores)
#pragma unroll
        for (int q = 0; q < XL; q++) {
            const uint64_t idx = ((uint64_t)q * 256 * KS + tid) * 4;
            if (idx < nl) {
                const uint32_t k = (uint32_t)(idx / l), j = (uint32_t)(idx % l);  // x[k][j .. j + 3]  (l % 4 == 0)
                float* d = sXs + ((uint64_t)(k >> 2) * l + j) * 4 + (k & 3);
                d[0] = xs[q].x; d[4] = xs[q].y; d[8] = xs[q].z; d[12] = xs[q].w;
            }
        }
    }
    __syncthreads();
    const bool jin = (uint32_t)i < l;                 // (the B operand's lane index j = lane % 32)
    const float* pb = sXs + ((uint64_t)h * l + (jin ? i : 0)) * 4;   // + (k / 4) * l * 4 for the trip's k
    f32x16 acc;
    auto contract = [&](const float4 (&buf)[U], uint64_t k) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            const float4 b = *reinterpret_cast<const float4*>(pb + (k + 8 * u) * l);  // ((k + 8 u) / 4 + h) * l * 4 floats
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(buf[u].x, b.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(buf[u].y, b.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(buf[u].z, b.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(buf[u].w, b.w, acc, 0, 0, 0);
        }
    };
    for (;;) {   // one row block per turn (KS = 2: exactly one turn, the launcher gives every block its workgroup)
#pragma unroll
        for (int q = 0; q < 16; q++) acc[q] = 0.f;
        uint64_t k0 = k_lo, t = 0;
        for (; t + 2 < full; t += 2, k0 += 16 * U) {  // trips t and t + 1, and a trip t + 2 exists: every issue is unconditional
            issue(buf1, k0 + 8 * U);
            __builtin_amdgcn_sched_barrier(0);
            contract(buf0, k0);
            issue(buf0, k0 + 16 * U);
            __builtin_amdgcn_sched_barrier(0);
            contract(buf1, k0 + 8 * U);
        }
        if (t + 2 == full) {
            issue(buf1, k0 + 8 * U);
            __builtin_amdgcn_sched_barrier(0);
            contract(buf0, k0);
            contract(buf1, k0 + 8 * U);
            k0 += 16 * U;
        } else if (t + 1 == full) {  // its data sit in buf0
            contract(buf0, k0);
            k0 += 8 * U;
        }
#pragma unroll
        for (int u = 0; u < U - 1; u++) {  // remainder (loaded at the start)
            if (u < nrem) {
                const float4 b = *reinterpret_cast<const float4*>(pb + (k0 + 8 * u) * l);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(rem[u].x, b.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(rem[u].y, b.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(rem[u].z, b.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(rem[u].w, b.w, acc, 0, 0, 0);
            }
        }
        const uint64_t rb = blk * 128ull + (uint64_t)w * 32u;   // this turn's rows
        if constexpr (KS == 1) {   // the next block's first loads travel while this block's results are stored
            blk += gridDim.x;
            if (blk < nblk) {
                row = blk * 128ull + (uint64_t)w * 32u + (uint64_t)i;
                pa = a + (row < m ? row : 0) * n + 4 * h;
                if (full) issue(buf0, k_lo);
                issue_rem();
            }
        }
        if constexpr (KS == 2) {  // upper half -> LDS -> lower half adds (low + high: one fixed order)
            __syncthreads();      // every wave is done with the panel
            float* ex = sXs + ((uint64_t)w * 64 + lane) * 16;
            if (ks) {
#pragma unroll
                for (int q = 0; q < 16; q += 4) *reinterpret_cast<float4*>(ex + q) = make_float4(acc[q], acc[q + 1], acc[q + 2], acc[q + 3]);
            }
            __syncthreads();
            if (ks) return;
#pragma unroll
            for (int q = 0; q < 16; q += 4) {
                const float4 v = *reinterpret_cast<const float4*>(ex + q);
                acc[q] += v.x; acc[q + 1] += v.y; acc[q + 2] += v.z; acc[q + 3] += v.w;
            }
        }
        if (jin) {
#pragma unroll
            for (int q = 0; q < 16; q++) {
                const uint64_t r = rb + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
                if (r < m) y[r * l + i] = acc[q];
            }
        }
        if (KS == 2 || blk >= nblk) break;
    }
}

// partial[chunk][n x l] = A[rows of chunk]^T * X[rows of chunk].  A workgroup owns a strip of 128 columns of A and a
// chunk of rows.  Lane (h = lane / 32, i = lane % 32) loads the float4 A[row + h][c0 + 4 i .. 4 i + 3] -- 512 contiguous
// bytes per row, one 1 KB load instruction per 2 rows -- and feeds FOUR MFMAs: MFMA q multiplies the strided column set
// {c0 + 4 i + q} (its 32 "rows" of the A^T operand) with the X rows, so no cross-lane movement is needed; only the
// output rows are strided.  The X operand (2 rows x 32 padded columns) comes straight from L2.
template <bool VEC4>
__global__ void __launch_bounds__(256) dense_t_mul_panel_mfma_kernel(const float* __restrict__ a, uint64_t m, uint64_t n,
                                                                     const float* __restrict__ x, float* __restrict__ partial,
                                                                     uint32_t l, uint64_t rows_per_chunk) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint64_t c0 = blockIdx.x * 128ull;
    const uint64_t chunk = blockIdx.y;
    const uint64_t i0 = chunk * rows_per_chunk;
    const uint64_t i1 = i0 + rows_per_chunk < m ? i0 + rows_per_chunk : m;
    const uint64_t col = c0 + 4ull * (lane & 31);
    const int kh = lane >> 5;
    const bool jok = (uint32_t)(lane & 31) < l;
    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
        for (int q = 0; q < 16; q++) acc[t][q] = 0.f;
    // the 4 waves of the workgroup interleave the rows of the chunk; their accumulators are summed through LDS
    // U steps per trip: first ALL the loads of the trip (clamped addresses, no branches), then the masking and the
    // MFMAs -- 2 U row pairs (U KB of A) in flight per wave; the scheduling barrier keeps the compiler from pairing each
    // load with its use again (double-buffering the trips on top of this was measured slower)
    constexpr int U = 8;
    const uint64_t rbeg = i0 + 2 * (uint64_t)__builtin_amdgcn_readfirstlane(w);
    const bool cin = col < n;           // VEC4: n % 4 == 0, a float4 is entirely inside or outside the row
    const uint64_t colc = cin ? col : 0;
    const uint32_t jc = jok ? (uint32_t)(lane & 31) : 0u;
    for (uint64_t r0 = rbeg; r0 < i1; r0 += 8 * U) {
        float av[U][4], bx[U];
        bool ok[U];
#pragma unroll
        for (int uu = 0; uu < U; uu++) {
            const uint64_t rr = r0 + 8 * (uint64_t)uu + kh;
            ok[uu] = rr < i1;
            const uint64_t rc = ok[uu] ? rr : i0;
            if constexpr (VEC4) {
                const float4 t4 = *reinterpret_cast<const float4*>(a + rc * n + colc);
                av[uu][0] = t4.x; av[uu][1] = t4.y; av[uu][2] = t4.z; av[uu][3] = t4.w;
            } else {
#pragma unroll
                for (int t = 0; t < 4; t++) av[uu][t] = a[rc * n + (col + t < n ? col + t : 0)];
            }
            bx[uu] = x[rc * l + jc];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int uu = 0; uu < U; uu++) {
            const float bv = (ok[uu] && jok) ? bx[uu] : 0.f;
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const bool use = ok[uu] && (VEC4 ? cin : col + t < n);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(use ? av[uu][t] : 0.f, bv, acc[t], 0, 0, 0);
            }
        }
    }
    __shared__ float red[4 * 64 * 16];
#pragma unroll
    for (int t = 0; t < 4; t++) {  // one strided column set at a time through the reduction buffer
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 16; q++) red[(w * 16 + q) * 64 + lane] = acc[t][q];
        __syncthreads();
        if (w == 0 && jok) {
#pragma unroll
            for (int q = 0; q < 16; q++) {
                const float v = (red[q * 64 + lane] + red[(16 + q) * 64 + lane]) + (red[(32 + q) * 64 + lane] + red[(48 + q) * 64 + lane]);
                const uint64_t irow = (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);  // row of the MFMA output tile
                const uint64_t crow = c0 + 4 * irow + t;                          // = column of A (strided set t)
                if (crow < n) partial[(chunk * n + crow) * l + (lane & 31)] = v;
            }
        }
    }
}

// The same product with two register sets (round 3; n % 4 == 0): while the MFMAs of one trip run, the loads of the next trip are in
// flight and those of the trip after it are being issued -- the form above issues a trip's loads only after the previous trip's
// MFMAs and relies on the other waves to cover the gap.  Chunks sized so that the grid is about two workgroups per CU (a wave then
// runs ~20 trips instead of 2); rows beyond the chunk are read from clamped addresses and contracted against a zero X operand.
__global__ void __launch_bounds__(256) dense_t_mul_panel_mfma_pp_kernel(const float* __restrict__ a, uint64_t m, uint64_t n,
                                                                           const float* __restrict__ x, float* __restrict__ partial,
                                                                           uint32_t l, uint64_t rows_per_chunk) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint64_t c0 = blockIdx.x * 128ull;
    const uint64_t chunk = blockIdx.y;
    const uint64_t i0 = chunk * rows_per_chunk;
    const uint64_t i1 = i0 + rows_per_chunk < m ? i0 + rows_per_chunk : m;
    const uint64_t col = c0 + 4ull * (lane & 31);
    const int kh = lane >> 5;
    const bool jok = (uint32_t)(lane & 31) < l;
    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
        for (int q = 0; q < 16; q++) acc[t][q] = 0.f;
    constexpr int U = 4;
    const uint64_t colc = col < n ? col : 0;           // (a strip's columns beyond n are computed on column 0 and never stored)
    const uint32_t jc = jok ? (uint32_t)(lane & 31) : 0u;
    const float* pa = a + colc;
    const float* px = x + jc;
    float4 a0[U], a1[U];
    float b0[U], b1[U];
    auto issue = [&](float4 (&av)[U], float (&bv)[U], uint64_t r0) {
#pragma unroll
        for (int uu = 0; uu < U; uu++) {
            const uint64_t rr = r0 + 8 * (uint64_t)uu + kh;
            const uint64_t rc = rr < i1 ? rr : i0;
            av[uu] = *reinterpret_cast<const float4*>(pa + rc * n);
            bv[uu] = px[rc * l];
        }
    };
    auto contract = [&](const float4 (&av)[U], const float (&bv)[U], uint64_t r0) {
#pragma unroll
        for (int uu = 0; uu < U; uu++) {
            const float b = (r0 + 8 * (uint64_t)uu + kh < i1) ? bv[uu] : 0.f;
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[uu].x, b, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[uu].y, b, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[uu].z, b, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[uu].w, b, acc[3], 0, 0, 0);
        }
    };
    // the 4 waves of the workgroup interleave the rows of the chunk (wave w: row pairs 2 w, 2 w + 8, ...); every issue is unconditional
    uint64_t r0 = i0 + 2 * (uint64_t)__builtin_amdgcn_readfirstlane(w);
    issue(a0, b0, r0);
    for (; r0 < i1; r0 += 16 * U) {  // (a third register set, two trips in flight behind the contraction, measured no faster: 47.6-49.2 us)
        issue(a1, b1, r0 + 8 * U);
        __builtin_amdgcn_sched_barrier(0);
        contract(a0, b0, r0);
        issue(a0, b0, r0 + 16 * U);
        __builtin_amdgcn_sched_barrier(0);
        contract(a1, b1, r0 + 8 * U);
    }
    __shared__ float red[4 * 64 * 16];
#pragma unroll
    for (int t = 0; t < 4; t++) {  // one strided column set at a time through the reduction buffer
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 16; q++) red[(w * 16 + q) * 64 + lane] = acc[t][q];
        __syncthreads();
        if (w == 0 && jok) {
#pragma unroll
            for (int q = 0; q < 16; q++) {
                const float v = (red[q * 64 + lane] + red[(16 + q) * 64 + lane]) + (red[(32 + q) * 64 + lane] + red[(48 + q) * 64 + lane]);
                const uint64_t irow = (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);  // row of the MFMA output tile
                const uint64_t crow = c0 + 4 * irow + t;                          // = column of A (strided set t)
                if (crow < n) partial[(chunk * n + crow) * l + (lane & 31)] = v;
            }
        }
    }
}

__global__ void reduce_chunks_kernel(const float* __restrict__ partial, uint64_t chunks, uint64_t count, float* __restrict__ out) {
    uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i >= count) return;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f, s5 = 0.f, s6 = 0.f, s7 = 0.f;  // fixed 8-way interleave: deterministic, eight loads in flight
    uint64_t r = 0;
    for (; r + 8 <= chunks; r += 8) {
        s0 += partial[r * count + i];
        s1 += partial[(r + 1) * count + i];
        s2 += partial[(r + 2) * count + i];
        s3 += partial[(r + 3) * count + i];
        s4 += partial[(r + 4) * count + i];
        s5 += partial[(r + 5) * count + i];
        s6 += partial[(r + 6) * count + i];
        s7 += partial[(r + 7) * count + i];
    }
    for (; r < chunks; r++) s0 += partial[r * count + i];
    out[i] = ((s0 + s1) + (s2 + s3)) + ((s4 + s5) + (s6 + s7));
}

constexpr int kGramTile = 64;

// (A row-per-thread form of the kernel below -- 256 rows per trip, the row in f64 registers, as chol_apply_kernel -- was measured in round 6
// and is not kept: 1.30 ms against 1.01 ms on a 6.25 M x 20 panel.)
// out tile = y tile * M ; each workgroup owns kGramTile rows, staged through LDS so that in-place is safe
__global__ void __launch_bounds__(256) apply_panel_kernel(const float* __restrict__ y, uint64_t rows, uint32_t l,
                                                          const double* __restrict__ mat, uint32_t lout, float* __restrict__ out) {
    __shared__ float tile[kGramTile * kMaxL];
    __shared__ double sm[kMaxL * kMaxL];
    for (uint32_t idx = threadIdx.x; idx < l * lout; idx += 256) sm[idx] = mat[idx];
    const uint64_t ntiles = (rows + kGramTile - 1) / kGramTile;
    for (uint64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const uint64_t r0 = t * kGramTile;
        const uint32_t nr = (uint32_t)(rows - r0 < kGramTile ? rows - r0 : kGramTile);
        __syncthreads();
        for (uint32_t idx = threadIdx.x; idx < nr * l; idx += 256) tile[idx] = y[r0 * l + idx];
        __syncthreads();
        for (uint32_t idx = threadIdx.x; idx < nr * lout; idx += 256) {
            const uint32_t r = idx / lout, c2 = idx % lout;
            double s = 0.;
            for (uint32_t c = 0; c < l; c++) s += (double)tile[r * l + c] * sm[c * lout + c2];
            out[(r0 + r) * lout + c2] = (float)s;
        }
    }
}

// cyclic Jacobi with round-robin parallel ordering on one workgroup; G is destroyed in LDS.
__global__ void __launch_bounds__(256) jacobi_eigh_kernel(const double* __restrict__ gin, uint32_t l, double* __restrict__ evals,
                                                          double* __restrict__ evecs, const int* __restrict__ run_if) {
    if (run_if && *run_if == 0) return;  // the Cholesky route succeeded: nothing to do
    __shared__ double G[kMaxL * kMaxL];
    __shared__ double V[kMaxL * kMaxL];
    __shared__ double cs[kMaxL];  // c,s per pair
    __shared__ int pp[kMaxL / 2], pq[kMaxL / 2];
    __shared__ double offn;
    __shared__ int order[kMaxL];
    const uint32_t L = (l + 1) & ~1u;  // even
    const uint32_t tid = threadIdx.x;
    for (uint32_t idx = tid; idx < l * l; idx += blockDim.x) {
        G[idx] = gin[idx];
        V[idx] = (idx / l == idx % l) ? 1. : 0.;
    }
    __syncthreads();
    for (int sweep = 0; sweep < 30; sweep++) {
        {   // off-diagonal vs diagonal mass, reduced by the first wave (l*l <= 4096 entries)
            double off = 0., diag = 0.;
            if (tid < 64) {
                for (uint32_t idx = tid; idx < l * l; idx += 64) {
                    const double v = G[idx];
                    if (idx / l == idx % l) diag += v * v; else off += v * v;
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) { off += __shfl_xor(off, o); diag += __shfl_xor(diag, o); }
                // converged when the off-diagonal mass is at f64 roundoff of the diagonal (relative 1e-13 in norm)
                if (tid == 0) offn = (off <= 1e-26 * diag || off == 0.) ? 0. : off;
            }
        }
        __syncthreads();
        if (offn == 0.) break;
        for (uint32_t step = 0; step + 1 < L; step++) {
            // round-robin pairing of 0..L-1
            if (tid < L / 2) {
                uint32_t a, b;
                if (tid == 0) { a = L - 1; b = step; }
                else { a = (step + tid) % (L - 1); b = (step + (L - 1) - tid) % (L - 1); }
                if (a > b) { uint32_t t = a; a = b; b = t; }
                pp[tid] = (int)a; pq[tid] = (int)b;
                double c = 1., s = 0.;
                if (b < l) {
                    const double apq = G[a * l + b];
                    if (apq != 0.) {
                        const double tau = (G[b * l + b] - G[a * l + a]) / (2. * apq);
                        const double t = (tau >= 0. ? 1. : -1.) / (fabs(tau) + sqrt(1. + tau * tau));
                        c = 1. / sqrt(1. + t * t);
                        s = t * c;
                    }
                }
                cs[2 * tid] = c; cs[2 * tid + 1] = s;
            }
            __syncthreads();
            // columns: G <- G J, V <- V J
            for (uint32_t idx = tid; idx < l * (L / 2); idx += blockDim.x) {
                const uint32_t r = idx / (L / 2), pr = idx % (L / 2);
                const uint32_t a = pp[pr], b = pq[pr];
                if (b >= l) continue;
                const double c = cs[2 * pr], s = cs[2 * pr + 1];
                const double ga = G[r * l + a], gb = G[r * l + b];
                G[r * l + a] = c * ga - s * gb;
                G[r * l + b] = s * ga + c * gb;
                const double va = V[r * l + a], vb = V[r * l + b];
                V[r * l + a] = c * va - s * vb;
                V[r * l + b] = s * va + c * vb;
            }
            __syncthreads();
            // rows: G <- J^T G
            for (uint32_t idx = tid; idx < l * (L / 2); idx += blockDim.x) {
                const uint32_t k = idx / (L / 2), pr = idx % (L / 2);
                const uint32_t a = pp[pr], b = pq[pr];
                if (b >= l) continue;
                const double c = cs[2 * pr], s = cs[2 * pr + 1];
                const double ga = G[a * l + k], gb = G[b * l + k];
                G[a * l + k] = c * ga - s * gb;
                G[b * l + k] = s * ga + c * gb;
            }
            __syncthreads();
        }
    }
    if (tid == 0) {  // sort eigenvalues descending (stable selection)
        bool used[kMaxL];
        for (uint32_t a = 0; a < l; a++) used[a] = false;
        for (uint32_t o = 0; o < l; o++) {
            int best = -1;
            for (uint32_t a = 0; a < l; a++)
                if (!used[a] && (best < 0 || G[a * l + a] > G[best * l + best])) best = (int)a;
            used[best] = true;
            order[o] = best;
        }
    }
    __syncthreads();
    for (uint32_t idx = tid; idx < l * l; idx += blockDim.x) {
        const uint32_t r = idx / l, o = idx % l;
        evecs[idx] = V[r * l + order[o]];
    }
    if (tid < l) evals[tid] = G[order[tid] * l + order[tid]];
}

// Cholesky attempt on the Gram matrix (one lane, l <= 64: a few thousand f64 flops).  G = R^T R, M = R^-1 so that
// Y M has orthonormal columns.  flag[0] = 1 when a pivot is not safely positive (rank-deficient / ill-conditioned
// panel): the caller's next kernels then take the eigen (SVQB) route instead.
__global__ void __launch_bounds__(64) chol_inverse_kernel(const double* __restrict__ g, uint32_t l, double rel_tol, double* __restrict__ m,
                                                          int* __restrict__ flag) {
    // one wave: lane i owns column i of R during the factorisation and column i of M = R^-1 afterwards
    __shared__ double R[kMaxL * kMaxL];
    __shared__ int s_bad;
    const uint32_t i = threadIdx.x;
    double dmax = 0.;
    for (uint32_t q = 0; q < l; q++) dmax = g[q * l + q] > dmax ? g[q * l + q] : dmax;
    if (i == 0) s_bad = dmax > 0. ? 0 : 1;
    __syncthreads();
    for (uint32_t j = 0; j < l; j++) {  // upper Cholesky G = R^T R, row j of R per step
        double v = 0.;
        if (i >= j && i < l) {
            v = g[j * l + i];
            for (uint32_t k = 0; k < j; k++) v -= R[k * l + j] * R[k * l + i];
        }
        const double d = __shfl(v, (int)j);
        if (!(d > rel_tol * dmax)) { if (i == 0) s_bad = 1; break; }  // wave-uniform
        const double rjj = sqrt(d);
        if (i >= j && i < l) R[j * l + i] = (i == j) ? rjj : v / rjj;
        __syncthreads();
    }
    __syncthreads();
    const bool ok = s_bad == 0;
    if (ok && i < l) {  // column i of M = R^-1 by back substitution
        double col[kMaxL];
        for (uint32_t q = 0; q < l; q++) col[q] = 0.;
        col[i] = 1. / R[i * l + i];
        for (int r = (int)i - 1; r >= 0; r--) {
            double v = 0.;
            for (uint32_t k = r + 1; k <= i; k++) v -= R[r * l + k] * col[k];
            col[r] = v / R[r * l + r];
        }
        for (uint32_t q = 0; q < l; q++) m[q * l + i] = col[q];
    }
    if (i == 0) *flag = ok ? 0 : 1;
}

// M[c][o] = evecs[c][o] * (evals[o] > tol ? evals[o]^-1/2 : 0)
__global__ void svqb_scale_kernel(const double* __restrict__ evals, const double* __restrict__ evecs, uint32_t l, double rel_tol,
                                  double* __restrict__ m, const int* __restrict__ run_if) {
    uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= l * l) return;
    if (run_if && *run_if == 0) return;
    const uint32_t o = idx % l;
    const double lam = evals[o], lmax = evals[0];
    m[idx] = (lam > rel_tol * lmax && lam > 0.) ? evecs[idx] / sqrt(lam) : 0.;
}

__global__ void __launch_bounds__(1024) seq_sum_kernel(const float* __restrict__ x, uint64_t n, uint64_t stride, float* __restrict__ out) {
    // exact left-to-right f32 sum (the reference's iter().sum::<f32>() / fold order): lane 0 adds the values in order out of
    // LDS, always 16 values ahead in registers (the add chain, not the LDS latency, sets the pace), while waves 1..15 stage
    // the NEXT chunk into the other buffer with all their loads in flight together -- the staging (one memory round trip
    // per 256 values with the former 256-thread loop) was 3/4 of the time, not the chain
    constexpr int CH = 6144, LOADERS = 960;
    __shared__ __attribute__((aligned(16))) float buf[2][CH + 16];
    const uint64_t nchunks = (n + CH - 1) / CH;
    auto fill = [&](int b, uint64_t c) {  // threads 64..1023
        const uint64_t c0 = c * CH;
        const uint32_t m = (uint32_t)((n - c0) < (uint64_t)CH ? (n - c0) : (uint64_t)CH);
        const uint32_t t = threadIdx.x - 64u;
        if (stride == 1 && m == (uint32_t)CH && ((reinterpret_cast<uintptr_t>(x + c0) & 15u) == 0)) {
            const float4* src = reinterpret_cast<const float4*>(x + c0);
            float4 v[2];
#pragma unroll
            for (int r = 0; r < 2; r++) { const uint32_t j = t + (uint32_t)r * LOADERS; v[r] = j < CH / 4 ? src[j] : make_float4(0.f, 0.f, 0.f, 0.f); }
#pragma unroll
            for (int r = 0; r < 2; r++) { const uint32_t j = t + (uint32_t)r * LOADERS; if (j < CH / 4) reinterpret_cast<float4*>(buf[b])[j] = v[r]; }
        } else {
            float v[7];
#pragma unroll
            for (int r = 0; r < 7; r++) { const uint32_t i = t + (uint32_t)r * LOADERS; v[r] = i < m ? x[(c0 + i) * stride] : 0.f; }
#pragma unroll
            for (int r = 0; r < 7; r++) { const uint32_t i = t + (uint32_t)r * LOADERS; if (i < (uint32_t)CH) buf[b][i] = v[r]; }
        }
    };
    if (threadIdx.x >= 64 && nchunks) fill(0, 0);
    __syncthreads();
    float s = 0.f;
    for (uint64_t c = 0; c < nchunks; c++) {
        if (threadIdx.x >= 64 && c + 1 < nchunks) fill((int)((c + 1) & 1), c + 1);
        if (threadIdx.x == 0) {
            const uint64_t c0 = c * CH;
            const uint32_t m = (uint32_t)((n - c0) < (uint64_t)CH ? (n - c0) : (uint64_t)CH);
            const float* bb = buf[c & 1];
            const float4* b4 = reinterpret_cast<const float4*>(bb);
            // two register sets in turn (no copies: the compiler would fold a copied "next" set back into a load at the
            // top of the trip and wait for it there); the scheduling barriers keep each set's loads ahead of the other's adds
            float4 a0 = b4[0], a1 = b4[1], a2 = b4[2], a3 = b4[3], c0_, c1_, c2_, c3_;
#define AE_ADD16(p0, p1, p2, p3) s += p0.x; s += p0.y; s += p0.z; s += p0.w; s += p1.x; s += p1.y; s += p1.z; s += p1.w; \
                                 s += p2.x; s += p2.y; s += p2.z; s += p2.w; s += p3.x; s += p3.y; s += p3.z; s += p3.w;
            uint32_t i = 0;
            for (; i + 32 <= m; i += 32) {
                c0_ = b4[i / 4 + 4]; c1_ = b4[i / 4 + 5]; c2_ = b4[i / 4 + 6]; c3_ = b4[i / 4 + 7];
                __builtin_amdgcn_sched_barrier(0);
                AE_ADD16(a0, a1, a2, a3)
                __builtin_amdgcn_sched_barrier(0);
                a0 = b4[i / 4 + 8]; a1 = b4[i / 4 + 9]; a2 = b4[i / 4 + 10]; a3 = b4[i / 4 + 11];  // (padding past m)
                __builtin_amdgcn_sched_barrier(0);
                AE_ADD16(c0_, c1_, c2_, c3_)
                __builtin_amdgcn_sched_barrier(0);
            }
            if (i + 16 <= m) { AE_ADD16(a0, a1, a2, a3) i += 16; }
#undef AE_ADD16
            for (; i < m; i++) s += bb[i];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = s;
}

// ndarray's Array1::sum() order (numeric_util::unrolled_fold: eight interleaved accumulators, then
// ((p0+p4) + (p1+p5)) ... and the tail), used by the reference for q.sum() (diffmaps.rs:469, :546, :889, :932).
// Eight lanes run the eight chains.
__global__ void __launch_bounds__(1024) ndarray_sum_kernel(const float* __restrict__ x, uint64_t n, float* __restrict__ out) {
    constexpr int CH = 6144, LOADERS = 960;  // staging by waves 1..15 into the other buffer while the eight lanes add (see seq_sum_kernel)
    __shared__ float buf[2][CH];
    __shared__ float p8[8];
    float p = 0.f;
    const uint64_t n8 = n & ~7ull;
    const uint64_t nchunks = (n8 + CH - 1) / CH;
    auto fill = [&](int b, uint64_t c) {  // threads 64..1023
        const uint64_t c0 = c * CH;
        const uint32_t m = (uint32_t)((n8 - c0) < (uint64_t)CH ? (n8 - c0) : (uint64_t)CH);
        const uint32_t t = threadIdx.x - 64u;
        float v[7];
#pragma unroll
        for (int r = 0; r < 7; r++) { const uint32_t i = t + (uint32_t)r * LOADERS; v[r] = i < m ? x[c0 + i] : 0.f; }
#pragma unroll
        for (int r = 0; r < 7; r++) { const uint32_t i = t + (uint32_t)r * LOADERS; if (i < (uint32_t)CH) buf[b][i] = v[r]; }
    };
    if (threadIdx.x >= 64 && nchunks) fill(0, 0);
    __syncthreads();
    for (uint64_t c = 0; c < nchunks; c++) {
        if (threadIdx.x >= 64 && c + 1 < nchunks) fill((int)((c + 1) & 1), c + 1);
        if (threadIdx.x < 8) {
            const uint64_t c0 = c * CH;
            const uint32_t m = (uint32_t)((n8 - c0) < (uint64_t)CH ? (n8 - c0) : (uint64_t)CH);
            const float* bb = buf[c & 1];
            uint32_t i = threadIdx.x;
            for (; i + 56 < m; i += 64) {
                float v[8];
#pragma unroll
                for (int q = 0; q < 8; q++) v[q] = bb[i + 8 * q];
#pragma unroll
                for (int q = 0; q < 8; q++) p += v[q];
            }
            for (; i < m; i += 8) p += bb[i];
        }
        __syncthreads();
    }
    if (threadIdx.x < 8) p8[threadIdx.x] = p;
    __syncthreads();
    if (threadIdx.x == 0) {
        float acc = 0.f;
        acc = acc + (p8[0] + p8[4]);
        acc = acc + (p8[1] + p8[5]);
        acc = acc + (p8[2] + p8[6]);
        acc = acc + (p8[3] + p8[7]);
        for (uint64_t i = n8; i < n; i++) acc = acc + x[i];
        *out = acc;
    }
}

// transpose support ---------------------------------------------------------------------------
__global__ void coo_keys_kernel(uint64_t m, const uint64_t* __restrict__ indptr, const uint32_t* __restrict__ ind,
                                uint64_t* __restrict__ keys, uint32_t* __restrict__ payload) {
    uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i >= m) return;
    for (uint64_t e = indptr[i]; e < indptr[i + 1]; e++) {
        keys[e] = ((uint64_t)ind[e] << 32) | i;  // (col, row)
        payload[e] = (uint32_t)e;
    }
}
__global__ void transpose_fill_kernel(uint64_t nnz, const uint64_t* __restrict__ keys, const uint32_t* __restrict__ perm,
                                      const float* __restrict__ val, uint32_t* __restrict__ tind, float* __restrict__ tval) {
    uint64_t e = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (e >= nnz) return;
    tind[e] = (uint32_t)(keys[e] & 0xFFFFFFFFull);
    tval[e] = val[perm[e]];
}
// rowptr[r] = first sorted position whose key's high word >= r
__global__ void rowptr_from_sorted_keys_kernel(const uint64_t* __restrict__ keys, uint64_t nnz, uint64_t nrows,
                                               uint64_t* __restrict__ rowptr) {
    uint64_t r = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (r > nrows) return;
    uint64_t lo = 0, hi = nnz;
    while (lo < hi) {
        uint64_t mid = (lo + hi) >> 1;
        if ((keys[mid] >> 32) < r) lo = mid + 1; else hi = mid;
    }
    rowptr[r] = lo;
}
__global__ void transpose_dense_kernel(const float* __restrict__ in, uint64_t rows, uint64_t cols, float* __restrict__ out) {
    uint64_t idx = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (idx >= rows * cols) return;
    uint64_t r = idx / cols, c = idx % cols;
    out[c * rows + r] = in[idx];
}

}  // namespace

namespace ae {

void sort_pairs_u64_u32(uint64_t* d_keys_in, uint64_t* d_keys_out, uint32_t* d_vals_in, uint32_t* d_vals_out, uint64_t count) {
    size_t tmp_bytes = 0;
    if (rocprim::radix_sort_pairs(nullptr, tmp_bytes, d_keys_in, d_keys_out, d_vals_in, d_vals_out, count, 0, 64, stream()) != hipSuccess)
        fail(AE_ERR_NO_DEVICE, "rocprim radix_sort_pairs (size query) failed");
    DevBuf<char> tmp(tmp_bytes ? tmp_bytes : 1);
    if (rocprim::radix_sort_pairs(tmp.p, tmp_bytes, d_keys_in, d_keys_out, d_vals_in, d_vals_out, count, 0, 64, stream()) != hipSuccess)
        fail(AE_ERR_NO_DEVICE, "rocprim radix_sort_pairs failed");
    sync();
}

// keys only, bits [0, end_bit); no host synchronisation (scratch from the stream-ordered pool)
void sort_keys_u64(uint64_t* d_keys_in, uint64_t* d_keys_out, uint64_t count, unsigned end_bit) {
    size_t tmp_bytes = 0;
    if (rocprim::radix_sort_keys(nullptr, tmp_bytes, d_keys_in, d_keys_out, count, 0, end_bit, stream()) != hipSuccess)
        fail(AE_ERR_NO_DEVICE, "rocprim radix_sort_keys (size query) failed");
    DevBuf<char> tmp;
    tmp.alloc_pooled(tmp_bytes ? tmp_bytes : 1);
    if (rocprim::radix_sort_keys(tmp.p, tmp_bytes, d_keys_in, d_keys_out, count, 0, end_bit, stream()) != hipSuccess)
        fail(AE_ERR_NO_DEVICE, "rocprim radix_sort_keys failed");
}

// stable (LSD radix) sort of (key, value) pairs on the key bits [0, end_bit)
void sort_pairs_u32_u32(uint32_t* d_keys_in, uint32_t* d_keys_out, uint32_t* d_vals_in, uint32_t* d_vals_out, uint64_t count, unsigned end_bit) {
    size_t tmp_bytes = 0;
    if (rocprim::radix_sort_pairs(nullptr, tmp_bytes, d_keys_in, d_keys_out, d_vals_in, d_vals_out, count, 0, end_bit, stream()) != hipSuccess)
        fail(AE_ERR_NO_DEVICE, "rocprim radix_sort_pairs (size query) failed");
    DevBuf<char> tmp;
    tmp.alloc_pooled(tmp_bytes ? tmp_bytes : 1);
    if (rocprim::radix_sort_pairs(tmp.p, tmp_bytes, d_keys_in, d_keys_out, d_vals_in, d_vals_out, count, 0, end_bit, stream()) != hipSuccess)
        fail(AE_ERR_NO_DEVICE, "rocprim radix_sort_pairs failed");
}

void rowptr_from_sorted_keys(const uint64_t* d_keys, uint64_t nnz, uint64_t nrows, uint64_t* d_rowptr) {
    hipLaunchKernelGGL(rowptr_from_sorted_keys_kernel, dim3(blocks_for(nrows + 1, 256)), dim3(256), 0, stream(), d_keys, nnz, nrows,
                       d_rowptr);
    check_launch("rowptr_from_sorted_keys");
}

void gaussian_fill_device(float* d_out, uint64_t count, uint64_t seed, uint32_t tag) {
    hipLaunchKernelGGL(gaussian_fill_kernel, dim3(grid_cap((count + 3) / 4, 256)), dim3(256), 0, stream(), d_out, count, seed, tag);
    check_launch("gaussian_fill");
}

// ---- tree sums (TreeSums, linalg.h): column sums of a row-major n x dim array in f64, fixed launch shape -> reproducible
__global__ void __launch_bounds__(256) tree_sum_cols_partial_kernel(const float* __restrict__ x, uint64_t n, uint32_t dim, uint64_t stride,
                                                                     double* __restrict__ partial) {
    __shared__ double sh[256];
    // thread t adds elements t, t + T, ... of the flattened rows (T a multiple of dim: a thread stays in one column)
    const uint32_t per = 256u / dim * dim;  // active threads per workgroup
    double acc = 0.;
    if (threadIdx.x < per) {
        const uint64_t total = n * dim, T = (uint64_t)gridDim.x * per;
        for (uint64_t e = (uint64_t)blockIdx.x * per + threadIdx.x; e < total; e += T) acc += (double)x[(e / dim) * stride + e % dim];
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x < dim) {
        double s = 0.;
        for (uint32_t t = threadIdx.x; t < per; t += dim) s += sh[t];
        partial[(uint64_t)blockIdx.x * dim + threadIdx.x] = s;
    }
}
__global__ void __launch_bounds__(128) tree_sum_cols_final_kernel(const double* __restrict__ partial, uint32_t nblocks, uint32_t dim, float* __restrict__ out) {
    if (threadIdx.x >= dim) return;
    double s = 0.;
    for (uint32_t b = 0; b < nblocks; b++) s += partial[(uint64_t)b * dim + threadIdx.x];
    out[threadIdx.x] = (float)s;
}
// Three states per thread: no scope (-1: the process-wide default decides), an explicit scope asking for the reference order (0) or for
// tree sums (1).  An explicit scope always wins: Embedder::embed / EntropyOptim::new in the bit-exact mode get the reference order
// whatever ae_set_summation_order says.
static thread_local int g_tree_scope = -1;
static bool g_tree_sums_default = true;   // ae_set_summation_order: what the stage-level entry points do outside an Embedder
TreeSums::TreeSums(bool on) : prev(g_tree_scope) { g_tree_scope = on ? 1 : 0; }
TreeSums::~TreeSums() { g_tree_scope = prev; }
bool tree_sums() { return g_tree_scope >= 0 ? g_tree_scope == 1 : g_tree_sums_default; }
int tree_sums_scope() { return g_tree_scope; }
void set_tree_sums_default(bool on) { g_tree_sums_default = on; }
// x: n rows of `dim` consecutive values, `stride` floats apart
static void tree_sum_cols(const float* d_x, uint64_t n, uint32_t dim, uint64_t stride, float* host_out) {
    const uint32_t per = 256u / dim * dim;
    const unsigned nblocks = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((n * dim + per * 16ull - 1) / (per * 16ull), 1024));
    DevBuf<double> partial;
    DevBuf<float> out;
    partial.alloc_pooled((size_t)nblocks * dim);
    out.alloc_pooled(dim);
    hipLaunchKernelGGL(tree_sum_cols_partial_kernel, dim3(nblocks), dim3(256), 0, stream(), d_x, n, dim, stride, partial.p);
    check_launch("tree_sum_cols_partial");
    hipLaunchKernelGGL(tree_sum_cols_final_kernel, dim3(1), dim3(128), 0, stream(), (const double*)partial.p, nblocks, dim, out.p);
    check_launch("tree_sum_cols_final");
    out.download(host_out, dim);
}

float seq_sum_f32(const float* d_x, uint64_t n, uint64_t stride) {
    if (tree_sums()) { float h; tree_sum_cols(d_x, n, 1, stride, &h); return h; }
    static DevBuf<float> out;
    if (!out.n) out.alloc(1);
    hipLaunchKernelGGL(seq_sum_kernel, dim3(1), dim3(1024), 0, stream(), d_x, n, stride, out.p);
    check_launch("seq_sum");
    float h;
    out.download(&h, 1);
    return h;
}
// column sums of a row-major n x dim array, each column added in row order (the reference's `for i { means[j] += data[[i, j]] }`,
// embedder.rs:1391-1394): dim independent sequential chains, one lane each, over LDS-staged row blocks
__global__ void __launch_bounds__(1024) seq_sum_cols_kernel(const float* __restrict__ x, uint64_t n, uint32_t dim, float* __restrict__ out) {
    // lanes 0..dim-1 add their column in row order out of LDS, 8 rows ahead in registers; threads 256..1023 stage the next
    // row block into the other buffer meanwhile (all their loads in flight together)
    extern __shared__ __attribute__((aligned(16))) float cbuf[];  // 2 x (rpc + 8) x dim
    constexpr uint32_t CHF = 6144, LOADERS = 768;
    const uint32_t rpc = CHF / dim, bstride = (rpc + 8u) * dim;
    const uint64_t nchunks = (n + rpc - 1) / rpc;
    auto fill = [&](int b, uint64_t c) {  // threads 256..1023
        const uint64_t r0 = c * rpc;
        const uint32_t nr = (uint32_t)((n - r0) < (uint64_t)rpc ? (n - r0) : (uint64_t)rpc);
        const uint32_t t = threadIdx.x - 256u, cntf = nr * dim;
        const float* src = x + r0 * dim;
        float v[8];
#pragma unroll
        for (int r = 0; r < 8; r++) { const uint32_t i = t + (uint32_t)r * LOADERS; v[r] = i < cntf ? src[i] : 0.f; }
#pragma unroll
        for (int r = 0; r < 8; r++) { const uint32_t i = t + (uint32_t)r * LOADERS; if (i < bstride) cbuf[(uint32_t)b * bstride + i] = v[r]; }
    };
    if (threadIdx.x >= 256 && nchunks) fill(0, 0);
    __syncthreads();
    float s = 0.f;
    for (uint64_t c = 0; c < nchunks; c++) {
        if (threadIdx.x >= 256 && c + 1 < nchunks) fill((int)((c + 1) & 1), c + 1);
        if (threadIdx.x < dim) {
            const uint64_t r0 = c * rpc;
            const uint32_t nr = (uint32_t)((n - r0) < (uint64_t)rpc ? (n - r0) : (uint64_t)rpc);
            const float* bb = cbuf + (uint32_t)(c & 1) * bstride + threadIdx.x;
            float a[8], b2[8];  // two register sets in turn (see seq_sum_kernel)
#pragma unroll
            for (int q = 0; q < 8; q++) a[q] = bb[q * dim];
            uint32_t r = 0;
            for (; r + 16 <= nr; r += 16) {
#pragma unroll
                for (int q = 0; q < 8; q++) b2[q] = bb[(r + 8 + q) * dim];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 8; q++) s += a[q];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 8; q++) a[q] = bb[(r + 16 + q) * dim];  // (padding rows past nr)
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 8; q++) s += b2[q];
                __builtin_amdgcn_sched_barrier(0);
            }
            if (r + 8 <= nr) {
#pragma unroll
                for (int q = 0; q < 8; q++) s += a[q];
                r += 8;
            }
            for (; r < nr; r++) s += bb[r * dim];
        }
        __syncthreads();
    }
    if (threadIdx.x < dim) out[threadIdx.x] = s;
}
void seq_sum_cols_f32(const float* d_x, uint64_t n, uint32_t dim, float* host_out) {
    if (dim == 0 || dim > 128) fail(AE_ERR_INVALID_ARG, "seq_sum_cols: dimension %u unsupported", dim);
    if (tree_sums()) { tree_sum_cols(d_x, n, dim, dim, host_out); return; }
    DevBuf<float> out;
    out.alloc_pooled(dim);
    hipLaunchKernelGGL(seq_sum_cols_kernel, dim3(1), dim3(1024), sizeof(float) * 2u * (6144u / dim + 8u) * dim, stream(), d_x, n, dim, out.p);
    check_launch("seq_sum_cols");
    out.download(host_out, dim);
}

float ndarray_sum_f32(const float* d_x, uint64_t n) {
    if (tree_sums()) { float h; tree_sum_cols(d_x, n, 1, 1, &h); return h; }
    static DevBuf<float> out;
    if (!out.n) out.alloc(1);
    hipLaunchKernelGGL(ndarray_sum_kernel, dim3(1), dim3(1024), 0, stream(), d_x, n, out.p);
    check_launch("ndarray_sum");
    float h = 0.f;
    out.download(&h, 1);
    return h;
}

// l % 4 == 0, l <= 32: eight lanes per edge, each gathering one float4 of the X row (8 edges x 80 bytes per wave
// instruction for l = 20 instead of 2 edges), partial rows summed across the 8 edge slots by three xor-shuffle steps.
constexpr uint32_t kWideLd = 32;   // floats per row of a panel's gather copy (128 bytes)
// ldx: floats from row to row of the GATHERED operand x (l, or 32: rows of 128 bytes on 128-byte boundaries -- a gathered row is then ONE
// line where an 80-byte row at an 80-byte stride straddles two in 62 % of the cases; the product is bound by the number of lines it asks
// the memory system for, tools/ubench_rowgather.hip)
template <int U>
__global__ void __launch_bounds__(256) spmm_csr_vec4_kernel(uint64_t m, const uint64_t* __restrict__ indptr, const uint32_t* __restrict__ ind,
                                                            const float* __restrict__ val, const float* __restrict__ x, float* __restrict__ y,
                                                            uint32_t l, uint32_t ldx) {
    const int lane = threadIdx.x & 63;
    const int sl = lane >> 3, q = lane & 7;
    const bool active = (uint32_t)(4 * q) < l;
    const uint32_t qo = active ? 4u * (uint32_t)q : 0u;
    const uint64_t wave = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 6;
    const uint64_t nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    // Round 6: U rows in flight per wave.  A row is three DEPENDENT hops (row pointers -> indices / values -> panel rows) and a wave took them one
    // row at a time: at 11 M rows of ~14 entries the product ran at the pace of its latencies (3.1 ms; its gathers alone would take ~2).  Now the
    // hops of U rows travel together: all row pointers, then all first trips' indices, then all gathers; rows of more than 16 entries finish in
    // a tail loop of their own.  (11 M-node laplacian: 46.6 -> 44.7 ms per do_svd with U = 4; at 60 000 rows there are fewer rows than waves and
    // U = 4 costs 0.77 -> 0.84 ms: the launcher takes U = 1 below 2^20 rows.)
    for (uint64_t row0 = wave; row0 < m; row0 += (uint64_t)U * nwaves) {
        uint64_t e0[U], e1[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint64_t r = row0 + (uint64_t)u * nwaves;
            const uint64_t rc = r < m ? r : row0;
            e0[u] = indptr[rc];
            e1[u] = r < m ? indptr[rc + 1] : e0[u];   // (a row beyond the matrix: no entries)
        }
        float va[U], vc[U];
        uint64_t ca[U], cc[U];
        bool ina[U], inc[U];
#pragma unroll
        for (int u = 0; u < U; u++) {   // the first trip of every row: two edges per lane slot (clamped, masked: no branches)
            const uint64_t ea = e0[u] + sl, ec = e0[u] + 8 + sl;
            ina[u] = ea < e1[u]; inc[u] = ec < e1[u];
            va[u] = val[ina[u] ? ea : e0[u]]; vc[u] = val[inc[u] ? ec : e0[u]];
            ca[u] = ind[ina[u] ? ea : e0[u]]; cc[u] = ind[inc[u] ? ec : e0[u]];
        }
        float4 xa[U], xc[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            xa[u] = *reinterpret_cast<const float4*>(x + ca[u] * ldx + qo);
            xc[u] = *reinterpret_cast<const float4*>(x + cc[u] * ldx + qo);
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint64_t row = row0 + (uint64_t)u * nwaves;
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
            {
                const float wa = (ina[u] && active) ? va[u] : 0.f, wc = (inc[u] && active) ? vc[u] : 0.f;
                a0 = fmaf(wa, xa[u].x, a0); a1 = fmaf(wa, xa[u].y, a1); a2 = fmaf(wa, xa[u].z, a2); a3 = fmaf(wa, xa[u].w, a3);
                a0 = fmaf(wc, xc[u].x, a0); a1 = fmaf(wc, xc[u].y, a1); a2 = fmaf(wc, xc[u].z, a2); a3 = fmaf(wc, xc[u].w, a3);
            }
            for (uint64_t eb = e0[u] + 16; eb < e1[u]; eb += 16) {  // (rows of more than 16 entries: the trips in the same order as before)
                const uint64_t ea = eb + sl, ec = eb + 8 + sl;
                const bool in_a = ea < e1[u], in_c = ec < e1[u];
                const float v_a = val[in_a ? ea : e0[u]], v_c = val[in_c ? ec : e0[u]];
                const uint64_t c_a = ind[in_a ? ea : e0[u]], c_c = ind[in_c ? ec : e0[u]];
                const float4 x_a = *reinterpret_cast<const float4*>(x + c_a * ldx + qo);
                const float4 x_c = *reinterpret_cast<const float4*>(x + c_c * ldx + qo);
                const float wa = (in_a && active) ? v_a : 0.f, wc = (in_c && active) ? v_c : 0.f;
                a0 = fmaf(wa, x_a.x, a0); a1 = fmaf(wa, x_a.y, a1); a2 = fmaf(wa, x_a.z, a2); a3 = fmaf(wa, x_a.w, a3);
                a0 = fmaf(wc, x_c.x, a0); a1 = fmaf(wc, x_c.y, a1); a2 = fmaf(wc, x_c.z, a2); a3 = fmaf(wc, x_c.w, a3);
            }
#pragma unroll
            for (int off = 8; off < 64; off <<= 1) {
                a0 += __shfl_xor(a0, off); a1 += __shfl_xor(a1, off); a2 += __shfl_xor(a2, off); a3 += __shfl_xor(a3, off);
            }
            if (row < m && sl == 0 && active) *reinterpret_cast<float4*>(y + row * l + qo) = make_float4(a0, a1, a2, a3);
        }
    }
}

// MEASURED, NOT ENABLED (round 6; AE_SPMM_WIDE under AE_DEBUG_KNOBS to repeat it): on the laplacian of an 11 M-node graph the product went
// from 3.21 to 3.09 ms with the aligned copy, and chol_apply from 0.84 to 1.19 ms for writing it: 46.6 -> 47.7 ms per do_svd.  The
// product is not bound by the lines it gathers alone (a wave per row: three dependent hops -- row pointers, indices, panel rows).
static bool spmm_wide_ok(const ae_matrepr& a, uint32_t l) {
    return debug_knob("AE_SPMM_WIDE") && a.is_csr && l % 4 == 0 && l < kWideLd && (uint64_t)std::max(a.nrows, a.ncols) * kWideLd * 4ull > (256ull << 20) &&
           !debug_knob("AE_SPMM_SCALAR");
}
static void spmm(const ae_matrepr& a, const float* d_x, float* d_y, uint32_t l, uint32_t ldx = 0) {
    const unsigned grid = grid_cap(a.nrows * 64, 256);
    if (l % 4 == 0 && l <= 32 && !debug_knob("AE_SPMM_SCALAR")) {
        if (a.nrows >= (1ull << 20))
            hipLaunchKernelGGL((spmm_csr_vec4_kernel<4>), dim3(grid), dim3(256), 0, stream(), a.nrows, a.indptr.p, a.indices.p, a.values.p, d_x, d_y, l, ldx ? ldx : l);
        else
            hipLaunchKernelGGL((spmm_csr_vec4_kernel<1>), dim3(grid), dim3(256), 0, stream(), a.nrows, a.indptr.p, a.indices.p, a.values.p, d_x, d_y, l, ldx ? ldx : l);
        check_launch("spmm_csr_vec4");
        return;
    }
    if (ldx && ldx != l) fail(AE_ERR_INVALID_ARG, "internal: strided gather operand outside the vec4 product");
    if (l <= 16)
        hipLaunchKernelGGL((spmm_csr_kernel<16>), dim3(grid), dim3(256), 0, stream(), a.nrows, a.indptr.p, a.indices.p, a.values.p, d_x, d_y, l);
    else if (l <= 32)
        hipLaunchKernelGGL((spmm_csr_kernel<32>), dim3(grid), dim3(256), 0, stream(), a.nrows, a.indptr.p, a.indices.p, a.values.p, d_x, d_y, l);
    else
        hipLaunchKernelGGL((spmm_csr_kernel<64>), dim3(grid), dim3(256), 0, stream(), a.nrows, a.indptr.p, a.indices.p, a.values.p, d_x, d_y, l);
    check_launch("spmm_csr");
}

void mat_mul_panel(const ae_matrepr& a, const float* d_x, float* d_y, uint32_t l) {
    if (l == 0 || l > kMaxL) fail(AE_ERR_INVALID_ARG, "panel width %u unsupported (max %d)", l, kMaxL);
    if (a.is_csr) { spmm(a, d_x, d_y, l); return; }
    if (l <= 32 && !debug_knob("AE_NO_MFMA")) {  // matrix-core path
        constexpr int BM = 128;  // (64-row workgroups measured 5 % slower)
        const unsigned g2 = blocks_for(a.nrows, BM);
        const size_t x_bytes = (size_t)a.ncols * l * sizeof(float);
        if (a.ncols % 8 == 0 && l % 4 == 0 && x_bytes <= 64 * 1024 && !debug_knob("AE_MFMA_TILED")) {  // the panel fits in LDS: barrier-free streaming form
            const size_t lds = std::max<size_t>(x_bytes, 4 * 64 * 16 * sizeof(float));
            // (U = 2 / 3 / 4 / 6 / 8 loads per trip: 37.4 / 37.4 / 36.2 / 36.4 / 36.7 us; the contraction split over two waves per row group,
            // 16 waves per CU: 39 us)
            if (debug_knob("AE_SVD_KSPLIT")) {
                hipLaunchKernelGGL((dense_mul_panel_mfma_stream_kernel<4, 2>), dim3(g2), dim3(512), lds, stream(), a.values.p, a.nrows, a.ncols, d_x, d_y, l);
            } else {
                // (a workgroup walks over row blocks once there are more blocks than the device holds workgroups: the panel is staged once)
                int per_cu = 2;
                (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, dense_mul_panel_mfma_stream_kernel<4, 1>, 256, lds);
                const unsigned cap = debug_knob("AE_SVD_STREAM_GRID") ? (unsigned)atoi(debug_knob("AE_SVD_STREAM_GRID")) : 256u * (unsigned)std::max(1, per_cu);
                hipLaunchKernelGGL((dense_mul_panel_mfma_stream_kernel<4, 1>), dim3(std::min(g2, std::max(1u, cap))), dim3(256), lds, stream(), a.values.p, a.nrows, a.ncols,
                                   d_x, d_y, l);
            }
            check_launch("dense_mul_panel_mfma_stream");
            return;
        }
        if (a.ncols % 4 == 0)
            hipLaunchKernelGGL((dense_mul_panel_mfma_kernel<true, BM>), dim3(g2), dim3(BM * 2), 0, stream(), a.values.p, a.nrows, a.ncols, d_x, d_y, l);
        else
            hipLaunchKernelGGL((dense_mul_panel_mfma_kernel<false, BM>), dim3(g2), dim3(BM * 2), 0, stream(), a.values.p, a.nrows, a.ncols, d_x, d_y, l);
        check_launch("dense_mul_panel_mfma");
        return;
    }
    const unsigned grid = grid_cap(a.nrows * 64, 256);
    if (l <= 32)
        hipLaunchKernelGGL((dense_mul_panel_kernel<32>), dim3(grid), dim3(256), 0, stream(), a.values.p, a.nrows, a.ncols, d_x, d_y, l);
    else
        hipLaunchKernelGGL((dense_mul_panel_kernel<64>), dim3(grid), dim3(256), 0, stream(), a.values.p, a.nrows, a.ncols, d_x, d_y, l);
    check_launch("dense_mul_panel");
}

void build_transpose(ae_matrepr& a) {
    if (a.transpose || !a.is_csr) return;
    std::unique_ptr<ae_matrepr> t(new ae_matrepr);
    t->is_csr = true;
    t->nrows = a.ncols; t->ncols = a.nrows; t->nnz = a.nnz;
    t->indptr.alloc(t->nrows + 1);
    t->indices.alloc(a.nnz ? a.nnz : 1);
    t->values.alloc(a.nnz ? a.nnz : 1);
    if (a.nnz) {
        DevBuf<uint64_t> k0(a.nnz), k1(a.nnz);
        DevBuf<uint32_t> p0(a.nnz), p1(a.nnz);
        hipLaunchKernelGGL(coo_keys_kernel, dim3(blocks_for(a.nrows, 256)), dim3(256), 0, stream(), a.nrows, a.indptr.p, a.indices.p, k0.p, p0.p);
        check_launch("coo_keys");
        sort_pairs_u64_u32(k0.p, k1.p, p0.p, p1.p, a.nnz);
        hipLaunchKernelGGL(transpose_fill_kernel, dim3(blocks_for(a.nnz, 256)), dim3(256), 0, stream(), a.nnz, k1.p, p1.p, a.values.p,
                           t->indices.p, t->values.p);
        check_launch("transpose_fill");
        rowptr_from_sorted_keys(k1.p, a.nnz, t->nrows, t->indptr.p);
        sync();
    } else {
        t->indptr.zero();
    }
    a.transpose = std::move(t);
}

void mat_t_mul_panel(ae_matrepr& a, const float* d_x, float* d_y, uint32_t l) {
    if (l == 0 || l > kMaxL) fail(AE_ERR_INVALID_ARG, "panel width %u unsupported (max %d)", l, kMaxL);
    if (a.is_csr) {
        if (a.symmetric) { spmm(a, d_x, d_y, l); return; }
        build_transpose(a);
        spmm(*a.transpose, d_x, d_y, l);
        return;
    }
    const uint64_t m = a.nrows, n = a.ncols;
    if (l <= 32 && !debug_knob("AE_NO_MFMA")) {  // matrix-core path: 32-column tiles x row chunks, then a deterministic reduce
        const uint64_t ctiles = (n + 127) / 128;  // strips of 128 columns
        const bool pp = n % 4 == 0 && !debug_knob("AE_SVD_T_OLD");
        // the two-buffer form wants long chunks: about two workgroups per CU (256 CUs) in all (256 / 512 / 768 / 1024 workgroups: 46.7 / 46.7 /
        // 49.8 / 54.6 us per product at 60000 x 784); the older form many short ones
        uint64_t chunks = std::max<uint64_t>(1, std::min<uint64_t>((m + 511) / 512, std::max<uint64_t>(1, (pp ? 512 : 2048) / ctiles)));
        const uint64_t rpc = ((m + chunks - 1) / chunks + 7) & ~7ull;
        chunks = (m + rpc - 1) / rpc;
        static DevBuf<float> part;
        if (part.n < chunks * n * l) { sync(); part.alloc(chunks * n * l); }
        if (pp)
            hipLaunchKernelGGL(dense_t_mul_panel_mfma_pp_kernel, dim3((unsigned)ctiles, (unsigned)chunks), dim3(256), 0, stream(), a.values.p, m,
                               n, d_x, part.p, l, rpc);
        else if (n % 4 == 0)
            hipLaunchKernelGGL((dense_t_mul_panel_mfma_kernel<true>), dim3((unsigned)ctiles, (unsigned)chunks), dim3(256), 0, stream(), a.values.p, m,
                               n, d_x, part.p, l, rpc);
        else
            hipLaunchKernelGGL((dense_t_mul_panel_mfma_kernel<false>), dim3((unsigned)ctiles, (unsigned)chunks), dim3(256), 0, stream(), a.values.p, m,
                               n, d_x, part.p, l, rpc);
        check_launch("dense_t_mul_panel_mfma");
        hipLaunchKernelGGL(reduce_chunks_kernel, dim3(blocks_for(n * l, 256)), dim3(256), 0, stream(), part.p, chunks, n * l, d_y);
        check_launch("reduce_chunks");
        return;
    }
    const uint64_t colblocks = (n + 63) / 64;
    uint64_t chunks = std::max<uint64_t>(1, std::min<uint64_t>((m + 255) / 256, std::max<uint64_t>(1, 4096 / colblocks)));
    const uint64_t rpc = (m + chunks - 1) / chunks;
    chunks = (m + rpc - 1) / rpc;
    DevBuf<float> partial(chunks * n * l);
    if (l <= 32)
        hipLaunchKernelGGL((dense_t_mul_panel_kernel<32>), dim3((unsigned)colblocks, (unsigned)chunks), dim3(64), 0, stream(), a.values.p, m, n, d_x, partial.p, l, rpc);
    else
        hipLaunchKernelGGL((dense_t_mul_panel_kernel<64>), dim3((unsigned)colblocks, (unsigned)chunks), dim3(64), 0, stream(), a.values.p, m, n, d_x, partial.p, l, rpc);
    check_launch("dense_t_mul_panel");
    hipLaunchKernelGGL(reduce_chunks_kernel, dim3(blocks_for(n * l, 256)), dim3(256), 0, stream(), partial.p, chunks, n * l, d_y);
    check_launch("reduce_chunks");
    sync();
}

// ---- f64 Gram on the matrix cores: G (l x l, f64) += Y^T Y with v_mfma_f64_16x16x4_f64.  For G = Y^T Y both
// operands of a 16x16 tile are the same register: lane (k = lane / 16, c = lane % 16) holds Y[row0 + k][16 a + c]
// for column tile a (4 rows per step).  One wave accumulates TB x TB tiles over its rows, the four waves of a block
// meet in LDS, one f64 atomic per entry and block lands in G.  Products of f32 values are exact in f64, so one
// Cholesky pass on this Gram orthonormalises the panel to f32 rounding (CholeskyQR with an f64 Gram).
typedef double d4_t __attribute__((ext_vector_type(4)));
template <int TB>
__global__ void __launch_bounds__(256) gram_mfma_f64_kernel(const float* __restrict__ y, uint64_t rows, uint32_t l, double* __restrict__ g) {
    __shared__ double sg[TB * 16 * TB * 16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, kq = lane >> 4;
    d4_t acc[TB][TB];
#pragma unroll
    for (int a = 0; a < TB; a++)
#pragma unroll
        for (int b = 0; b < TB; b++) acc[a][b] = d4_t{0., 0., 0., 0.};
    const uint64_t nsteps = (rows + 3) / 4;
    const uint64_t sstride = gridDim.x * 4ull;
    // steps whose loads are issued together (clamped addresses, masked values: no branches).  Round 6: 8 for panels of <= 32 columns and
    // a grid of up to 2048 workgroups -- at 11 M rows the kernel ran at 1.6 TB/s with 4 MB of loads in flight (512 workgroups x 4 waves
    // x 2 KB); HBM wants ~16 MB
    constexpr int U = TB <= 2 ? 8 : 4;
    for (uint64_t s0 = blockIdx.x * 4ull + (uint64_t)__builtin_amdgcn_readfirstlane(wave); s0 < nsteps; s0 += U * sstride) {
        float raw[U][TB];
#pragma unroll
        for (int uu = 0; uu < U; uu++) {
            const uint64_t row = (s0 + uu * sstride) * 4 + kq;
            const uint64_t rc = row < rows ? row : 0;
#pragma unroll
            for (int a = 0; a < TB; a++) {
                const uint32_t colx = 16u * a + c;
                raw[uu][a] = y[rc * l + (colx < l ? colx : 0u)];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int uu = 0; uu < U; uu++) {
            const uint64_t row = (s0 + uu * sstride) * 4 + kq;
            double v[TB];
#pragma unroll
            for (int a = 0; a < TB; a++) v[a] = (row < rows && 16u * a + c < l) ? (double)raw[uu][a] : 0.;
#pragma unroll
            for (int a = 0; a < TB; a++)
#pragma unroll
                for (int b = 0; b < TB; b++) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(v[a], v[b], acc[a][b], 0, 0, 0);
        }
    }
    constexpr int W = TB * 16;
    for (int idx = threadIdx.x; idx < W * W; idx += 256) sg[idx] = 0.;
    __syncthreads();
    for (int w = 0; w < 4; w++) {
        if (wave == w) {
#pragma unroll
            for (int a = 0; a < TB; a++)
#pragma unroll
                for (int b = 0; b < TB; b++)
#pragma unroll
                    for (int q = 0; q < 4; q++) sg[(16 * a + kq + 4 * q) * W + 16 * b + c] += acc[a][b][q];  // D[i = lane / 16 + 4 q][j = lane % 16] (tools/test_mfma_f64.hip)
        }
        __syncthreads();
    }
    for (uint32_t idx = threadIdx.x; idx < l * l; idx += 256) atomicAdd(&g[idx], sg[(idx / l) * W + idx % l]);
}

static void launch_gram_mfma(const float* d_y, uint64_t rows, uint32_t l, double* d_g) {
    const uint64_t nsteps = (rows + 3) / 4;
    const unsigned grid = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((nsteps + 63) / 64, debug_knob("AE_GRAM_GRID") ? atoi(debug_knob("AE_GRAM_GRID")) : 2048));
    const int tb = (int)((l + 15) / 16);
    switch (tb) {
        case 1: hipLaunchKernelGGL((gram_mfma_f64_kernel<1>), dim3(grid), dim3(256), 0, stream(), d_y, rows, l, d_g); break;
        case 2: hipLaunchKernelGGL((gram_mfma_f64_kernel<2>), dim3(grid), dim3(256), 0, stream(), d_y, rows, l, d_g); break;
        case 3: hipLaunchKernelGGL((gram_mfma_f64_kernel<3>), dim3(grid), dim3(256), 0, stream(), d_y, rows, l, d_g); break;
        default: hipLaunchKernelGGL((gram_mfma_f64_kernel<4>), dim3(grid), dim3(256), 0, stream(), d_y, rows, l, d_g); break;
    }
    check_launch("gram_mfma_f64");
}

// Cholesky of the Gram, R^-1, and the panel update Y <- Y R^-1 in ONE launch: every workgroup redoes the tiny
// factorisation in LDS (a few microseconds, in parallel) instead of waiting for a single-workgroup kernel.  A pivot
// that is not safely positive (rank-deficient / ill-conditioned panel) sets the sticky flag and leaves Y alone: the
// caller then repeats its computation through the eigen (SVQB) route.  Workgroup 0 also clears `g_zero`, the
// accumulator of the next Gram.
__device__ __forceinline__ double readlane_f64(double v, int srclane) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, srclane);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), srclane);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
// Upper Cholesky G = R^T R of an l x l matrix (l <= LT <= 64) by ONE wave with the matrix in registers: lane i owns
// column i (col[k] = R[k][i]); step j needs row j of the factor so far, i.e. lane j's registers, which the scalar
// broadcast v_readlane supplies at a few cycles each -- ~5 us for l = 20 against ~20 us through LDS round trips.
// Returns false when a pivot is not safely positive.
template <int LT>
__device__ __forceinline__ bool chol_wave_registers(const double* R_in /*LDS, l x l*/, uint32_t l, double rel_tol, int lane, double* col,
                                                    double* rinv_out /*LDS, l*/) {
    const uint32_t ic = (uint32_t)lane < l ? (uint32_t)lane : l - 1;
    double dmax = 0.;
    for (uint32_t q = 0; q < l; q++) dmax = R_in[q * l + q] > dmax ? R_in[q * l + q] : dmax;
#pragma unroll
    for (int k = 0; k < LT; k++) col[k] = (uint32_t)k < l ? R_in[k * l + ic] : 0.;  // column of G (lanes >= l: a copy, unused)
    bool ok = true;
#pragma unroll
    for (int j = 0; j < LT; j++) {
        if ((uint32_t)j < l && ok) {  // wave-uniform
            double s0 = 0., s1 = 0.;
#pragma unroll
            for (int k = 0; k < j; k++) {
                const double rkj = readlane_f64(col[k], j);
                if (k & 1) s1 += rkj * col[k]; else s0 += rkj * col[k];
            }
            const double v = col[j] - (s0 + s1);
            const double dd = readlane_f64(v, j);
            if (!(dd > rel_tol * dmax)) ok = false;
            else {
                const double inv = rsqrt(dd);
                col[j] = (lane == j) ? dd * inv : ((lane > j) ? v * inv : 0.);
                if (lane == j) rinv_out[j] = inv;
            }
        }
    }
    return ok;
}

// y_wide (may be null): the orthonormalised rows once more, kWideLd floats apart -- the gather copy the next sparse product reads (spmm)
__global__ void __launch_bounds__(256) chol_apply_kernel(float* __restrict__ y, uint64_t rows, uint32_t l, const double* __restrict__ g,
                                                         double rel_tol, int* __restrict__ flag, double* __restrict__ g_zero, uint32_t rp,
                                                         float* __restrict__ y_wide = nullptr) {
    extern __shared__ double smem[];         // R[l*l] | rinv[l] | tile[rp rows * (l + 1)] (f64, one row per thread, rp <= 256)
    double* R = smem;
    double* rinv = smem + (size_t)l * l;
    double* tile = rinv + l;
    __shared__ int s_bad;
    const uint32_t tid = threadIdx.x;
    for (uint32_t idx = tid; idx < l * l; idx += 256) R[idx] = g[idx];
    if (blockIdx.x == 0 && g_zero)
        for (uint32_t idx = tid; idx < l * l; idx += 256) g_zero[idx] = 0.;
    if (tid == 0) s_bad = 0;
    __syncthreads();
    if (tid < 64 && l <= 32) {  // wave 0 factorises in registers and writes the factor back for the row solves
        double col[32];
        const bool okc = chol_wave_registers<32>(R, l, rel_tol, (int)tid, col, rinv);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        if (!okc) { if (tid == 0) s_bad = 1; }
        else if (tid < l) {
#pragma unroll
            for (int k = 0; k < 32; k++)
                if ((uint32_t)k < l) R[k * l + tid] = col[k];
        }
    } else if (tid < 64) {  // wider panels: wave 0 factorises through LDS (lane i owns column i)
        const uint32_t i = tid;
        const uint32_t ic = i < l ? i : l - 1;
        double dmax = 0.;
        for (uint32_t q = 0; q < l; q++) dmax = R[q * l + q] > dmax ? R[q * l + q] : dmax;
        bool bad = false;
        for (uint32_t j = 0; j < l; j++) {  // upper Cholesky G = R^T R, row j per step
            double v0 = 0., v1 = 0., v2 = 0., v3 = 0.;
            uint32_t k = 0;
            for (; k + 4 <= j; k += 4) {
                v0 += R[k * l + j] * R[k * l + ic];
                v1 += R[(k + 1) * l + j] * R[(k + 1) * l + ic];
                v2 += R[(k + 2) * l + j] * R[(k + 2) * l + ic];
                v3 += R[(k + 3) * l + j] * R[(k + 3) * l + ic];
            }
            for (; k < j; k++) v0 += R[k * l + j] * R[k * l + ic];
            const double v = R[j * l + ic] - ((v0 + v1) + (v2 + v3));
            const double dd = __shfl(v, (int)j);
            if (!(dd > rel_tol * dmax)) { bad = true; break; }  // wave-uniform
            const double inv = rsqrt(dd);
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            if (i >= j && i < l) R[j * l + i] = (i == j) ? dd * inv : v * inv;
            if (i == j) rinv[j] = inv;
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        }
        if (bad && i == 0) s_bad = 1;
    }
    __syncthreads();
    if (s_bad) {
        if (blockIdx.x == 0 && tid == 0) atomicOr(flag, 1);
        return;
    }
    // Y <- Y R^-1 row by row: x R = y by forward substitution, one thread per row (no explicit inverse)
    const uint64_t ntiles = (rows + rp - 1) / rp;
    for (uint64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const uint64_t r0 = t * rp;
        const uint32_t nr = (uint32_t)(rows - r0 < rp ? rows - r0 : rp);
        __syncthreads();
        if (l <= 32) {  // rp * l / 256 <= 32 elements per thread: all loads first (clamped), then the LDS stores
            float tmp[32];
            const uint32_t tot = nr * l;
#pragma unroll
            for (int it = 0; it < 32; it++) {
                const uint32_t idx = tid + (uint32_t)it * 256u;
                tmp[it] = y[r0 * l + (idx < tot ? idx : 0u)];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int it = 0; it < 32; it++) {
                const uint32_t idx = tid + (uint32_t)it * 256u;
                if (idx < tot) tile[(idx / l) * (l + 1) + idx % l] = (double)tmp[it];
            }
        } else {
            for (uint32_t idx = tid; idx < nr * l; idx += 256) tile[(idx / l) * (l + 1) + idx % l] = (double)y[r0 * l + idx];  // coalesced load
        }
        __syncthreads();
        if (tid < nr) {
            double* x = tile + (size_t)tid * (l + 1);  // odd stride in doubles: conflict-free rows
            if (l <= 32) {  // the row in registers, the factor by LDS broadcast reads (every thread reads the same R[k][c])
                double xr[32];
#pragma unroll
                for (int c = 0; c < 32; c++) xr[c] = (uint32_t)c < l ? x[c] : 0.;
#pragma unroll
                for (int c = 0; c < 32; c++) {
                    if ((uint32_t)c < l) {  // uniform
                        double a0 = xr[c], a1 = 0.;
#pragma unroll
                        for (int k = 0; k < c; k++) {
                            if (k & 1) a1 -= xr[k] * R[k * l + c]; else a0 -= xr[k] * R[k * l + c];
                        }
                        xr[c] = (a0 + a1) * rinv[c];
                    }
                }
#pragma unroll
                for (int c = 0; c < 32; c++)
                    if ((uint32_t)c < l) x[c] = xr[c];
            } else {
                for (uint32_t c = 0; c < l; c++) {
                    double acc = x[c];
                    for (uint32_t k = 0; k < c; k++) acc -= x[k] * R[k * l + c];
                    x[c] = acc * rinv[c];
                }
            }
        }
        __syncthreads();
        for (uint32_t idx = tid; idx < nr * l; idx += 256) {
            const float v = (float)tile[(idx / l) * (l + 1) + idx % l];
            y[r0 * l + idx] = v;
            if (y_wide) y_wide[(r0 + idx / l) * kWideLd + idx % l] = v;
        }
    }
}

// x <- x R^-1 for one row held in registers (R upper triangular in LDS, read by broadcast; rinv = 1 / diag): forward substitution
template <int LT>
__device__ __forceinline__ void row_solve_upper(double (&xr)[LT], const double* R, const double* rinv, uint32_t l) {
#pragma unroll
    for (int c = 0; c < LT; c++) {
        if ((uint32_t)c < l) {  // uniform
            double a0 = xr[c], a1 = 0.;
#pragma unroll
            for (int k = 0; k < c; k++) {
                if (k & 1) a1 -= xr[k] * R[k * l + c]; else a0 -= xr[k] * R[k * l + c];
            }
            xr[c] = (a0 + a1) * rinv[c];
        }
    }
}
// The SMALL side of a tall dense matrix's range iteration in ONE single-workgroup launch (n x l floats in LDS, l <= 32).  The tall panel
// Y (m x l) is never orthonormalised explicitly: its Q = Y R^-1 (R^T R = g_tall, the f64 Gram of Y) enters the next product only through
// Z = A^T Q = (A^T Y) R^-1 -- an n x l matrix.  So this kernel (1) factorises g_tall, (2) Z <- Z R^-1, and unless `apply_only` (3) QR of Z
// itself: Gram in f64, Cholesky, Z <- Z R2^-1 -- what used to be a chol_apply over the m rows of Y (read + write of the tall panel), a
// Gram launch and a chol_apply launch on Z.  g_zero: the Gram accumulator the NEXT tall panel adds into.  A pivot that is not safely
// positive raises the sticky flag and leaves Z as it is (the caller redoes the iteration through the explicit route).
__global__ void __launch_bounds__(512) small_panel_qr_kernel(float* __restrict__ z, uint32_t n, uint32_t l, const double* __restrict__ g_tall, double rel_tol,
                                                              int* __restrict__ flag, double* __restrict__ g_zero, int apply_only) {
    extern __shared__ double smem[];   // R[l * l] | rinv[l] | G2[l * l] | Zs[n * ls] (f32, ls = l | 1: odd row stride, conflict-free rows)
    double* R = smem;
    double* rinv = smem + (size_t)l * l;
    double* G2 = rinv + l;
    float* Zs = reinterpret_cast<float*>(G2 + (size_t)l * l);
    __shared__ int s_bad;
    const uint32_t tid = threadIdx.x, ls = l | 1u;
    if (tid == 0) s_bad = 0;
    for (uint32_t idx = tid; idx < l * l; idx += 512) { R[idx] = g_tall[idx]; if (g_zero) g_zero[idx] = 0.; }
    for (uint32_t idx = tid; idx < n * l; idx += 512) Zs[(idx / l) * ls + idx % l] = z[idx];
    __syncthreads();
    auto factorise = [&](double* M) {   // wave 0: M (l x l, LDS) -> its upper Cholesky factor in place, rinv
        if (tid < 64) {
            double col[32];
            const bool okc = chol_wave_registers<32>(M, l, rel_tol, (int)tid, col, rinv);
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            if (!okc) { if (tid == 0) s_bad = 1; }
            else if (tid < l) {
#pragma unroll
                for (int k = 0; k < 32; k++)
                    if ((uint32_t)k < l) M[k * l + tid] = col[k];
            }
        }
        __syncthreads();
    };
    auto solve_rows = [&](const double* M) {   // Zs <- Zs M^-1, one thread per row, the row in f64 registers
        for (uint32_t r = tid; r < n; r += 512) {
            double xr[32];
            float* x = Zs + (size_t)r * ls;
#pragma unroll
            for (int c = 0; c < 32; c++) xr[c] = (uint32_t)c < l ? (double)x[c] : 0.;
            row_solve_upper<32>(xr, M, rinv, l);
#pragma unroll
            for (int c = 0; c < 32; c++)
                if ((uint32_t)c < l) x[c] = (float)xr[c];
        }
        __syncthreads();
    };
    factorise(R);
    if (s_bad) { if (tid == 0) atomicOr(flag, 1); return; }
    solve_rows(R);
    if (!apply_only) {
        // Gram of the (f32-rounded) rows in f64: entry (a, b) by one thread, two accumulators
        for (uint32_t e = tid; e < l * l; e += 512) {
            const uint32_t a = e / l, b = e % l;
            double s0 = 0., s1 = 0.;
            uint32_t r = 0;
            for (; r + 2 <= n; r += 2) {
                s0 += (double)Zs[(size_t)r * ls + a] * (double)Zs[(size_t)r * ls + b];
                s1 += (double)Zs[(size_t)(r + 1) * ls + a] * (double)Zs[(size_t)(r + 1) * ls + b];
            }
            if (r < n) s0 += (double)Zs[(size_t)r * ls + a] * (double)Zs[(size_t)r * ls + b];
            G2[e] = s0 + s1;
        }
        __syncthreads();
        factorise(G2);
        if (s_bad) { if (tid == 0) atomicOr(flag, 1); return; }
        solve_rows(G2);
    }
    for (uint32_t idx = tid; idx < n * l; idx += 512) z[idx] = Zs[(idx / l) * ls + idx % l];
}

// optimistic orthonormalisation state: two Gram accumulators used alternately (the update kernel of call k clears
// the accumulator of call k + 1) and the sticky failure flag
struct FastOrth {
    DevBuf<double> g[2];
    DevBuf<int> flag;
    unsigned k = 0;
    void init() {
        if (!flag.n) {
            g[0].alloc(kMaxL * kMaxL); g[1].alloc(kMaxL * kMaxL); flag.alloc(1);
            g[0].zero(); g[1].zero(); flag.zero();
        }
    }
};
static FastOrth& fast_orth() { static FastOrth f; f.init(); return f; }

static void orthonormalize_panel_fast_wide(float* d_y, uint64_t rows, uint32_t l, float* d_y_wide);
void orthonormalize_panel_fast(float* d_y, uint64_t rows, uint32_t l) { orthonormalize_panel_fast_wide(d_y, rows, l, nullptr); }
static void orthonormalize_panel_fast_wide(float* d_y, uint64_t rows, uint32_t l, float* d_y_wide) {
    FastOrth& f = fast_orth();
    double* g = f.g[f.k & 1].p;
    double* gz = f.g[(f.k + 1) & 1].p;
    f.k++;
    launch_gram_mfma(d_y, rows, l, g);
    // every workgroup pays the factorisation once; rp rows per workgroup and pass, sized to stay under 64 KB of LDS
    const uint32_t rp = l <= 24 ? 256u : (l <= 40 ? 128u : 32u);
    const uint64_t ntiles = (rows + rp - 1) / rp;
    const unsigned nblocks = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(ntiles, 1024));
    const size_t smem = sizeof(double) * ((size_t)l * l + l + rp * (size_t)(l + 1));
    hipLaunchKernelGGL(chol_apply_kernel, dim3(nblocks), dim3(256), smem, stream(), d_y, rows, l, (const double*)g, 1e-10, f.flag.p, gz, rp, d_y_wide);
    check_launch("chol_apply");
}
// true when a Cholesky pivot failed since the last call (synchronises the stream)
bool orthonormalize_fast_failed() {
    FastOrth& f = fast_orth();
    int h = 0;
    f.flag.download(&h, 1);
    if (h) { f.flag.zero(); f.g[0].zero(); f.g[1].zero(); sync(); }
    return h != 0;
}

void gram_panel(const float* d_y, uint64_t rows, uint32_t l, double* d_g) {
    AE_HIP(hipMemsetAsync(d_g, 0, sizeof(double) * l * l, stream()));
    launch_gram_mfma(d_y, rows, l, d_g);
}
void apply_panel(const float* d_y, uint64_t rows, uint32_t l, const double* d_m, uint32_t lout, float* d_out) {
    if (d_out == d_y && lout > l) fail(AE_ERR_INVALID_ARG, "in-place apply needs lout <= l");
    const uint64_t ntiles = (rows + kGramTile - 1) / kGramTile;
    const unsigned nblocks = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(ntiles, 8192));
    hipLaunchKernelGGL(apply_panel_kernel, dim3(nblocks), dim3(256), 0, stream(), d_y, rows, l, d_m, lout, d_out);
    check_launch("apply_panel");
}

void jacobi_eigh_device(const double* d_g, uint32_t l, double* d_evals, double* d_evecs) {
    hipLaunchKernelGGL(jacobi_eigh_kernel, dim3(1), dim3(256), 0, stream(), d_g, l, d_evals, d_evecs, (const int*)nullptr);
    check_launch("jacobi_eigh");
}

void orthonormalize_panel(float* d_y, uint64_t rows, uint32_t l, double* d_work) {
    double* g = d_work;                       // Gram
    double* mt = d_work + (uint64_t)l * l;    // transform M (Cholesky route: R^-1; eigen route: V diag(lambda^-1/2))
    double* ev = d_work + 2ull * l * l;       // eigenvectors (eigen route)
    static DevBuf<double> evals_buf;          // caller's work area is 3*l*l + l doubles: eigenvalues live here
    if (evals_buf.n < (size_t)kMaxL) evals_buf.alloc(kMaxL);
    static DevBuf<int> flag;
    if (!flag.n) flag.alloc(1);
    for (int pass = 0; pass < 2; pass++) {
        gram_panel(d_y, rows, l, g);
        // fast route: Cholesky QR (the panels of the power iteration are well conditioned).  When a pivot is not
        // safely positive the device flag routes the SAME launch sequence through the eigen (SVQB) kernels, which
        // turn rank-deficient directions into zero columns; no host round trip either way.
        hipLaunchKernelGGL(chol_inverse_kernel, dim3(1), dim3(64), 0, stream(), g, l, 1e-10, mt, flag.p);
        check_launch("chol_inverse");
        hipLaunchKernelGGL(jacobi_eigh_kernel, dim3(1), dim3(256), 0, stream(), g, l, evals_buf.p, ev, (const int*)flag.p);
        check_launch("jacobi_eigh");
        hipLaunchKernelGGL(svqb_scale_kernel, dim3(blocks_for(l * l, 256)), dim3(256), 0, stream(), evals_buf.p, ev, l,
                           pass == 0 ? 1e-12 : 1e-6, mt, (const int*)flag.p);
        check_launch("svqb_scale");
        apply_panel(d_y, rows, l, mt, l, d_y);
    }
}

// A tall dense matrix (m >= n, the n x l panel fits one workgroup's LDS): the tall panel is kept UN-normalised with the f64 Gram that
// defines its QR -- Q = Y R^-1, R^T R = g -- and the R^-1 is applied on the small side (small_panel_qr_kernel).  Per iteration: product,
// Gram, transposed product (+ its reduction), one small launch -- instead of product, Gram, chol_apply over m rows, transposed
// product, reduction, Gram, chol_apply; the tall panel is written once and read twice per iteration instead of written twice and read
// four times.
struct TallFactor {
    bool deferred = false;   // q holds Y, not Q
    const double* g = nullptr;
    DevBuf<float> q_wide;    // (sparse matrices with panels beyond the caches) Q once more, rows 128 bytes apart: what the next sparse product gathers
};
static size_t small_panel_lds(uint64_t n, uint32_t l) { return sizeof(double) * (2 * (size_t)l * l + l) + sizeof(float) * (size_t)n * (l | 1u); }
// (from 4 M panel entries on: below, the chol_apply over the tall panel it saves -- 23 us at 60 000 x 20 -- costs less than the
// single-workgroup launch that replaces it -- measured at 60 000 x 784: 1.12 ms per direct_svd explicit, 1.27 ms deferred; at
// 6.25 M x 128: 12.5 -> 10.4 ms)
static bool dense_deferred_ok(const ae_matrepr& a, uint32_t l) {
    return !a.is_csr && a.nrows >= a.ncols && (uint64_t)a.nrows * l >= 4000000ull && l <= 32 && small_panel_lds(a.ncols, l) <= 150 * 1024 &&
           !debug_knob("AE_SVD_NO_DEFER");
}
static void launch_small_panel_qr(float* z, uint64_t n, uint32_t l, const double* g_tall, int* flag, double* g_zero, bool apply_only) {
    const size_t lds = small_panel_lds(n, l);
    static bool attr_set = false;
    if (!attr_set) {   // (more than the 64 KB a launch may ask for by default: 784 x 20 floats + the factors = 71 KB)
        AE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(small_panel_qr_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        attr_set = true;
    }
    hipLaunchKernelGGL(small_panel_qr_kernel, dim3(1), dim3(512), lds, stream(), z, (uint32_t)n, l, g_tall, 1e-10, flag, g_zero, apply_only ? 1 : 0);
    check_launch("small_panel_qr");
}

// subspace_iteration_full / _csr, svdapprox.rs:285-408.  d_q: m x l panel (output, orthonormal columns -- or, with `tf` and a tall dense
// matrix, the un-normalised panel and its Gram: TallFactor)
static uint32_t subspace_iteration_device(ae_matrepr& a, uint64_t rank, uint64_t nbiter, DevBuf<float>& q, TallFactor* tf = nullptr, bool no_defer = false) {
    const uint64_t m = a.nrows, n = a.ncols;
    const uint32_t l = (uint32_t)std::min<uint64_t>(std::min(m, n), rank);  // :294 / :358
    if (l == 0 || l > kMaxL) fail(AE_ERR_INVALID_ARG, "rank %llu unsupported (1..%d)", (unsigned long long)rank, kMaxL);
    DevBuf<float> omega, yn;
    DevBuf<double> work;
    omega.alloc_pooled(n * l);
    yn.alloc_pooled(n * l);
    work.alloc_pooled(3ull * l * l + l);
    q.alloc_pooled(m * l);
    if (tf) { tf->deferred = false; tf->g = nullptr; tf->q_wide.release(); }
    if (!no_defer && dense_deferred_ok(a, l)) {
        FastOrth& f = fast_orth();
        f.g[0].zero(); f.g[1].zero();
        unsigned k = 0;
        gaussian_fill_device(omega.p, n * l, kDefaultSeed, kTagOmega);  // :69-76, :299
        mat_mul_panel(a, omega.p, q.p, l);                              // :300
        launch_gram_mfma(q.p, m, l, f.g[k].p);                          // the Gram IS the QR of :307 (R^T R = g)
        for (uint64_t j = 1; j < nbiter; j++) {                         // :308
            mat_t_mul_panel(a, q.p, yn.p, l);                           // :311, on Y: (A^T Y) ...
            launch_small_panel_qr(yn.p, n, l, f.g[k].p, f.flag.p, f.g[k ^ 1].p, false);   // ... R^-1, then the QR of :313-319
            mat_mul_panel(a, yn.p, q.p, l);                             // :321
            k ^= 1;
            launch_gram_mfma(q.p, m, l, f.g[k].p);                      // :323-329
        }
        if (!orthonormalize_fast_failed()) {   // (synchronises)
            if (tf) { tf->deferred = true; tf->g = f.g[k].p; return l; }
            // the caller wants Q itself: one explicit Y <- Y R^-1
            const uint32_t rp = l <= 24 ? 256u : (l <= 40 ? 128u : 32u);
            const uint64_t ntiles = (m + rp - 1) / rp;
            const size_t smem = sizeof(double) * ((size_t)l * l + l + rp * (size_t)(l + 1));
            hipLaunchKernelGGL(chol_apply_kernel, dim3((unsigned)std::max<uint64_t>(1, std::min<uint64_t>(ntiles, 1024))), dim3(256), smem, stream(), q.p, m, l,
                               (const double*)f.g[k].p, 1e-10, f.flag.p, (double*)nullptr, rp);
            check_launch("chol_apply");
            if (!orthonormalize_fast_failed()) { f.g[0].zero(); f.g[1].zero(); sync(); return l; }   // (the accumulators are left zeroed: orthonormalize_panel_fast counts on it)
        }
        // a rank-deficient panel somewhere: the explicit route below (its own fallback turns dependent directions into zero columns)
    }
    // do_qr (Householder, :998-1013) is replaced by CholeskyQR on an f64 Gram.  First optimistically (two launches per
    // QR, no host round trip); if any panel was rank deficient (sticky device flag) the whole iteration is redone
    // with the eigen route, which turns the dependent directions into zero columns.
    // Sparse matrices whose panels do not fit the caches: every orthonormalised panel is also written with its rows 128 bytes apart, and
    // the next product gathers from that copy (one line per gathered row instead of 1.6: spmm_csr_vec4_kernel)
    const bool wide = spmm_wide_ok(a, l);
    DevBuf<float> q_wide, yn_wide;
    if (wide) { q_wide.alloc_pooled(m * kWideLd); yn_wide.alloc_pooled(n * kWideLd); }
    for (int robust = 0; robust < 2; robust++) {
        const bool w = wide && !robust;
        auto qr = [&](float* y, uint64_t rows, float* yw) {
            if (robust) orthonormalize_panel(y, rows, l, work.p);
            else orthonormalize_panel_fast_wide(y, rows, l, w ? yw : nullptr);
        };
        auto mul = [&](const float* x, const float* xw, float* y) {       // y = A x
            if (w) spmm(a, xw, y, l, kWideLd); else mat_mul_panel(a, x, y, l);
        };
        auto tmul = [&](const float* x, const float* xw, float* y) {      // y = A^T x
            if (!w) { mat_t_mul_panel(a, x, y, l); return; }
            if (a.symmetric) { spmm(a, xw, y, l, kWideLd); return; }
            build_transpose(a);
            spmm(*a.transpose, xw, y, l, kWideLd);
        };
        gaussian_fill_device(omega.p, n * l, kDefaultSeed, kTagOmega);  // RandomGaussianMatrix::new, :69-76, :299/:363
        mat_mul_panel(a, omega.p, q.p, l);                              // :300 / :366
        qr(q.p, m, q_wide.p);                                           // :307 / :374
        for (uint64_t j = 1; j < nbiter; j++) {                         // :308 / :375
            tmul(q.p, q_wide.p, yn.p);                                  // :311 / :379
            qr(yn.p, n, yn_wide.p);                                     // :313-319 / :381-387
            mul(yn.p, yn_wide.p, q.p);                                  // :321 / :390
            qr(q.p, m, q_wide.p);                                       // :323-329 / :392-398
        }
        if (robust || !orthonormalize_fast_failed()) {
            if (w && tf) tf->q_wide = std::move(q_wide);
            break;
        }
    }
    sync();
    return l;
}

struct SvdOut {
    uint32_t l = 0;
    std::vector<float> s;
    DevBuf<float> u;   // m x l
    DevBuf<float> vtT; // n x l  (V, i.e. Vt transposed), only if want_vt
};

// SvdApprox::direct_svd (RANK mode), svdapprox.rs:721-799
static bool direct_svd_from_q(ae_matrepr& a, DevBuf<float>& q, uint32_t l, bool want_vt, SvdOut& out, const TallFactor* tf = nullptr);

// Eigendecomposition of the l x l (l <= 64) f64 Gram on the HOST: cyclic Jacobi, eigenvalues descending, eigenvectors
// in the columns of v (row-major).  The callers need the spectrum on the host anyway (one synchronisation either
// way); 20 x 20 takes ~50 us here against ~330 us of a single-workgroup device kernel (latency of ~1500 dependent
// LDS round trips).  This is the l x l tail of the reference's gesdd on B (svdapprox.rs:758), not a data path.
static void jacobi_eigh_host(const double* g_in, uint32_t l, std::vector<double>& evals, std::vector<double>& v) {
    std::vector<double> a(g_in, g_in + (size_t)l * l);
    v.assign((size_t)l * l, 0.);
    for (uint32_t i = 0; i < l; i++) v[(size_t)i * l + i] = 1.;
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = 0., diag = 0.;
        for (uint32_t i = 0; i < l; i++)
            for (uint32_t j = 0; j < l; j++) (i == j ? diag : off) += a[(size_t)i * l + j] * a[(size_t)i * l + j];
        if (off <= 1e-26 * diag || off == 0.) break;
        for (uint32_t p = 0; p + 1 < l; p++)
            for (uint32_t q2 = p + 1; q2 < l; q2++) {
                const double apq = a[(size_t)p * l + q2];
                if (apq == 0.) continue;
                const double tau = (a[(size_t)q2 * l + q2] - a[(size_t)p * l + p]) / (2. * apq);
                const double t = (tau >= 0. ? 1. : -1.) / (std::fabs(tau) + std::sqrt(1. + tau * tau));
                const double c = 1. / std::sqrt(1. + t * t), sn = t * c;
                for (uint32_t r = 0; r < l; r++) {  // A <- A J, V <- V J
                    const double ap = a[(size_t)r * l + p], aq = a[(size_t)r * l + q2];
                    a[(size_t)r * l + p] = c * ap - sn * aq;
                    a[(size_t)r * l + q2] = sn * ap + c * aq;
                    const double vp = v[(size_t)r * l + p], vq = v[(size_t)r * l + q2];
                    v[(size_t)r * l + p] = c * vp - sn * vq;
                    v[(size_t)r * l + q2] = sn * vp + c * vq;
                }
                for (uint32_t k = 0; k < l; k++) {  // A <- J^T A
                    const double ap = a[(size_t)p * l + k], aq = a[(size_t)q2 * l + k];
                    a[(size_t)p * l + k] = c * ap - sn * aq;
                    a[(size_t)q2 * l + k] = sn * ap + c * aq;
                }
            }
    }
    std::vector<uint32_t> order(l);
    for (uint32_t i = 0; i < l; i++) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y2) { return a[(size_t)x * l + x] > a[(size_t)y2 * l + y2]; });
    evals.resize(l);
    std::vector<double> vs((size_t)l * l);
    for (uint32_t o = 0; o < l; o++) {
        evals[o] = a[(size_t)order[o] * l + order[o]];
        for (uint32_t r = 0; r < l; r++) vs[(size_t)r * l + o] = v[(size_t)r * l + order[o]];
    }
    v.swap(vs);
}
static void direct_svd_device(ae_matrepr& a, uint64_t rank, uint64_t nbiter, bool want_vt, SvdOut& out) {
    DevBuf<float> q;
    TallFactor tf;
    uint32_t l = subspace_iteration_device(a, rank, nbiter, q, &tf);
    if (direct_svd_from_q(a, q, l, want_vt, out, &tf)) return;
    // the last tall panel's Gram did not factorise (rank deficient): the explicit route
    l = subspace_iteration_device(a, rank, nbiter, q, nullptr, true);
    direct_svd_from_q(a, q, l, want_vt, out);
}
// host: upper Cholesky R of the l x l f64 Gram g (R^T R = g); false if a pivot is not safely positive
static bool chol_upper_host(const double* g, uint32_t l, std::vector<double>& R) {
    R.assign((size_t)l * l, 0.);
    double dmax = 0.;
    for (uint32_t i = 0; i < l; i++) dmax = std::max(dmax, g[(size_t)i * l + i]);
    for (uint32_t j = 0; j < l; j++) {
        for (uint32_t i = j; i < l; i++) {
            double v = g[(size_t)j * l + i];
            for (uint32_t k = 0; k < j; k++) v -= R[(size_t)k * l + j] * R[(size_t)k * l + i];
            if (i == j) {
                if (!(v > 1e-10 * dmax)) return false;
                R[(size_t)j * l + j] = std::sqrt(v);
            } else {
                R[(size_t)j * l + i] = v / R[(size_t)j * l + j];
            }
        }
    }
    return true;
}
// the part of direct_svd after the range approximation Q (m x l), :737-799.  tf: q holds the un-normalised tall panel Y and tf->g its
// Gram (Q = Y R^-1); returns false if that Gram does not factorise (nothing computed: the caller takes the explicit route)
static bool direct_svd_from_q(ae_matrepr& a, DevBuf<float>& q, uint32_t l, bool want_vt, SvdOut& out, const TallFactor* tf_in) {
    const uint64_t m = a.nrows, n = a.ncols;
    const TallFactor* tf = (tf_in && tf_in->deferred) ? tf_in : nullptr;
    const float* q_wide = (tf_in && tf_in->q_wide.n) ? tf_in->q_wide.p : nullptr;
    std::vector<double> hgt, hR;
    if (tf) {
        hgt.resize((size_t)l * l);
        AE_HIP(hipMemcpyAsync(hgt.data(), tf->g, sizeof(double) * l * l, hipMemcpyDeviceToHost, stream()));
        sync();
        if (!chol_upper_host(hgt.data(), l, hR)) { fast_orth().g[0].zero(); fast_orth().g[1].zero(); return false; }
    }
    // B = Q^T A (l x n), kept transposed: Bt = A^T Q (n x l)                       :737-743
    DevBuf<float> bt;
    bt.alloc_pooled(n * l);
    if (q_wide) {   // (the gather copy of Q: spmm_wide_ok)
        if (a.symmetric) spmm(a, q_wide, bt.p, l, kWideLd);
        else { build_transpose(a); spmm(*a.transpose, q_wide, bt.p, l, kWideLd); }
    } else {
        mat_t_mul_panel(a, q.p, bt.p, l);
    }
    if (tf) {   // ... = (A^T Y) R^-1, on the small side
        FastOrth& f = fast_orth();
        launch_small_panel_qr(bt.p, n, l, tf->g, f.flag.p, nullptr, true);
    }
    // svd(B) through the l x l Gram B B^T = U_b S^2 U_b^T                          :758
    DevBuf<double> work;
    work.alloc_pooled(3ull * l * l + l);
    double* g = work.p;
    double* ub = work.p + (uint64_t)l * l;
    double* evals = work.p + 2ull * l * l;
    gram_panel(bt.p, n, l, g);
    std::vector<double> hg((size_t)l * l), hev, hub_sorted;
    AE_HIP(hipMemcpyAsync(hg.data(), g, sizeof(double) * l * l, hipMemcpyDeviceToHost, stream()));
    sync();
    jacobi_eigh_host(hg.data(), l, hev, hub_sorted);
    AE_HIP(hipMemcpyAsync(ub, hub_sorted.data(), sizeof(double) * l * l, hipMemcpyHostToDevice, stream()));
    (void)evals;
    out.l = l;
    out.s.resize(l);
    for (uint32_t i = 0; i < l; i++) out.s[i] = (float)std::sqrt(std::max(hev[i], 0.));
    // U = Q U_b                                                                   :781
    out.u.alloc_pooled(m * l);
    if (tf) {   // Q U_b = Y (R^-1 U_b): back substitution on the l x l side
        std::vector<double> mm((size_t)l * l);
        for (uint32_t c = 0; c < l; c++)
            for (int i = (int)l - 1; i >= 0; i--) {
                double v = hub_sorted[(size_t)i * l + c];
                for (uint32_t k = (uint32_t)i + 1; k < l; k++) v -= hR[(size_t)i * l + k] * mm[(size_t)k * l + c];
                mm[(size_t)i * l + c] = v / hR[(size_t)i * l + i];
            }
        if (l % 4 == 0 && !debug_knob("AE_SVD_APPLY_F64")) {
            // U = Y M through the matrix-core product (Y as an m x l dense matrix, M as its l x l panel, f32 as the reference's own
            // `q.dot(&u_b)`, svdapprox.rs:781): the f64 row-by-row kernel ran at 1 TB/s on a 6.25 M x 20 panel (1.0 ms of a 9.8 ms direct_svd)
            std::vector<float> mf((size_t)l * l);
            for (size_t idx = 0; idx < mf.size(); idx++) mf[idx] = (float)mm[idx];
            DevBuf<float> dmf;
            dmf.alloc_pooled((size_t)l * l);
            dmf.upload(mf.data(), mf.size());
            ae_matrepr view;   // (borrows q's storage: released from the view before it dies)
            view.is_csr = false;
            view.nrows = m; view.ncols = l; view.nnz = m * l;
            view.values.p = q.p; view.values.n = m * l;
            try { mat_mul_panel(view, dmf.p, out.u.p, l); } catch (...) { view.values.p = nullptr; view.values.n = 0; throw; }
            view.values.p = nullptr; view.values.n = 0;
            sync();   // (mf, dmf leave scope)
        } else {
            AE_HIP(hipMemcpyAsync(ub, mm.data(), sizeof(double) * l * l, hipMemcpyHostToDevice, stream()));
            apply_panel(q.p, m, l, ub, l, out.u.p);
            sync();   // (mm leaves scope)
        }
    } else {
        apply_panel(q.p, m, l, ub, l, out.u.p);
    }
    if (want_vt) {
        // Vt = S^-1 U_b^T B  <=>  V = Bt U_b S^-1 ; null directions get zero rows
        const std::vector<double>& hub = hub_sorted;
        std::vector<double> hm(l * l);
        for (uint32_t c = 0; c < l; c++)
            for (uint32_t o = 0; o < l; o++) {
                const double sv = std::sqrt(std::max(hev[o], 0.));
                hm[c * l + o] = (sv > 1e-7 * std::sqrt(std::max(hev[0], 0.)) && sv > 0.) ? hub[c * l + o] / sv : 0.;
            }
        DevBuf<double> dm(l * l);
        dm.upload(hm.data(), l * l);
        out.vtT.alloc(n * l);
        apply_panel(bt.p, n, l, dm.p, l, out.vtT.p);
    }
    sync();
    if (tf) {
        if (orthonormalize_fast_failed()) return false;   // (cannot happen after the host factorisation succeeded; kept as a guard)
        fast_orth().g[0].zero(); fast_orth().g[1].zero();   // (the accumulators are left zeroed: orthonormalize_panel_fast counts on it)
        sync();
    }
    return true;
}

// Leading singular triplets of a dense symmetric matrix, converged: stands in for the full LAPACK
// gesdd of do_full_svd / svd_f32 (graphlaplace.rs:82-94, 296-344), of which the embedder only reads
// s[0..d] and U[:, 1..d] (diffmaps.rs:1213-1236).  Block subspace iteration on A^2 with
// Rayleigh-Ritz every few steps until the first `rank` singular values move by < tol.
void full_svd_leading(ae_matrepr& a, uint32_t rank, std::vector<float>& s, DevBuf<float>& u) {
    const uint64_t n = a.nrows;
    const uint32_t l = (uint32_t)std::min<uint64_t>(n, std::min<uint32_t>(kMaxL, rank + 12));
    rank = std::min<uint32_t>(rank, l);
    DevBuf<float> q(n * l), z(n * l);
    DevBuf<double> work(3ull * l * l + l);
    bool robust = false;  // optimistic CholeskyQR until a pivot fails (checked at the per-iteration sync below)
    auto qr = [&](float* y) {
        if (robust) orthonormalize_panel(y, n, l, work.p);
        else orthonormalize_panel_fast(y, n, l);
    };
    gaussian_fill_device(z.p, n * l, kDefaultSeed, kTagOmega);
    mat_mul_panel(a, z.p, q.p, l);
    qr(q.p);
    std::vector<double> prev(l, 0.), cur(l);
    double* g = work.p;
    double* ub = work.p + (uint64_t)l * l;
    double* evals = work.p + 2ull * l * l;
    const int max_outer = 400;
    for (int it = 0; it < max_outer; it++) {
        for (int inner = 0; inner < 4; inner++) {
            mat_t_mul_panel(a, q.p, z.p, l);
            qr(z.p);
            mat_mul_panel(a, z.p, q.p, l);
            qr(q.p);
        }
        if (!robust && orthonormalize_fast_failed()) {  // rank-deficient panel: restart on the eigen route
            robust = true;
            gaussian_fill_device(z.p, n * l, kDefaultSeed, kTagOmega);
            mat_mul_panel(a, z.p, q.p, l);
            qr(q.p);
            std::fill(prev.begin(), prev.end(), 0.);
            it = -1;
            continue;
        }
        // Rayleigh-Ritz on B = Q^T A: sigma^2 = eig(B B^T), rotate Q onto the Ritz vectors
        mat_t_mul_panel(a, q.p, z.p, l);
        gram_panel(z.p, n, l, g);
        std::vector<double> hg((size_t)l * l), hub;
        AE_HIP(hipMemcpyAsync(hg.data(), g, sizeof(double) * l * l, hipMemcpyDeviceToHost, stream()));
        sync();
        jacobi_eigh_host(hg.data(), l, cur, hub);
        AE_HIP(hipMemcpyAsync(ub, hub.data(), sizeof(double) * l * l, hipMemcpyHostToDevice, stream()));
        apply_panel(q.p, n, l, ub, l, q.p);
        sync();  // hub leaves scope
        (void)evals;
        double delta = 0.;
        for (uint32_t i = 0; i < rank; i++) {
            double a1 = std::sqrt(std::max(cur[i], 0.)), a0 = std::sqrt(std::max(prev[i], 0.));
            delta = std::max(delta, std::fabs(a1 - a0) / std::max(a1, 1e-30));
        }
        prev = cur;
        if (it > 0 && delta < 2e-8) break;
    }
    s.resize(rank);
    for (uint32_t i = 0; i < rank; i++) s[i] = (float)std::sqrt(std::max(prev[i], 0.));
    // keep the first `rank` Ritz vectors
    std::vector<double> sel((uint64_t)l * rank, 0.);
    for (uint32_t i = 0; i < rank; i++) sel[(uint64_t)i * rank + i] = 1.;
    DevBuf<double> dsel((uint64_t)l * rank);
    dsel.upload(sel.data(), (uint64_t)l * rank);
    u.alloc(n * rank);
    apply_panel(q.p, n, l, dsel.p, rank, u.p);
    sync();
}

// ---- adaptative_range_finder_matrep, svdapprox.rs:444-597 (Halko-Martinsson-Tropp algorithm 4.2) ------------
// Vector-at-a-time, as the reference: r probe vectors y_k = A w_k (w ~ N(0, 1/n)), each iteration turns the
// oldest into the next basis vector q, replaces it by a fresh probe made orthogonal to Q, removes q from the
// other probes, and stops when the largest probe norm is below the threshold, at max_rank, or on a vanishing
// vector.  Q and Y live on the device as row-major panels (m x max_rank, m x r); the two decisions per
// iteration (||y_j||, max_k ||y_k||) are the only host round trips.
__global__ void col_dots_kernel(const float* __restrict__ qp, uint32_t ldq, uint32_t nq, const float* __restrict__ y, uint32_t ldy,
                                uint64_t rows, double* __restrict__ out) {
    // out[i] += sum_row Q[row, i] * y[row]   (one wave per block of rows, f64 accumulation)
    const uint32_t i = blockIdx.y;
    double acc = 0.;
    for (uint64_t row = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; row < rows; row += (uint64_t)gridDim.x * blockDim.x)
        acc += (double)qp[row * ldq + i] * (double)y[row * ldy];
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    __shared__ double part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(&out[i], part[0] + part[1] + part[2] + part[3]);
    (void)nq;
}
__global__ void col_axpy_kernel(float* __restrict__ y, uint32_t ldy, const float* __restrict__ qp, uint32_t ldq, uint32_t nq,
                                const double* __restrict__ coef, uint64_t rows) {
    // y[row] -= sum_i Q[row, i] * coef[i]   (orthogonalize_with_q, :975-992)
    const uint64_t row = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (row >= rows) return;
    double p = 0.;
    for (uint32_t i = 0; i < nq; i++) p += (double)qp[row * ldq + i] * coef[i];
    y[row * ldy] = (float)((double)y[row * ldy] - p);
}
__global__ void col_norms2_kernel(const float* __restrict__ y, uint32_t ldy, uint32_t ncol, uint64_t rows, double* __restrict__ out) {
    const uint32_t k = blockIdx.y;
    double acc = 0.;
    for (uint64_t row = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; row < rows; row += (uint64_t)gridDim.x * blockDim.x) {
        const double v = (double)y[row * ldy + k];
        acc += v * v;
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    __shared__ double part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(&out[k], part[0] + part[1] + part[2] + part[3]);
    (void)ncol;
}
__global__ void col_scale_copy_kernel(const float* __restrict__ src, uint32_t lds, float* __restrict__ dst, uint32_t ldd, float scale, uint64_t rows) {
    const uint64_t row = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (row < rows) dst[row * ldd] = src[row * lds] * scale;
}
__global__ void scale_vec_f32_kernel(float* __restrict__ x, uint64_t n, float c) {
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i < n) x[i] *= c;
}

uint32_t adaptive_range_finder_device(ae_matrepr& a, double epsil, uint64_t r_arg, uint64_t max_rank_arg, DevBuf<float>& q_out) {
    const uint64_t m = a.nrows, n = a.ncols;
    const uint32_t r = (uint32_t)r_arg;
    if (r < 1 || r > (uint32_t)kMaxL) fail(AE_ERR_INVALID_ARG, "range finder: step must be in 1..%d", kMaxL);
    if (max_rank_arg == 0) fail(AE_ERR_INVALID_ARG, "range finder: max_rank must be positive");
    // the basis may be wider than a panel (kMaxL bounds the QR / SVD panels, not this vector-at-a-time finder)
    const uint32_t max_rank = (uint32_t)std::min<uint64_t>(max_rank_arg, std::min<uint64_t>(4096ull, std::min(m, n) + 1));
    const unsigned rg = std::min<unsigned>(blocks_for(m, 256), 512);
    DevBuf<float> omega(n * r), yp(m * r), qp(m * (uint64_t)max_rank), wv(n), yv(m);
    DevBuf<double> coef(std::max<uint32_t>(max_rank, kMaxL)), norms(kMaxL);
    std::vector<double> hn(r);
    auto col_norms = [&]() {  // squared norms of the r probes
        norms.zero();
        hipLaunchKernelGGL(col_norms2_kernel, dim3(rg, r), dim3(256), 0, stream(), (const float*)yp.p, r, r, m, norms.p);
        norms.download(hn.data(), r);
    };
    auto orthogonalize = [&](float* y, uint32_t ldy, uint32_t first, uint32_t nq) {  // y -= sum_{i in [first, first+nq)} q_i (q_i . y)
        if (!nq) return;
        coef.zero();
        hipLaunchKernelGGL(col_dots_kernel, dim3(rg, nq), dim3(256), 0, stream(), (const float*)qp.p + first, max_rank, nq, (const float*)y, ldy, m, coef.p);
        hipLaunchKernelGGL(col_axpy_kernel, dim3(blocks_for(m, 256)), dim3(256), 0, stream(), y, ldy, (const float*)qp.p + first, max_rank, nq,
                           (const double*)coef.p, m);
    };
    const float coeff_norm = (float)(1.0 / std::sqrt((double)n));                       // :481
    gaussian_fill_device(omega.p, n * r, kDefaultSeed, kTagOmega);                        // :479
    hipLaunchKernelGGL(scale_vec_f32_kernel, dim3(blocks_for(n * r, 256)), dim3(256), 0, stream(), omega.p, n * r, coeff_norm);
    mat_mul_panel(a, omega.p, yp.p, r);                                                   // :485-491
    col_norms();
    double norm_sup = 0.;
    for (uint32_t k = 0; k < r; k++) {
        if (!(hn[k] == hn[k])) fail(AE_ERR_SVD, "adaptative_range_finder: NaN in the probe norms");  // :505-509
        norm_sup = std::max(norm_sup, std::sqrt(hn[k]));
    }
    const double stop_rel = epsil / (10. * std::sqrt(2. * M_PI));                         // :465
    const double stop_val = norm_sup * stop_rel;                                          // :515
    const uint64_t max_iter = std::min(m, n);
    uint32_t nq = 0, j = 0;
    uint64_t nb_iter = 0;
    while (norm_sup > stop_val && nb_iter <= max_iter && nq < max_rank) {                 // :517
        orthogonalize(yp.p + j, r, 0, nq);                                                // :519-521
        norms.zero();
        hipLaunchKernelGGL(col_norms2_kernel, dim3(rg, 1), dim3(256), 0, stream(), (const float*)yp.p + j, r, 1u, m, norms.p);
        double nj2 = 0.;
        norms.download(&nj2, 1);
        const double n_j = std::sqrt(nj2);
        if (n_j < std::sqrt((double)FLT_EPSILON)) break;                                  // :524-532
        hipLaunchKernelGGL(col_scale_copy_kernel, dim3(blocks_for(m, 256)), dim3(256), 0, stream(), (const float*)yp.p + j, r, qp.p + nq, max_rank,
                           (float)(1.0 / n_j), m);                                        // :533-535
        nq++;
        gaussian_fill_device(wv.p, n, kDefaultSeed, 0xFFFE0000u + (uint32_t)nb_iter);     // :537-538
        hipLaunchKernelGGL(scale_vec_f32_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, stream(), wv.p, n, coeff_norm);
        mat_mul_panel(a, wv.p, yv.p, 1);                                                  // :539
        orthogonalize(yv.p, 1, 0, nq);                                                    // :541
        hipLaunchKernelGGL(col_scale_copy_kernel, dim3(blocks_for(m, 256)), dim3(256), 0, stream(), (const float*)yv.p, 1u, yp.p + j, r, 1.0f, m);  // :544
        for (uint32_t k = 0; k < r; k++)                                                  // :546-553
            if (k != j) orthogonalize(yp.p + k, r, nq - 1, 1);
        col_norms();                                                                      // :555-561
        norm_sup = 0.;
        for (uint32_t k = 0; k < r; k++) norm_sup = std::max(norm_sup, std::sqrt(hn[k]));
        j = (j + 1) % r;
        nb_iter++;
    }
    // compact m x nq copy
    q_out.alloc(m * (uint64_t)std::max(nq, 1u));
    if (nq) {
        AE_HIP(hipMemcpy2DAsync(q_out.p, sizeof(float) * nq, qp.p, sizeof(float) * max_rank, sizeof(float) * nq, m, hipMemcpyDeviceToDevice, stream()));
    }
    sync();
    return nq;
}

void direct_svd_rank(ae_matrepr& a, uint64_t rank, uint64_t nbiter, std::vector<float>& s, DevBuf<float>& u) {
    SvdOut o;
    direct_svd_device(a, rank, nbiter, false, o);
    s = o.s;
    u = std::move(o.u);
}

}  // namespace ae

extern "C" {

int32_t ae_matrepr_from_csr(const uint64_t* indptr, const uint32_t* indices, const float* values, uint64_t nrows, uint64_t ncols,
                            ae_matrepr** out) {
    return guard([&] {
        require_device();
        if (!indptr || !out || nrows == 0 || ncols == 0) fail(AE_ERR_INVALID_ARG, "null argument or empty matrix");
        const uint64_t nnz = indptr[nrows];
        if (nnz && (!indices || !values)) fail(AE_ERR_INVALID_ARG, "null indices/values");
        for (uint64_t i = 0; i < nrows; i++)
            if (indptr[i + 1] < indptr[i]) fail(AE_ERR_INVALID_ARG, "indptr not monotone");
        for (uint64_t e = 0; e < nnz; e++)
            if (indices[e] >= ncols) fail(AE_ERR_INVALID_ARG, "column index out of range");
        std::unique_ptr<ae_matrepr> m(new ae_matrepr);
        m->is_csr = true;
        m->nrows = nrows; m->ncols = ncols; m->nnz = nnz;
        m->indptr.alloc(nrows + 1);
        m->indptr.upload(indptr, nrows + 1);
        m->indices.alloc(nnz ? nnz : 1);
        m->values.alloc(nnz ? nnz : 1);
        if (nnz) { m->indices.upload(indices, nnz); m->values.upload(values, nnz); }
        sync();
        *out = m.release();
    });
}
int32_t ae_matrepr_from_dense(const float* values, uint64_t nrows, uint64_t ncols, ae_matrepr** out) {
    return guard([&] {
        require_device();
        if (!values || !out || nrows == 0 || ncols == 0) fail(AE_ERR_INVALID_ARG, "null argument or empty matrix");
        std::unique_ptr<ae_matrepr> m(new ae_matrepr);
        m->is_csr = false;
        m->nrows = nrows; m->ncols = ncols; m->nnz = nrows * ncols;
        m->values.alloc(nrows * ncols);
        m->values.upload(values, nrows * ncols);
        sync();
        *out = m.release();
    });
}
int32_t ae_matrepr_destroy(ae_matrepr* m) {
    return guard([&] { delete m; });
}

int32_t ae_subspace_iteration(const ae_matrepr* m, uint64_t rank, uint64_t nbiter, float* q, uint64_t* l_out) {
    return guard([&] {
        require_device();
        if (!m || !q) fail(AE_ERR_INVALID_ARG, "null argument");
        DevBuf<float> dq;
        const uint32_t l = subspace_iteration_device(*const_cast<ae_matrepr*>(m), rank, nbiter, dq);
        dq.download(q, m->nrows * l);
        if (l_out) *l_out = l;
    });
}

int32_t ae_svd_approx_rank(const ae_matrepr* m, uint64_t rank, uint64_t nbiter, float* s, float* u, float* vt, uint64_t* l_out) {
    return guard([&] {
        require_device();
        if (!m || !s) fail(AE_ERR_INVALID_ARG, "null argument");
        SvdOut o;
        direct_svd_device(*const_cast<ae_matrepr*>(m), rank, nbiter, vt != nullptr, o);
        memcpy(s, o.s.data(), sizeof(float) * o.l);
        if (u) o.u.download(u, m->nrows * o.l);
        if (vt) {
            DevBuf<float> t(m->ncols * o.l);
            hipLaunchKernelGGL(transpose_dense_kernel, dim3(blocks_for(m->ncols * o.l, 256)), dim3(256), 0, stream(), o.vtT.p, m->ncols,
                               (uint64_t)o.l, t.p);
            check_launch("transpose_dense");
            t.download(vt, m->ncols * o.l);
        }
        if (l_out) *l_out = o.l;
    });
}

int32_t ae_adaptative_range_finder(const ae_matrepr* m, double epsil, uint64_t r, uint64_t max_rank, float* q, uint64_t* l_out) {
    return guard([&] {
        require_device();
        if (!m || !q || !l_out) fail(AE_ERR_INVALID_ARG, "null argument");
        DevBuf<float> dq;
        const uint32_t l = adaptive_range_finder_device(*const_cast<ae_matrepr*>(m), epsil, r, max_rank, dq);
        if (l) dq.download(q, m->nrows * l);
        *l_out = l;
    });
}

int32_t ae_svd_approx_epsil(const ae_matrepr* m, double epsil, uint64_t step, uint64_t max_rank, float* s, float* u, float* vt,
                            uint64_t* l_out) {
    return guard([&] {
        require_device();
        if (!m || !s || !l_out) fail(AE_ERR_INVALID_ARG, "null argument");
        if (step <= 1) step = 2;  // RangePrecision::new, svdapprox.rs:167-179
        DevBuf<float> dq;
        ae_matrepr& a = *const_cast<ae_matrepr*>(m);
        const uint32_t l = adaptive_range_finder_device(a, epsil, step, std::min<uint64_t>(max_rank, (uint64_t)kMaxL), dq);
        if (l == 0) fail(AE_ERR_SVD, "adaptative range finder returned an empty basis");
        SvdOut o;
        direct_svd_from_q(a, dq, l, vt != nullptr, o);
        memcpy(s, o.s.data(), sizeof(float) * o.l);
        if (u) o.u.download(u, m->nrows * o.l);
        if (vt) {
            DevBuf<float> t(m->ncols * o.l);
            hipLaunchKernelGGL(transpose_dense_kernel, dim3(blocks_for(m->ncols * o.l, 256)), dim3(256), 0, stream(), o.vtT.p, m->ncols,
                               (uint64_t)o.l, t.p);
            check_launch("transpose_dense");
            t.download(vt, m->ncols * o.l);
        }
        *l_out = o.l;
    });
}

int32_t ae_transpose_dense_mult(const ae_matrepr* m, const float* q, uint64_t l, float* b) {
    return guard([&] {
        require_device();
        if (!m || !q || !b || l == 0 || l > (uint64_t)kMaxL) fail(AE_ERR_INVALID_ARG, "bad argument");
        DevBuf<float> dq(m->nrows * l), bt(m->ncols * l), bb(m->ncols * l);
        dq.upload(q, m->nrows * l);
        mat_t_mul_panel(*const_cast<ae_matrepr*>(m), dq.p, bt.p, (uint32_t)l);
        hipLaunchKernelGGL(transpose_dense_kernel, dim3(blocks_for(m->ncols * l, 256)), dim3(256), 0, stream(), bt.p, m->ncols, l, bb.p);
        check_launch("transpose_dense");
        bb.download(b, m->ncols * l);
    });
}

}  // extern "C"
