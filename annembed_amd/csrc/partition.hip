// partition.hip -- the locality partitioner of the multi-GPU path: a graph in the REFERENCE's node order goes in, contiguous node
// ranges with few cross-range edges come out.
//
// The reference numbers its nodes in IndexSet insertion order of the HNSW points, i.e. in file order
// (src/fromhnsw/kgraph.rs:489,500): the ids carry no locality, and a sharded CE loop that cut such a graph into contiguous id
// ranges would find (world - 1) / world of its edges crossing shards.  SURVEY 8e: "contiguous node ranges of N/8 AFTER LOCALITY
// REORDERING".  This file is that reordering, on the device, inside the library (Embedder::embed with a communicator calls it;
// ae_kgraph_partition / ae_kgraph_permuted expose it):
//
//   1. connected components of the undirected graph: union-find with compare-and-swap hooking of the larger root under the
//      smaller (the root of a component is its smallest node id whatever the order of the hooks: the labels are deterministic),
//      re-run until a pass over the edges finds both ends of every edge under one root (kernel boundaries are the only
//      coherence the passes rely on);
//   2. the components are packed into the ranks by recursive two-way LPT (largest first into the side with more room left; sides
//      sized world/2 : world - world/2); a side that ends more than 1.5 % off its target gets it back from ONE component, which is
//      split -- a disconnected graph of many components (kNN graphs of separated clusters) is partitioned with no cut edge at all;
//   3. a component (or a piece of one) that has to be split is split by COORDINATE BISECTION of the coordinates the caller gives
//      (the diffusion-map initialisation, or the projection initialisation of the hierarchical embedding: embedder.rs:231-269 --
//      both place graph neighbours near each other): its nodes are sorted along the axis of largest variance of the piece and cut
//      at the count the packing asks for; pieces are split again along THEIR widest axis further down the recursion (recursive
//      coordinate bisection).  The coordinates are first SMOOTHED ON THE GRAPH (a few dozen lazy random-walk steps inside the piece:
//      what is left are its slow modes, the directions a short cut is perpendicular to -- the reference's rank-20 / 5-iteration
//      initialisation resolves clusters, not the slow modes of a sheet), and the cut is then polished by a balanced majority vote of
//      the nodes along it.  Without coordinates the walk starts from noise (a partition; a worse one).
//
// Output: order[pos] = caller's id of the node at position pos, the ranks' position ranges, and a report (components, splits,
// cross-range edge mass as the sharded time-sliced mode will see it, imbalance).
#include "internal.h"

#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <functional>

namespace ae {
void sort_pairs_u32_u32(uint32_t* d_keys_in, uint32_t* d_keys_out, uint32_t* d_vals_in, uint32_t* d_vals_out, uint64_t count, unsigned end_bit);
}

using namespace ae;

namespace {

// ---------------------------------------------------------------------------------------------
// connected components
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t ld_agent(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// the root above x (parent[v] <= v always; a root is its own parent), halving the path on the way
__device__ __forceinline__ uint32_t cc_root(uint32_t* parent, uint32_t x) {
    uint32_t cur = ld_agent(parent + x);
    if (cur != x) {
        uint32_t prev = x, next;
        while (cur > (next = ld_agent(parent + cur))) {
            st_agent(parent + prev, next);
            prev = cur;
            cur = next;
        }
    }
    return cur;
}
__global__ void __launch_bounds__(256) cc_init_kernel(uint64_t n, uint32_t* __restrict__ parent) {
    const uint64_t v = blockIdx.x * 256ull + threadIdx.x;
    if (v < n) parent[v] = (uint32_t)v;
}
__global__ void __launch_bounds__(256) cc_hook_kernel(uint64_t n, uint32_t uniform_k, const uint64_t* __restrict__ indptr, const uint32_t* __restrict__ nbr,
                                                      uint32_t* parent) {
    const uint64_t u = blockIdx.x * 256ull + threadIdx.x;
    if (u >= n) return;
    uint64_t b, e;
    if (uniform_k) { b = u * uniform_k; e = b + uniform_k; } else { b = indptr[u]; e = indptr[u + 1]; }
    for (uint64_t x = b; x < e; x++) {
        uint32_t ru = cc_root(parent, (uint32_t)u), rv = cc_root(parent, nbr[x]);
        for (int guard = 0; ru != rv && guard < 1024; guard++) {
            const uint32_t hi = ru > rv ? ru : rv, lo = ru > rv ? rv : ru;
            const uint32_t old = atomicCAS(parent + hi, hi, lo);
            if (old == hi) break;               // hooked: hi was still a root
            ru = cc_root(parent, old);          // somebody hooked hi meanwhile: go on from where it hangs now
            rv = lo;
        }
    }
}
__global__ void __launch_bounds__(256) cc_flatten_kernel(uint64_t n, uint32_t* parent) {
    const uint64_t v = blockIdx.x * 256ull + threadIdx.x;
    if (v >= n) return;
    uint32_t r = (uint32_t)v, p = ld_agent(parent + r);
    while (p != r) { r = p; p = ld_agent(parent + r); }
    st_agent(parent + v, r);
}
// edges whose ends are under different labels (after a flatten): zero = done
__global__ void __launch_bounds__(256) cc_check_kernel(uint64_t n, uint32_t uniform_k, const uint64_t* __restrict__ indptr, const uint32_t* __restrict__ nbr,
                                                       const uint32_t* __restrict__ label, unsigned long long* __restrict__ bad) {
    const uint64_t u = blockIdx.x * 256ull + threadIdx.x;
    unsigned long long c = 0;
    if (u < n) {
        uint64_t b, e;
        if (uniform_k) { b = u * uniform_k; e = b + uniform_k; } else { b = indptr[u]; e = indptr[u + 1]; }
        const uint32_t lu = label[u];
        for (uint64_t x = b; x < e; x++) c += label[nbr[x]] != lu ? 1ull : 0ull;
    }
    for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(bad, c);
}
__global__ void __launch_bounds__(256) cc_sizes_kernel(uint64_t n, const uint32_t* __restrict__ label, uint32_t* __restrict__ size) {
    const uint64_t v = blockIdx.x * 256ull + threadIdx.x;
    if (v < n) atomicAdd(size + label[v], 1u);
}
__global__ void __launch_bounds__(256) iota_kernel(uint64_t n, uint32_t* __restrict__ x) {
    const uint64_t v = blockIdx.x * 256ull + threadIdx.x;
    if (v < n) x[v] = (uint32_t)v;
}
// roots in ascending id order: flag -> scan -> compact {root's size}
__global__ void __launch_bounds__(256) cc_root_flag_kernel(uint64_t n, const uint32_t* __restrict__ label, uint32_t* __restrict__ flag) {
    const uint64_t v = blockIdx.x * 256ull + threadIdx.x;
    if (v < n) flag[v] = label[v] == (uint32_t)v ? 1u : 0u;
}
__global__ void __launch_bounds__(256) cc_compact_kernel(uint64_t n, const uint32_t* __restrict__ label, const uint32_t* __restrict__ idx,
                                                         const uint32_t* __restrict__ size, uint32_t* __restrict__ out_size) {
    const uint64_t v = blockIdx.x * 256ull + threadIdx.x;
    if (v < n && label[v] == (uint32_t)v) out_size[idx[v]] = size[v];
}

// ---------------------------------------------------------------------------------------------
// coordinate bisection of one piece: order[b, e) sorted along the piece's widest axis
// ---------------------------------------------------------------------------------------------
constexpr int kStatBlocks = 512;
// per block and axis: sum and sum of squares (f64) of the piece's coordinates -> host adds the blocks in order (deterministic)
__global__ void __launch_bounds__(256) piece_stats_kernel(const uint32_t* __restrict__ order, uint64_t b, uint64_t e, const float* __restrict__ y, uint32_t dim,
                                                          uint32_t stride, double* __restrict__ out) {
    __shared__ double s_acc[4][2 * 64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (uint32_t a = 0; a < dim; a++) {
        double s = 0., q = 0.;
        for (uint64_t p = b + blockIdx.x * 256ull + threadIdx.x; p < e; p += (uint64_t)gridDim.x * 256ull) {
            const double v = (double)y[(uint64_t)order[p] * stride + a];
            s += v;
            q += v * v;
        }
        for (int off = 32; off > 0; off >>= 1) { s += __shfl_xor(s, off); q += __shfl_xor(q, off); }
        if (lane == 0) { s_acc[wave][2 * a] = s; s_acc[wave][2 * a + 1] = q; }
    }
    __syncthreads();
    if (threadIdx.x < 2 * dim)
        out[(uint64_t)blockIdx.x * 2 * dim + threadIdx.x] = s_acc[0][threadIdx.x] + s_acc[1][threadIdx.x] + s_acc[2][threadIdx.x] + s_acc[3][threadIdx.x];
}
__device__ __forceinline__ uint32_t pcg_hash(uint32_t v) {
    const uint32_t state = v * 747796405u + 2891336453u;
    const uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
    return (word >> 22u) ^ word;
}
// floats in an order-preserving u32 encoding
__device__ __forceinline__ uint32_t ordered_bits(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__global__ void __launch_bounds__(256) piece_keys_kernel(const uint32_t* __restrict__ order, uint64_t b, uint64_t count, const float* __restrict__ y, uint32_t axis,
                                                         uint32_t stride, uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    const uint64_t p = blockIdx.x * 256ull + threadIdx.x;
    if (p >= count) return;
    const uint32_t v = order[b + p];
    const float f = y[(uint64_t)v * stride + axis];
    keys[p] = ordered_bits(f == f ? f : 0.f);
    vals[p] = v;
}

// ---------------------------------------------------------------------------------------------
// the coordinates a piece is cut along: the caller's, smoothed on the graph
// ---------------------------------------------------------------------------------------------
// z (kZ columns per node) starts as the first columns of the caller's coordinates (or hashed noise when there are none) and takes
// lazy random-walk steps restricted to the piece: z_u <- (z_u + mean of z over u's out-neighbours inside the piece) / 2.  A step
// damps a mode of wavelength L hops by ~(1 - c / L^2): a few dozen steps remove everything shorter than ~10 hops and leave the slow
// modes -- the directions a short cut is perpendicular to (power iteration towards the Fiedler vectors, started from an embedding
// that already knows the clusters).  Pure gathers: deterministic.
constexpr uint32_t kZ = 4;
__global__ void __launch_bounds__(256) smooth_init_kernel(const uint32_t* __restrict__ order, uint64_t b, uint64_t count, const float* __restrict__ y, uint32_t ydim,
                                                          uint32_t ystride, float* __restrict__ z, uint8_t* __restrict__ side) {
    const uint64_t p = blockIdx.x * 256ull + threadIdx.x;
    if (p >= count) return;
    const uint32_t v = order[b + p];
    side[v] = 1;
    for (uint32_t a = 0; a < kZ; a++) {
        float f;
        if (y && a < ydim) { f = y[(uint64_t)v * ystride + a]; f = f == f ? f : 0.f; }
        else f = (float)(pcg_hash(v * kZ + a + 0x51ED270Bu) >> 8) * (1.0f / 16777216.0f) - 0.5f;
        z[(uint64_t)v * kZ + a] = f;
    }
}
__global__ void __launch_bounds__(256) smooth_step_kernel(const uint32_t* __restrict__ order, uint64_t b, uint64_t count, uint32_t uniform_k, const uint64_t* __restrict__ indptr,
                                                          const uint32_t* __restrict__ nbr, const uint8_t* __restrict__ side, const float* __restrict__ z, float* __restrict__ z2) {
    const uint64_t p = blockIdx.x * 256ull + threadIdx.x;
    if (p >= count) return;
    const uint32_t u = order[b + p];
    uint64_t rb, re;
    if (uniform_k) { rb = (uint64_t)u * uniform_k; re = rb + uniform_k; } else { rb = indptr[u]; re = indptr[u + 1]; }
    float acc[kZ] = {0.f, 0.f, 0.f, 0.f};
    uint32_t cnt = 0;
    for (uint64_t x = rb; x < re; x++) {
        const uint32_t v = nbr[x];
        if (!side[v]) continue;
        const float4 zv = *reinterpret_cast<const float4*>(z + (uint64_t)v * kZ);
        acc[0] += zv.x; acc[1] += zv.y; acc[2] += zv.z; acc[3] += zv.w;
        cnt++;
    }
    const float4 zu = *reinterpret_cast<const float4*>(z + (uint64_t)u * kZ);
    float4 out = zu;
    if (cnt) {
        const float inv = 0.5f / (float)cnt;
        out.x = 0.5f * zu.x + inv * acc[0]; out.y = 0.5f * zu.y + inv * acc[1]; out.z = 0.5f * zu.z + inv * acc[2]; out.w = 0.5f * zu.w + inv * acc[3];
    }
    *reinterpret_cast<float4*>(z2 + (uint64_t)u * kZ) = out;
}
// z <- (z - mean) / std per column (the constant is the walk's dominant mode: it goes; f32 range is kept)
struct ZNorm { float mean[kZ], inv_std[kZ]; };
__global__ void __launch_bounds__(256) smooth_norm_kernel(const uint32_t* __restrict__ order, uint64_t b, uint64_t count, ZNorm nm, float* __restrict__ z) {
    const uint64_t p = blockIdx.x * 256ull + threadIdx.x;
    if (p >= count) return;
    const uint32_t v = order[b + p];
    for (uint32_t a = 0; a < kZ; a++) z[(uint64_t)v * kZ + a] = (z[(uint64_t)v * kZ + a] - nm.mean[a]) * nm.inv_std[a];
}

// ---------------------------------------------------------------------------------------------
// refinement of a bisection: balanced majority vote along the cut
// ---------------------------------------------------------------------------------------------
// The coordinates a piece is cut along are as good as the embedding they come from, and the reference's diffusion-map initialisation
// (rank 20, 5 subspace iterations: graphlaplace.rs:97-125) resolves clusters, not the slow modes of a sheet: on a graph without
// clusters its level sets are wiggly curves and the coordinate cut is several times longer than it has to be.  So the cut is smoothed
// on the GRAPH: every node of the piece adds up the (quantised) probability mass of its edges -- both directions -- to its own side and to
// the other; nodes that would cut less mass on the other side are candidates, half of them (a coin per node and round: neighbours do not
// swap places for ever) move, the same number from either side (the larger candidate set is thinned by a hash threshold; the sides'
// sizes are steered back to the target every round).  Integer atomics only: the result does not depend on the order of execution.
constexpr uint8_t kSideNone = 0, kSideLeft = 1, kSideRight = 2;
__global__ void __launch_bounds__(256) refine_set_side_kernel(const uint32_t* __restrict__ order, uint64_t b, uint64_t count, uint64_t left_count, uint8_t* __restrict__ side,
                                                              long long* __restrict__ gain, int clear) {
    const uint64_t p = blockIdx.x * 256ull + threadIdx.x;
    if (p >= count) return;
    const uint32_t v = order[b + p];
    if (clear) { side[v] = kSideNone; return; }
    side[v] = p < left_count ? kSideLeft : kSideRight;
    gain[v] = 0;
}
__global__ void __launch_bounds__(256) refine_gain_kernel(const uint32_t* __restrict__ order, uint64_t b, uint64_t count, uint32_t uniform_k, const uint64_t* __restrict__ indptr,
                                                          const uint32_t* __restrict__ nbr, const float* __restrict__ proba, const uint8_t* __restrict__ side,
                                                          long long* __restrict__ gain) {
    // (64-bit gains: an edge weighs up to 4096 in both directions, and a hub of in-degree ~5e5 inside one piece would wrap 32 bits)
    const uint64_t p = blockIdx.x * 256ull + threadIdx.x;
    if (p >= count) return;
    const uint32_t u = order[b + p];
    uint64_t rb, re;
    if (uniform_k) { rb = (uint64_t)u * uniform_k; re = rb + uniform_k; } else { rb = indptr[u]; re = indptr[u + 1]; }
    const uint8_t su = side[u];
    long long mine = 0;
    for (uint64_t x = rb; x < re; x++) {
        const uint32_t v = nbr[x];
        const uint8_t sv = side[v];
        if (sv == kSideNone) continue;   // an edge out of the piece: it is cut (or not) whatever side u takes... by another bisection
        const int32_t w = proba ? max(1, (int32_t)(proba[x] * 4096.f)) : 1;
        const int32_t s = su == sv ? -w : w;
        mine += s;
        atomicAdd(reinterpret_cast<unsigned long long*>(gain + v), (unsigned long long)(long long)s);   // (two's complement: adds a signed weight)
    }
    if (mine) atomicAdd(reinterpret_cast<unsigned long long*>(gain + u), (unsigned long long)mine);
}
// counts[0 / 1]: candidates on the left / right (gain > 0 and this round's coin)
__global__ void __launch_bounds__(256) refine_count_kernel(const uint32_t* __restrict__ order, uint64_t b, uint64_t count, const uint8_t* __restrict__ side,
                                                           const long long* __restrict__ gain, uint32_t round_key, unsigned long long* __restrict__ counts) {
    const uint64_t p = blockIdx.x * 256ull + threadIdx.x;
    unsigned long long cl = 0, cr = 0;
    if (p < count) {
        const uint32_t v = order[b + p];
        if (gain[v] > 0 && (pcg_hash(v ^ round_key) & 1u)) { if (side[v] == kSideLeft) cl = 1; else cr = 1; }
    }
    for (int off = 32; off > 0; off >>= 1) { cl += __shfl_xor(cl, off); cr += __shfl_xor(cr, off); }
    if ((threadIdx.x & 63) == 0) { if (cl) atomicAdd(counts, cl); if (cr) atomicAdd(counts + 1, cr); }
}
// candidates move with probability thr / 2^32 of their side; gains are cleared for the next round; counts[2]: moved left -> right, [3]: right -> left
__global__ void __launch_bounds__(256) refine_move_kernel(const uint32_t* __restrict__ order, uint64_t b, uint64_t count, uint8_t* __restrict__ side, long long* __restrict__ gain,
                                                          uint32_t round_key, uint32_t thr_left, uint32_t thr_right, unsigned long long* __restrict__ counts) {
    const uint64_t p = blockIdx.x * 256ull + threadIdx.x;
    unsigned long long ml = 0, mr = 0;
    if (p < count) {
        const uint32_t v = order[b + p];
        const uint8_t sv = side[v];
        if (gain[v] > 0 && (pcg_hash(v ^ round_key) & 1u)) {
            const uint32_t h = pcg_hash(v * 0x9E3779B9u + round_key);
            if (sv == kSideLeft && h <= thr_left) { side[v] = kSideRight; ml = 1; }
            else if (sv == kSideRight && h <= thr_right) { side[v] = kSideLeft; mr = 1; }
        }
        gain[v] = 0;
    }
    for (int off = 32; off > 0; off >>= 1) { ml += __shfl_xor(ml, off); mr += __shfl_xor(mr, off); }
    if ((threadIdx.x & 63) == 0) { if (ml) atomicAdd(counts + 2, ml); if (mr) atomicAdd(counts + 3, mr); }
}
__global__ void __launch_bounds__(256) refine_keys_kernel(const uint32_t* __restrict__ order, uint64_t b, uint64_t count, const uint8_t* __restrict__ side,
                                                          uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    const uint64_t p = blockIdx.x * 256ull + threadIdx.x;
    if (p >= count) return;
    const uint32_t v = order[b + p];
    keys[p] = side[v] == kSideLeft ? 0u : 1u;
    vals[p] = v;
}

// ---------------------------------------------------------------------------------------------
// assembling the result
// ---------------------------------------------------------------------------------------------
// the pieces tile [0, n) of `order` (begin ascending); piece g goes to positions dest[g] ... of the output
__global__ void __launch_bounds__(256) assemble_kernel(uint64_t n, const uint32_t* __restrict__ order, const uint64_t* __restrict__ begin, const uint64_t* __restrict__ dest,
                                                       uint32_t pieces, uint32_t* __restrict__ out_order, uint32_t* __restrict__ out_perm) {
    const uint64_t p = blockIdx.x * 256ull + threadIdx.x;
    if (p >= n) return;
    uint32_t lo = 0, hi = pieces;   // last piece with begin <= p
    while (hi - lo > 1u) {
        const uint32_t mid = (lo + hi) >> 1;
        if (begin[mid] <= p) lo = mid; else hi = mid;
    }
    const uint64_t q = dest[lo] + (p - begin[lo]);
    const uint32_t v = order[p];
    out_order[q] = v;
    out_perm[v] = (uint32_t)q;
}
constexpr uint32_t kMaxWorld = 64;
struct RangeTable {
    uint64_t hi[kMaxWorld];   // end positions of the ranks' ranges
    uint32_t world;
};
__device__ __forceinline__ uint32_t rank_of(const RangeTable& t, uint64_t pos) {
    uint32_t r = 0;
    while (r + 1u < t.world && pos >= t.hi[r]) r++;
    return r;
}
// per block: [rank] mass of the edges a shard generates events for (an end in its range), [world + rank] of those with ONE end in it
__global__ void __launch_bounds__(256) cross_mass_kernel(uint64_t n, uint32_t uniform_k, const uint64_t* __restrict__ indptr, const uint32_t* __restrict__ nbr,
                                                         const float* __restrict__ proba, const uint32_t* __restrict__ perm, RangeTable t, double* __restrict__ out) {
    __shared__ double s_m[2 * kMaxWorld];
    for (int x = threadIdx.x; x < 2 * (int)kMaxWorld; x += 256) s_m[x] = 0.;
    __syncthreads();
    for (uint64_t u = blockIdx.x * 256ull + threadIdx.x; u < n; u += (uint64_t)gridDim.x * 256ull) {
        uint64_t b, e;
        if (uniform_k) { b = u * uniform_k; e = b + uniform_k; } else { b = indptr[u]; e = indptr[u + 1]; }
        const uint32_t ru = rank_of(t, perm[u]);
        for (uint64_t x = b; x < e; x++) {
            const double w = proba ? (double)proba[x] : 1.0;
            const uint32_t rv = rank_of(t, perm[nbr[x]]);
            atomicAdd(&s_m[ru], w);
            if (rv != ru) {
                atomicAdd(&s_m[t.world + ru], w);
                atomicAdd(&s_m[rv], w);
                atomicAdd(&s_m[t.world + rv], w);
            }
        }
    }
    __syncthreads();
    for (int x = threadIdx.x; x < 2 * (int)t.world; x += 256) out[(uint64_t)blockIdx.x * 2 * t.world + x] = s_m[x];
}

// ---------------------------------------------------------------------------------------------
// the permuted graph
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) perm_row_len_kernel(uint64_t n, const uint64_t* __restrict__ indptr, const uint32_t* __restrict__ order, uint64_t* __restrict__ len) {
    const uint64_t p = blockIdx.x * 256ull + threadIdx.x;
    if (p < n) { const uint32_t v = order[p]; len[p] = indptr[v + 1] - indptr[v]; }
    if (p == n) len[p] = 0;
}
// row at position p = row of order[p], neighbour ids through perm; `extra` (edge-aligned, may be null) moves with the edges
__global__ void __launch_bounds__(256) perm_rows_kernel(uint64_t n, uint32_t uniform_k, const uint64_t* __restrict__ indptr, const uint64_t* __restrict__ indptr2,
                                                        const uint32_t* __restrict__ order, const uint32_t* __restrict__ perm, const uint32_t* __restrict__ nbr,
                                                        const float* __restrict__ dist, const float* __restrict__ extra, uint32_t* __restrict__ nbr2,
                                                        float* __restrict__ dist2, float* __restrict__ extra2) {
    const uint64_t p = blockIdx.x * 256ull + threadIdx.x;
    if (p >= n) return;
    const uint32_t v = order[p];
    uint64_t b, b2, len;
    if (uniform_k) { b = (uint64_t)v * uniform_k; b2 = p * uniform_k; len = uniform_k; }
    else { b = indptr[v]; len = indptr[v + 1] - b; b2 = indptr2[p]; }
    for (uint64_t m = 0; m < len; m++) {
        nbr2[b2 + m] = perm[nbr[b + m]];
        if (dist2) dist2[b2 + m] = dist[b + m];
        if (extra2) extra2[b2 + m] = extra[b + m];
    }
}
__global__ void __launch_bounds__(256) gather_rows_kernel(uint64_t n, uint32_t dim, const uint32_t* __restrict__ order, const float* __restrict__ src, float* __restrict__ dst,
                                                          int scatter) {
    for (uint64_t t = blockIdx.x * 256ull + threadIdx.x; t < n * dim; t += (uint64_t)gridDim.x * 256ull) {
        const uint64_t p = t / dim, q = t % dim;
        const uint64_t v = order[p];
        if (scatter) dst[v * dim + q] = src[p * dim + q];   // back to the caller's order
        else dst[p * dim + q] = src[v * dim + q];
    }
}

struct Piece {
    uint64_t begin, size;
};

}  // namespace

namespace ae {

void partition_nodes_device(const ae_kgraph* g, const float* d_proba, const float* d_y, uint32_t ydim, uint32_t ystride, uint32_t world, Partition& out) {
    const uint64_t n = g->n;
    if (world == 0 || world > kMaxWorld) fail(AE_ERR_INVALID_ARG, "partition: world must be in [1, %u]", kMaxWorld);
    if (n < world) fail(AE_ERR_INVALID_ARG, "partition: fewer nodes than ranks");
    if (d_y && (ydim == 0 || ydim > 64 || ystride < ydim)) fail(AE_ERR_INVALID_ARG, "partition: bad coordinate shape");
    const unsigned ngrid = blocks_for(n, 256);
    // 1. components
    DevBuf<uint32_t> label;
    label.alloc_pooled(n);
    hipLaunchKernelGGL(cc_init_kernel, dim3(ngrid), dim3(256), 0, stream(), n, label.p);
    DevBuf<unsigned long long> bad;
    bad.alloc_pooled(1);
    for (int pass = 0;; pass++) {
        hipLaunchKernelGGL(cc_hook_kernel, dim3(ngrid), dim3(256), 0, stream(), n, g->uniform_k, (const uint64_t*)g->indptr.p, (const uint32_t*)g->nbr.p, label.p);
        hipLaunchKernelGGL(cc_flatten_kernel, dim3(ngrid), dim3(256), 0, stream(), n, label.p);
        bad.zero();
        hipLaunchKernelGGL(cc_check_kernel, dim3(ngrid), dim3(256), 0, stream(), n, g->uniform_k, (const uint64_t*)g->indptr.p, (const uint32_t*)g->nbr.p,
                           (const uint32_t*)label.p, bad.p);
        check_launch("partition: components");
        unsigned long long hb = 0;
        bad.download(&hb, 1);
        if (!hb) break;
        if (pass >= 64) fail(AE_ERR_STATE, "partition: the connected components did not converge");
    }
    // sizes, and the components in ascending root order
    DevBuf<uint32_t> size, flag, idx;
    size.alloc_pooled(n); flag.alloc_pooled(n); idx.alloc_pooled(n + 1);
    size.zero();
    hipLaunchKernelGGL(cc_sizes_kernel, dim3(ngrid), dim3(256), 0, stream(), n, (const uint32_t*)label.p, size.p);
    hipLaunchKernelGGL(cc_root_flag_kernel, dim3(ngrid), dim3(256), 0, stream(), n, (const uint32_t*)label.p, flag.p);
    {
        size_t tmp_bytes = 0;
        if (rocprim::exclusive_scan(nullptr, tmp_bytes, flag.p, idx.p, 0u, n, rocprim::plus<uint32_t>(), stream()) != hipSuccess)
            fail(AE_ERR_NO_DEVICE, "rocprim exclusive_scan (size query) failed");
        DevBuf<char> tmp;
        tmp.alloc_pooled(tmp_bytes ? tmp_bytes : 1);
        if (rocprim::exclusive_scan(tmp.p, tmp_bytes, flag.p, idx.p, 0u, n, rocprim::plus<uint32_t>(), stream()) != hipSuccess)
            fail(AE_ERR_NO_DEVICE, "rocprim exclusive_scan failed");
        sync();
    }
    uint32_t last_idx = 0, last_flag = 0;
    AE_HIP(hipMemcpyAsync(&last_idx, idx.p + (n - 1), 4, hipMemcpyDeviceToHost, stream()));
    AE_HIP(hipMemcpyAsync(&last_flag, flag.p + (n - 1), 4, hipMemcpyDeviceToHost, stream()));
    sync();
    const uint64_t ncomp = (uint64_t)last_idx + last_flag;
    DevBuf<uint32_t> csize;
    csize.alloc_pooled(ncomp);
    hipLaunchKernelGGL(cc_compact_kernel, dim3(ngrid), dim3(256), 0, stream(), n, (const uint32_t*)label.p, (const uint32_t*)idx.p, (const uint32_t*)size.p, csize.p);
    const std::vector<uint32_t> hsize = csize.to_host();
    // nodes sorted by component (stable: ids ascending inside a component); roots ascend with the components
    DevBuf<uint32_t> ids, keys_out, order;
    ids.alloc_pooled(n); keys_out.alloc_pooled(n); order.alloc_pooled(n);
    hipLaunchKernelGGL(iota_kernel, dim3(ngrid), dim3(256), 0, stream(), n, ids.p);
    unsigned bits = 1;
    while (bits < 32 && (n >> bits)) bits++;
    sort_pairs_u32_u32(label.p, keys_out.p, ids.p, order.p, n, bits);
    check_launch("partition: order by component");
    sync();
    // 2. + 3. packing with splits
    std::vector<Piece> pieces(ncomp);
    {
        uint64_t at = 0;
        for (uint64_t c = 0; c < ncomp; c++) { pieces[c] = Piece{at, hsize[c]}; at += hsize[c]; }
        if (at != n) fail(AE_ERR_STATE, "partition: component sizes do not add up");
    }
    std::vector<std::vector<Piece>> by_rank(world);
    uint64_t splits = 0;
    DevBuf<double> d_stats;
    DevBuf<uint32_t> skeys, skeys2, svals, svals2;
    DevBuf<uint8_t> d_side;
    DevBuf<float> d_z, d_z2;
    DevBuf<long long> d_gain;
    DevBuf<unsigned long long> d_counts;
    auto split_piece = [&](const Piece& pc, uint64_t left_count, Piece& left, Piece& right) {
        // sort order[begin, begin + size) along the piece's widest axis; the first left_count nodes are the left piece
        if (pc.size > 1 && (d_y || !debug_knob("AE_PART_NO_SMOOTH"))) {
            if (!d_stats.n) d_stats.alloc_pooled((size_t)kStatBlocks * 2 * 64);
            if (!d_side.n) { d_side.alloc_pooled(n); d_side.zero(); d_gain.alloc_pooled(n); d_counts.alloc_pooled(4); }
            if (skeys.n < pc.size) { skeys.alloc_pooled(pc.size); skeys2.alloc_pooled(pc.size); svals.alloc_pooled(pc.size); svals2.alloc_pooled(pc.size); }
            const unsigned sg = (unsigned)std::min<uint64_t>(kStatBlocks, (pc.size + 255) / 256), pg = blocks_for(pc.size, 256);
            // per-column mean / standard deviation of `arr` (stride `st`, `nc` columns) over the piece, blocks added in order
            auto column_stats = [&](const float* arr, uint32_t nc, uint32_t st, std::vector<double>& mean, std::vector<double>& sd) {
                hipLaunchKernelGGL(piece_stats_kernel, dim3(sg), dim3(256), 0, stream(), (const uint32_t*)order.p, pc.begin, pc.begin + pc.size, arr, nc, st, d_stats.p);
                std::vector<double> hs((size_t)sg * 2 * nc);
                d_stats.download(hs.data(), hs.size());
                mean.assign(nc, 0.); sd.assign(nc, 0.);
                for (uint32_t a = 0; a < nc; a++) {
                    double sum = 0., q = 0.;
                    for (unsigned bk = 0; bk < sg; bk++) { sum += hs[(size_t)bk * 2 * nc + 2 * a]; q += hs[(size_t)bk * 2 * nc + 2 * a + 1]; }
                    mean[a] = sum / (double)pc.size;
                    sd[a] = std::sqrt(std::max(0., q / (double)pc.size - mean[a] * mean[a]));
                }
            };
            const float* key_src = d_y;
            uint32_t key_stride = ystride, axis = 0;
            std::vector<double> mean, sd;
            const int blocks = debug_knob("AE_PART_NO_SMOOTH") ? 0 : (debug_knob("AE_PART_SMOOTH") ? atoi(debug_knob("AE_PART_SMOOTH")) : (d_y ? 12 : 32));
            if (blocks > 0 && pc.size >= 64) {
                // blocks of 8 walk steps, normalised in between; the column that decays least over the last block is the smoothest
                if (!d_z.n) { d_z.alloc_pooled(n * kZ); d_z2.alloc_pooled(n * kZ); }
                hipLaunchKernelGGL(smooth_init_kernel, dim3(pg), dim3(256), 0, stream(), (const uint32_t*)order.p, pc.begin, pc.size, d_y, ydim, ystride, d_z.p, d_side.p);
                std::vector<double> keep(kZ, 0.);
                for (int blk = 0; blk < blocks; blk++) {
                    column_stats(d_z.p, kZ, kZ, mean, sd);
                    ZNorm nm;
                    for (uint32_t a = 0; a < kZ; a++) { nm.mean[a] = (float)mean[a]; nm.inv_std[a] = sd[a] > 1e-30 ? (float)(1.0 / sd[a]) : 0.f; }
                    hipLaunchKernelGGL(smooth_norm_kernel, dim3(pg), dim3(256), 0, stream(), (const uint32_t*)order.p, pc.begin, pc.size, nm, d_z.p);
                    for (int st = 0; st < 8; st++) {
                        hipLaunchKernelGGL(smooth_step_kernel, dim3(pg), dim3(256), 0, stream(), (const uint32_t*)order.p, pc.begin, pc.size, g->uniform_k,
                                           (const uint64_t*)g->indptr.p, (const uint32_t*)g->nbr.p, (const uint8_t*)d_side.p, (const float*)d_z.p, d_z2.p);
                        std::swap(d_z.p, d_z2.p);
                    }
                }
                column_stats(d_z.p, kZ, kZ, mean, sd);   // (every column entered the last block with deviation 1)
                double best = -1.;
                for (uint32_t a = 0; a < kZ; a++) if (sd[a] > best) { best = sd[a]; axis = a; }
                hipLaunchKernelGGL(refine_set_side_kernel, dim3(pg), dim3(256), 0, stream(), (const uint32_t*)order.p, pc.begin, pc.size, 0ull, d_side.p, d_gain.p, 1);
                key_src = d_z.p;
                key_stride = kZ;
            } else if (d_y) {   // the caller's coordinates as they are: the widest axis
                column_stats(d_y, ydim, ystride, mean, sd);
                double best = -1.;
                for (uint32_t a = 0; a < ydim; a++) if (sd[a] > best) { best = sd[a]; axis = a; }
            }
            if (key_src) {
                hipLaunchKernelGGL(piece_keys_kernel, dim3(pg), dim3(256), 0, stream(), (const uint32_t*)order.p, pc.begin, pc.size, key_src, axis, key_stride, skeys.p, svals.p);
                sort_pairs_u32_u32(skeys.p, skeys2.p, svals.p, svals2.p, pc.size, 32);
                AE_HIP(hipMemcpyAsync(order.p + pc.begin, svals2.p, sizeof(uint32_t) * pc.size, hipMemcpyDeviceToDevice, stream()));
            }
            check_launch("partition: bisection");
        }
        // the cut, smoothed on the graph (see the refine_* kernels); the sides keep their sizes to within a fraction of a per cent
        if (pc.size >= 64 && left_count >= 16 && pc.size - left_count >= 16 && !debug_knob("AE_PART_NO_REFINE")) {
            if (!d_side.n) { d_side.alloc_pooled(n); d_side.zero(); d_gain.alloc_pooled(n); d_counts.alloc_pooled(4); }
            if (skeys.n < pc.size) { skeys.alloc_pooled(pc.size); skeys2.alloc_pooled(pc.size); svals.alloc_pooled(pc.size); svals2.alloc_pooled(pc.size); }
            const unsigned pg = blocks_for(pc.size, 256);
            hipLaunchKernelGGL(refine_set_side_kernel, dim3(pg), dim3(256), 0, stream(), (const uint32_t*)order.p, pc.begin, pc.size, left_count, d_side.p, d_gain.p, 0);
            int64_t left_now = (int64_t)left_count;
            const int rounds_max = debug_knob("AE_PART_ROUNDS") ? atoi(debug_knob("AE_PART_ROUNDS")) : 96;
            for (int round = 0; round < rounds_max; round++) {
                const uint32_t round_key = 0x9E3779B9u * (uint32_t)(round + 1) + (uint32_t)splits * 0x85EBCA6Bu;
                d_counts.zero();
                hipLaunchKernelGGL(refine_gain_kernel, dim3(pg), dim3(256), 0, stream(), (const uint32_t*)order.p, pc.begin, pc.size, g->uniform_k, (const uint64_t*)g->indptr.p,
                                   (const uint32_t*)g->nbr.p, d_proba, (const uint8_t*)d_side.p, d_gain.p);
                hipLaunchKernelGGL(refine_count_kernel, dim3(pg), dim3(256), 0, stream(), (const uint32_t*)order.p, pc.begin, pc.size, (const uint8_t*)d_side.p,
                                   (const long long*)d_gain.p, round_key, d_counts.p);
                unsigned long long hc[4];
                d_counts.download(hc, 4);
                // how many may move each way: the same number, corrected by what the sides are off their target
                const int64_t off = left_now - (int64_t)left_count;   // > 0: the left side is too big
                double want_l = (double)std::min(hc[0], hc[1]), want_r = want_l;
                if (off > 0) want_l = std::min((double)hc[0], want_l + (double)off); else want_r = std::min((double)hc[1], want_r - (double)off);
                if (want_l + want_r < 1.0) break;
                auto thr = [](double want, unsigned long long have) { return have ? (uint32_t)std::min(4294967295.0, want / (double)have * 4294967296.0) : 0u; };
                hipLaunchKernelGGL(refine_move_kernel, dim3(pg), dim3(256), 0, stream(), (const uint32_t*)order.p, pc.begin, pc.size, d_side.p, d_gain.p, round_key,
                                   thr(want_l, hc[0]), thr(want_r, hc[1]), d_counts.p);
                d_counts.download(hc, 4);
                left_now += (int64_t)hc[3] - (int64_t)hc[2];
                if (hc[2] + hc[3] < std::max<uint64_t>(2, pc.size / 20000)) break;
            }
            hipLaunchKernelGGL(refine_keys_kernel, dim3(pg), dim3(256), 0, stream(), (const uint32_t*)order.p, pc.begin, pc.size, (const uint8_t*)d_side.p, skeys.p, svals.p);
            sort_pairs_u32_u32(skeys.p, skeys2.p, svals.p, svals2.p, pc.size, 1);
            AE_HIP(hipMemcpyAsync(order.p + pc.begin, svals2.p, sizeof(uint32_t) * pc.size, hipMemcpyDeviceToDevice, stream()));
            hipLaunchKernelGGL(refine_set_side_kernel, dim3(pg), dim3(256), 0, stream(), (const uint32_t*)order.p, pc.begin, pc.size, 0ull, d_side.p, d_gain.p, 1);
            check_launch("partition: refinement");
            sync();
            if (left_now > 0 && left_now < (int64_t)pc.size) left_count = (uint64_t)left_now;
        }
        left = Piece{pc.begin, left_count};
        right = Piece{pc.begin + left_count, pc.size - left_count};
        splits++;
    };
    constexpr double kTolerance = 0.015;   // a side may end this far off its target before a component is split for it
    std::function<void(std::vector<Piece>&, uint32_t, uint32_t)> assign = [&](std::vector<Piece>& ps, uint32_t r0, uint32_t r1) {
        if (r1 - r0 == 1) { by_rank[r0] = std::move(ps); return; }
        const uint32_t w = r1 - r0, wl = w / 2;
        uint64_t total = 0;
        for (const Piece& p : ps) total += p.size;
        const uint64_t target_l = total / w * wl + (total % w) * wl / w, target_r = total - target_l;
        std::stable_sort(ps.begin(), ps.end(), [](const Piece& a, const Piece& b) { return a.size > b.size; });
        std::vector<Piece> left, right;
        uint64_t fl = 0, fr = 0;
        for (const Piece& p : ps) {   // LPT: into the side with more room left
            const double room_l = (double)target_l - (double)fl, room_r = (double)target_r - (double)fr;
            if (room_l >= room_r) { left.push_back(p); fl += p.size; } else { right.push_back(p); fr += p.size; }
        }
        const double per_rank = (double)total / (double)w;
        auto rebalance = [&](std::vector<Piece>& from, std::vector<Piece>& to, uint64_t excess) {
            // move `excess` nodes: whole pieces that fit (smallest first would scatter; take the largest that fits), then ONE split
            while (excess > 0) {
                size_t pick = from.size();
                for (size_t x = 0; x < from.size(); x++)   // the largest piece: a cut through a big piece has the best surface-to-volume ratio
                    if (pick == from.size() || from[x].size > from[pick].size) pick = x;
                if (pick == from.size()) break;
                Piece pc = from[pick];
                from.erase(from.begin() + (long)pick);
                if (pc.size <= excess) { to.push_back(pc); excess -= pc.size; continue; }
                Piece keep, move;
                split_piece(pc, pc.size - excess, keep, move);
                from.push_back(keep);
                to.push_back(move);
                excess = 0;
            }
        };
        if (fl > target_l && (double)(fl - target_l) > kTolerance * per_rank) rebalance(left, right, fl - target_l);
        else if (fr > target_r && (double)(fr - target_r) > kTolerance * per_rank) rebalance(right, left, fr - target_r);
        assign(left, r0, r0 + wl);
        assign(right, r0 + wl, r1);
    };
    assign(pieces, 0, world);
    // every rank needs at least one node
    for (uint32_t r = 0; r < world; r++) {
        uint64_t s = 0;
        for (const Piece& p : by_rank[r]) s += p.size;
        if (!s) fail(AE_ERR_INVALID_ARG, "partition: rank %u would own no node (graph too small for %u ranks)", r, world);
    }
    // assemble: pieces by begin, destinations in rank order
    std::vector<uint64_t> hbegin, hdest;
    out.ranges.assign(2 * (size_t)world, 0);
    {
        struct Item { uint64_t begin, dest; };
        std::vector<Item> items;
        uint64_t at = 0;
        for (uint32_t r = 0; r < world; r++) {
            std::sort(by_rank[r].begin(), by_rank[r].end(), [](const Piece& a, const Piece& b) { return a.begin < b.begin; });
            out.ranges[2 * r] = at;
            for (const Piece& p : by_rank[r]) {
                if (!p.size) continue;
                items.push_back(Item{p.begin, at});
                at += p.size;
            }
            out.ranges[2 * r + 1] = at;
        }
        std::sort(items.begin(), items.end(), [](const Item& a, const Item& b) { return a.begin < b.begin; });
        hbegin.reserve(items.size()); hdest.reserve(items.size());
        for (const Item& it : items) { hbegin.push_back(it.begin); hdest.push_back(it.dest); }
        if (hbegin.empty() || hbegin[0] != 0 || at != n) fail(AE_ERR_STATE, "partition: the pieces do not tile the nodes");
    }
    if (hbegin.size() >= 0xFFFFFFFFull) fail(AE_ERR_INVALID_ARG, "partition: too many pieces");
    DevBuf<uint64_t> d_begin, d_dest;
    d_begin.alloc_pooled(hbegin.size()); d_dest.alloc_pooled(hdest.size());
    d_begin.upload(hbegin.data(), hbegin.size());
    d_dest.upload(hdest.data(), hdest.size());
    out.order.alloc(n);
    out.perm.alloc(n);
    hipLaunchKernelGGL(assemble_kernel, dim3(ngrid), dim3(256), 0, stream(), n, (const uint32_t*)order.p, (const uint64_t*)d_begin.p, (const uint64_t*)d_dest.p,
                       (uint32_t)hbegin.size(), out.order.p, out.perm.p);
    check_launch("partition: assemble");
    sync();
    out.components = ncomp;
    out.splits = splits;
    partition_cross_mass_device(g, d_proba, out);
    double biggest = 0.;
    for (uint32_t r = 0; r < world; r++) biggest = std::max(biggest, (double)(out.ranges[2 * r + 1] - out.ranges[2 * r]));
    out.imbalance = biggest / ((double)n / (double)world) - 1.0;
}

// cross-range edge mass of a partition (perm, ranges), shard by shard as the sharded time-sliced mode counts it (ce_slice.hip:
// sl_cross_frac = mass of the edges with ONE end in the shard / mass of the edges with an end in it)
void partition_cross_mass_device(const ae_kgraph* g, const float* d_proba, Partition& p) {
    const uint32_t world = (uint32_t)(p.ranges.size() / 2);
    RangeTable t;
    memset(&t, 0, sizeof(t));
    t.world = world;
    for (uint32_t r = 0; r < world; r++) t.hi[r] = p.ranges[2 * r + 1];
    const unsigned grid = grid_cap(g->n, 256, 1024);
    DevBuf<double> d_out;
    d_out.alloc_pooled((size_t)grid * 2 * world);
    hipLaunchKernelGGL(cross_mass_kernel, dim3(grid), dim3(256), 0, stream(), g->n, g->uniform_k, (const uint64_t*)g->indptr.p, (const uint32_t*)g->nbr.p, d_proba,
                       (const uint32_t*)p.perm.p, t, d_out.p);
    check_launch("partition: cross mass");
    const std::vector<double> h = d_out.to_host();
    std::vector<double> gen(world, 0.), cross(world, 0.);
    for (unsigned b = 0; b < grid; b++)
        for (uint32_t r = 0; r < world; r++) { gen[r] += h[(size_t)b * 2 * world + r]; cross[r] += h[(size_t)b * 2 * world + world + r]; }
    double tg = 0., tc = 0., worst = 0.;
    for (uint32_t r = 0; r < world; r++) {
        tg += gen[r] - 0.5 * cross[r];   // (a cross edge is counted by both its shards)
        tc += 0.5 * cross[r];
        if (gen[r] > 0.) worst = std::max(worst, cross[r] / gen[r]);
    }
    p.cross_mass = tg > 0. ? tc / tg : 0.;
    p.cross_mass_worst_rank = worst;
}

// the graph with the node at position p = caller's node order[p]; rows keep their (distance) order.  extra / extra2: an edge-aligned
// array that moves with the edges (the edge probabilities), or null.
ae_kgraph* kgraph_permuted_device(const ae_kgraph* g, const uint32_t* d_order, const uint32_t* d_perm, const float* d_extra, DevBuf<float>* extra2) {
    std::unique_ptr<ae_kgraph> g2(new ae_kgraph);
    const uint64_t n = g->n;
    g2->n = n;
    g2->max_nbng = g->max_nbng;
    g2->nnz = g->nnz;
    g2->uniform_k = g->uniform_k;
    g2->indptr.alloc(n + 1);
    g2->nbr.alloc(g->nnz);
    g2->dist.alloc(g->nnz);
    if (extra2) extra2->alloc(g->nnz);
    if (g->uniform_k) {
        AE_HIP(hipMemcpyAsync(g2->indptr.p, g->indptr.p, sizeof(uint64_t) * (n + 1), hipMemcpyDeviceToDevice, stream()));
    } else {
        DevBuf<uint64_t> len;
        len.alloc_pooled(n + 1);
        hipLaunchKernelGGL(perm_row_len_kernel, dim3(blocks_for(n + 1, 256)), dim3(256), 0, stream(), n, (const uint64_t*)g->indptr.p, d_order, len.p);
        size_t tmp_bytes = 0;
        if (rocprim::exclusive_scan(nullptr, tmp_bytes, len.p, g2->indptr.p, (uint64_t)0, n + 1, rocprim::plus<uint64_t>(), stream()) != hipSuccess)
            fail(AE_ERR_NO_DEVICE, "rocprim exclusive_scan (size query) failed");
        DevBuf<char> tmp;
        tmp.alloc_pooled(tmp_bytes ? tmp_bytes : 1);
        if (rocprim::exclusive_scan(tmp.p, tmp_bytes, len.p, g2->indptr.p, (uint64_t)0, n + 1, rocprim::plus<uint64_t>(), stream()) != hipSuccess)
            fail(AE_ERR_NO_DEVICE, "rocprim exclusive_scan failed");
        sync();
    }
    hipLaunchKernelGGL(perm_rows_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, stream(), n, g->uniform_k, (const uint64_t*)g->indptr.p, (const uint64_t*)g2->indptr.p,
                       d_order, d_perm, (const uint32_t*)g->nbr.p, (const float*)g->dist.p, d_extra, g2->nbr.p, g2->dist.p, extra2 ? extra2->p : nullptr);
    check_launch("kgraph_permuted");
    sync();
    return g2.release();
}

// rows of `dim` floats: gather (dst[p] = src[order[p]]) or scatter back (dst[order[p]] = src[p])
void permute_rows_device(const float* d_src, float* d_dst, uint64_t n, uint32_t dim, const uint32_t* d_order, bool back) {
    hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_cap(n * dim, 256, 1u << 20)), dim3(256), 0, stream(), n, dim, d_order, d_src, d_dst, back ? 1 : 0);
    check_launch("permute_rows");
}

}  // namespace ae

extern "C" {

int32_t ae_kgraph_partition(const ae_kgraph* g, const ae_node_params* np, const float* y, uint64_t dim, uint32_t world, uint32_t* order, uint64_t* ranges,
                            ae_partition_report* report) {
    return guard([&] {
        require_device();
        if (!g || !order || !ranges) fail(AE_ERR_INVALID_ARG, "null argument");
        if (np && np->g != g) fail(AE_ERR_INVALID_ARG, "node params were not computed from this graph");
        if (y && (dim == 0 || dim > 64)) fail(AE_ERR_INVALID_ARG, "dim must be in [1, 64]");
        DevBuf<float> dy;
        if (y) { dy.alloc(g->n * dim); dy.upload(y, g->n * dim); }
        Partition part;
        partition_nodes_device(g, np ? np->proba.p : nullptr, y ? dy.p : nullptr, (uint32_t)dim, (uint32_t)dim, world, part);
        part.order.download(order, g->n);
        for (size_t x = 0; x < part.ranges.size(); x++) ranges[x] = part.ranges[x];
        if (report) {
            report->components = part.components;
            report->splits = part.splits;
            report->cross_mass = part.cross_mass;
            report->cross_mass_worst_rank = part.cross_mass_worst_rank;
            report->imbalance = part.imbalance;
        }
    });
}

int32_t ae_kgraph_permuted(const ae_kgraph* g, const uint32_t* order, ae_kgraph** out) {
    return guard([&] {
        require_device();
        if (!g || !order || !out) fail(AE_ERR_INVALID_ARG, "null argument");
        std::vector<uint32_t> perm(g->n, 0xFFFFFFFFu);
        for (uint64_t p = 0; p < g->n; p++) {
            if (order[p] >= g->n || perm[order[p]] != 0xFFFFFFFFu) fail(AE_ERR_INVALID_ARG, "order is not a permutation of the node ids");
            perm[order[p]] = (uint32_t)p;
        }
        DevBuf<uint32_t> d_order, d_perm;
        d_order.alloc(g->n); d_perm.alloc(g->n);
        d_order.upload(order, g->n);
        d_perm.upload(perm.data(), g->n);
        sync();
        *out = kgraph_permuted_device(g, d_order.p, d_perm.p, nullptr, nullptr);
    });
}

}  // extern "C"
