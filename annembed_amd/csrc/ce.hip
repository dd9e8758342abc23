// ce.hip -- a11-a13: EntropyOptim (src/embedder.rs:936-1345) on device.
//
//   ce_sgd_hogwild_kernel    gradient_iteration_threaded (:1311-1315): one thread per SGD sample,
//                            lock-free in-place updates of the coordinate array (the reference's
//                            rayon loop is Hogwild too, :1197).
//   ce_plan_kernel + ce_sgd_planned_kernel
//                            AE_CE_SEQUENTIAL: the node set of every sample depends only on the
//                            graph and the RNG stream, so it is drawn first (plan), the host derives
//                            a conflict-free level schedule that is equivalent to executing samples
//                            0,1,2,... one after the other (gradient_iteration, :1305-1309), and each
//                            level is one launch.  All arithmetic stays on the GPU.
//   ce_value_kernel          ce_compute_threaded (:1127-1163).
//
// Per-sample arithmetic follows ce_optim_edge_shannon (:1167-1302) operation by operation:
// coordinates f32, scalars f64, no FMA contraction, so that with b = 1 the sequential mode is
// bit-exact against the CPU oracle.
#include <algorithm>

#include "ce_internal.h"
#include "ce_sample_math.h"
#include "linalg.h"
#include <chrono>
#include "philox.h"

using namespace ae;

namespace ae {
void sort_pairs_u32_u32(uint32_t* d_keys_in, uint32_t* d_keys_out, uint32_t* d_vals_in, uint32_t* d_vals_out, uint64_t count, unsigned end_bit);
void rowptr_from_sorted_keys(const uint64_t* d_keys, uint64_t nnz, uint64_t nrows, uint64_t* d_rowptr);
}  // namespace ae

#pragma clang fp contract(off)

namespace {


struct Plan {
    uint32_t i, j, k[5];
    float w;
};

__device__ __forceinline__ void row_bounds(const CeDev& c, uint32_t i, uint64_t& b, uint32_t& len) {
    if (c.uniform_k) {
        b = (uint64_t)i * c.uniform_k;
        len = c.uniform_k;
    } else {
        b = c.indptr[i];
        len = (uint32_t)(c.indptr[i + 1] - b);
    }
}

// draws the positive edge and the 5 accepted negatives of sample s (embedder.rs:1182-1184, 1241-1253)
__device__ __forceinline__ bool make_plan(const CeDev& c, uint64_t s, uint32_t iter, Plan& p) {
    PhiloxStream st(c.seed, s, iter);
    uint64_t e;
    uint64_t ib;
    uint32_t ilen;
    if (c.sampler == AE_SAMPLER_ROWCDF) {
        uint32_t i = (uint32_t)(c.node_lo + st.index(c.node_hi - c.node_lo));
        float u = st.f32();
        row_bounds(c, i, ib, ilen);
        uint32_t m = ilen - 1;
        float acc = 0.f;
        for (uint32_t t = 0; t < ilen; t++) {
            acc += c.proba[ib + t];
            if (u < acc) { m = t; break; }
        }
        e = ib + m;
        p.i = i;
    } else {
        uint64_t x = st.index(c.shard_edges);
        float u = st.f32();
        if (!(u < c.edge_odds[x])) x = c.edge_alias[x];
        e = c.edge_lo + x;
        p.i = c.edge_src[x];
        row_bounds(c, p.i, ib, ilen);
    }
    p.j = c.nbr[e];
    p.w = c.proba[e];
    int got = 0;
    uint32_t attempts = 0;
    while (got < 5) {
        uint32_t k;
        if (c.hub_odds) {  // NodeSampler::sample, :927-930
            uint64_t x = st.index(c.n);
            float u = st.f32();
            const uint2 he = c.hub_tab[x];
            k = (u < __uint_as_float(he.x)) ? (uint32_t)x : he.y;
        } else {
            k = (uint32_t)st.index(c.n);  // :1121
        }
        if (++attempts > (1u << 20)) return false;
        bool reject = (k == p.i) || (k == p.j);
        if (!reject)
            for (uint32_t t = 0; t < ilen; t++)  // NodeParam::get_edge linear scan, nodeparam.rs:83-85
                if (c.nbr[ib + t] == k) { reject = true; break; }
        if (reject) continue;  // :1246-1253
        p.k[got++] = k;
    }
    return true;
}

// The same draws with the neighbour row of i held in registers (rows of at most KMAX entries): the row scan of the
// CDF sampler and the five rejection scans become branch-free passes over registers, their loads are issued together
// instead of one dependent load per loop trip.  Results are those of make_plan: the cumulative sums are formed in the
// same order, and "first t with u < acc_t" = "number of t with !(u < acc_t)" because acc is non-decreasing.
template <int KMAX>
__device__ __forceinline__ bool make_plan_rows(const CeDev& c, uint64_t s, uint32_t iter, Plan& p) {
    PhiloxStream st(c.seed, s, iter);
    uint64_t e, ib, x = 0;
    uint32_t ilen;
    const bool rowcdf = c.sampler == AE_SAMPLER_ROWCDF;
    if (rowcdf) p.i = (uint32_t)(c.node_lo + st.index(c.node_hi - c.node_lo));
    else x = st.index(c.shard_edges);
    const float u = st.f32();
    if (!rowcdf) {
        if (!(u < c.edge_odds[x])) x = c.edge_alias[x];
        p.i = c.edge_src[x];
    }
    row_bounds(c, p.i, ib, ilen);
    uint32_t nb[KMAX];
    float pr[KMAX];
    // The row and its KMAX - ilen successors in ONE unconditional read (wide loads; entries beyond the row masked afterwards): under
    // `t < ilen ? load : sentinel` every entry is a load instruction of its own, 2 KMAX address-unit passes per sample -- what bound
    // the kernel.  Only the rows at the very end of the arrays (where the read would leave them) take the entry-wise form.
    if (ib + KMAX <= c.nnz) {
#pragma unroll
        for (int t = 0; t < KMAX; t++) nb[t] = c.nbr[ib + t];
        if (rowcdf) {
#pragma unroll
            for (int t = 0; t < KMAX; t++) pr[t] = c.proba[ib + t];
        }
#pragma unroll
        for (int t = 0; t < KMAX; t++) {
            nb[t] = (uint32_t)t < ilen ? nb[t] : 0xFFFFFFFFu;  // node ids are < n < 2^32 - 1
            if (rowcdf) pr[t] = (uint32_t)t < ilen ? pr[t] : 0.f;
        }
    } else {
#pragma unroll
        for (int t = 0; t < KMAX; t++) nb[t] = (uint32_t)t < ilen ? c.nbr[ib + t] : 0xFFFFFFFFu;
        if (rowcdf) {
#pragma unroll
            for (int t = 0; t < KMAX; t++) pr[t] = (uint32_t)t < ilen ? c.proba[ib + t] : 0.f;
        }
    }
    if (rowcdf) {
        float acc = 0.f;
        uint32_t m = 0;
#pragma unroll
        for (int t = 0; t < KMAX; t++) {
            acc += pr[t];
            m += ((uint32_t)t < ilen && !(u < acc)) ? 1u : 0u;
        }
        e = ib + min(m, ilen - 1);
    } else {
        e = c.edge_lo + x;
    }
    p.j = c.nbr[e];
    p.w = c.proba[e];
    int got = 0;
    uint32_t attempts = 0;
    // (Looking the first six attempts' alias-table entries up together -- the stream is the sample's own, unused words cost nothing --
    // was built and is bit-identical, but buys nothing where it would matter: with hubness weighting on a graph beyond the L2s the
    // planner is bound by its ~10 random requests per sample, 16 ms per 100 M samples at 1.65 M nodes either way.)
    while (got < 5) {
        uint32_t k;
        if (c.hub_odds) {
            uint64_t xx = st.index(c.n);
            float uu = st.f32();
            const uint2 he = c.hub_tab[xx];
            k = (uu < __uint_as_float(he.x)) ? (uint32_t)xx : he.y;
        } else {
            k = (uint32_t)st.index(c.n);
        }
        if (++attempts > (1u << 20)) return false;
        bool reject = (k == p.i) || (k == p.j);
#pragma unroll
        for (int t = 0; t < KMAX; t++) reject |= nb[t] == k;
        if (reject) continue;
        // static indexing keeps p.k in registers
#pragma unroll
        for (int g = 0; g < 5; g++) p.k[g] = got == g ? k : p.k[g];
        got++;
    }
    return true;
}

template <int DIM>
struct Row {
    float v[DIM];
};

// the whole sample: yi / yj are updated in place (the values the reference stores at :1301 / :1239)
template <int DIM>
__device__ __forceinline__ void sample_update(float* yi, float* yj, const float (*yk)[DIM], float w, double scale, double b, double grad_step) {
    float grad[DIM];
    sample_attract<DIM>(yi, yj, grad, w, scale, b, grad_step);
#pragma unroll
    for (int g = 0; g < 5; g++) sample_repulse<DIM>(yi, yk[g], grad, scale, b, grad_step);  // :1244-1299
}

// in-place form: rows read from / written to the coordinate array (the negatives are never i or j, :1246-1253, so
// reading them before the stores equals the reference's order)
template <int DIM>
__device__ __forceinline__ void apply_sample(const CeDev& c, const Plan& p, double grad_step) {
    float yi[DIM], yj[DIM], yk[5][DIM];
    load_row<DIM>(c.y, p.i, yi);  // :1185
    load_row<DIM>(c.y, p.j, yj);  // :1186
#pragma unroll
    for (int g = 0; g < 5; g++) load_row<DIM>(c.y, p.k[g], yk[g]);
    sample_update<DIM>(yi, yj, yk, p.w, (double)c.emb_scale[p.i], c.b, grad_step);
    store_row<DIM>(c.y, p.j, yj);  // :1239
    store_row<DIM>(c.y, p.i, yi);  // :1301
}

template <int DIM>
__global__ void __launch_bounds__(256) ce_sgd_hogwild_kernel(CeDev c, uint64_t s_begin, uint64_t nb_sample, double grad_step,
                                                             uint32_t iter, unsigned int* err) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t s = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; s < nb_sample; s += stride) {
        Plan p;
        if (!make_plan(c, s_begin + s, iter, p)) { atomicOr(err, 1u); continue; }
        apply_sample<DIM>(c, p, grad_step);
    }
}

template <int KMAX>  // 0: rows of any length (make_plan)
__global__ void __launch_bounds__(256) ce_plan_kernel(CeDev c, uint64_t s_begin, uint64_t nb_sample, uint32_t iter,
                                                      uint32_t* __restrict__ plan_nodes, float* __restrict__ plan_w,
                                                      unsigned int* err) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t s = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; s < nb_sample; s += stride) {
        Plan p;
        bool ok;
        if constexpr (KMAX == 0) ok = make_plan(c, s_begin + s, iter, p);
        else ok = make_plan_rows<KMAX>(c, s_begin + s, iter, p);
        if (!ok) { atomicOr(err, 1u); p.i = p.j = 0; for (int g = 0; g < 5; g++) p.k[g] = 0; p.w = 0.f; }
        uint32_t* o = plan_nodes + s * 7;
        o[0] = p.i; o[1] = p.j;
        for (int g = 0; g < 5; g++) o[2 + g] = p.k[g];
        plan_w[s] = p.w;
    }
}
static void launch_plan(const ae_entropy_optim* o, uint64_t s_first, uint64_t S, uint32_t iter, uint32_t* nodes, float* w) {
    const uint32_t kmax = o->g->max_nbng;
    const dim3 grid(grid_cap(S, 256)), block(256);
#define AE_PLAN(K) hipLaunchKernelGGL(ce_plan_kernel<K>, grid, block, 0, stream(), o->dev, s_first, S, iter, nodes, w, o->err.p)
    if (kmax <= 16) AE_PLAN(16);
    else if (kmax <= 32) AE_PLAN(32);
    else AE_PLAN(0);
#undef AE_PLAN
    check_launch("ce_plan");
}

template <int DIM>
__global__ void __launch_bounds__(256) ce_sgd_planned_kernel(CeDev c, const uint32_t* __restrict__ order, uint64_t count,
                                                             const uint32_t* __restrict__ plan_nodes,
                                                             const float* __restrict__ plan_w, double grad_step) {
    uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (t >= count) return;
    const uint64_t s = order[t];
    Plan p;
    const uint32_t* o = plan_nodes + s * 7;
    p.i = o[0]; p.j = o[1];
    for (int g = 0; g < 5; g++) p.k[g] = o[2 + g];
    p.w = plan_w[s];
    apply_sample<DIM>(c, p, grad_step);
}

// ---------------------------------------------------------------------------------------------------------------
// AE_CE_SEQUENTIAL, device-scheduled ("dataflow") form.  Executing samples 0, 1, 2, ... in order is equivalent to:
// every sample reads, for each of its 7 nodes, the row produced by the LAST EARLIER sample that wrote that node (or the
// batch's initial row), and publishes its two new rows (y_i, y_j) as new versions.  With versions instead of in-place
// stores there are no write-after-read or write-after-write hazards, only the true dependencies remain (C2: depth 4 323
// instead of 7 191 levels).  So:
//   1. ce_plan_kernel draws the node sets (as before);
//   2. the 2 S write events (node, sample, slot) are radix-sorted, a CSR over nodes is built, and every sample finds
//      its 7 predecessors by binary search in its nodes' write lists -- all parallel, nothing on the host;
//   3. a persistent grid runs the samples in index order: a lane polls the version rows of its predecessors (a row is
//      published by its own stores, see df_try_load_version), takes each as soon as it exists, applies the sample
//      (same f64 arithmetic as the in-place form: bit-exact) and publishes its two rows.  Every sample only waits for
//      smaller sample indices and all lanes are resident (cooperative launch), so the smallest unfinished sample can
//      always run: no deadlock; a poll budget turns any violation of that argument into an error instead of a hang;
//   4. the last version of every written node is copied back into the coordinate array.
constexpr uint32_t kNoPred = 0xFFFFFFFFu;

// write events in sample order: key = node, value = version id (sample << 1 | slot).  A stable sort on the node bits
// alone then lists every node's versions in increasing order (3 radix passes instead of the 7 a 52-bit key needs).
__global__ void df_write_keys_kernel(uint64_t S, const uint32_t* __restrict__ plan_nodes, uint32_t* __restrict__ keys,
                                     uint32_t* __restrict__ vals) {
    const uint64_t s = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (s >= S) return;
    *reinterpret_cast<uint2*>(keys + 2 * s) = make_uint2(plan_nodes[s * 7], plan_nodes[s * 7 + 1]);
    *reinterpret_cast<uint2*>(vals + 2 * s) = make_uint2((uint32_t)(s << 1), (uint32_t)(s << 1) | 1u);
}
// rowptr[x] = first sorted position whose node >= x
__global__ void df_rowptr_kernel(const uint32_t* __restrict__ keys, uint64_t nnz, uint64_t n, uint64_t* __restrict__ rowptr) {
    const uint64_t x = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (x > n) return;
    uint64_t lo = 0, hi = nnz;
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if (keys[mid] < x) lo = mid + 1;
        else hi = mid;
    }
    rowptr[x] = lo;
}

// pred[s * 7 + t] = (sample << 1 | slot) of the last write of node plan_nodes[s * 7 + t] by a sample < s, or kNoPred.
// Slots 0 and 1 (the end points, which are the writes themselves): the predecessor of a write is its left neighbour in the
// node-sorted list -- one coalesced pass over the 2 S sorted events, no search.
__global__ void df_pred_writes_kernel(uint64_t S, const uint32_t* __restrict__ keys, const uint32_t* __restrict__ vals, uint32_t* __restrict__ pred) {
    for (uint64_t p = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; p < 2 * S; p += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t node = keys[p], v = vals[p];
        uint32_t out = kNoPred;
        if (p > 0 && keys[p - 1] == node) {
            out = vals[p - 1];
            if ((out >> 1) == (v >> 1)) out = (p > 1 && keys[p - 2] == node) ? vals[p - 2] : kNoPred;  // both ends of one sample on one node
        }
        pred[(uint64_t)(v >> 1) * 7 + (v & 1u)] = out;
    }
}
// Slots 2..6 (the negatives, read only).  A node's writers are spread evenly over the batch (i.i.d. samples): the answer lies
// within a few entries (sigma <= sqrt(len) / 2) of the interpolated position, so one 64-byte window around it settles most
// searches in a single memory round trip; the rest gallop from the window's edge and finish by bisection.
__global__ void df_pred_reads_kernel(uint64_t S, const uint32_t* __restrict__ plan_nodes, const uint32_t* __restrict__ vals,
                                     const uint64_t* __restrict__ rowptr, uint32_t* __restrict__ pred) {
    // grid-stride: the work items exceed the 2^32 a dispatch can carry from S = 8.6e8 on
    for (uint64_t it = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; it < S * 5; it += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t s = (uint32_t)(it / 5);
        const uint64_t idx = (uint64_t)s * 7 + 2 + (it - (uint64_t)s * 5);
        const uint32_t x = plan_nodes[idx];
        const uint64_t lo = rowptr[x], hi = rowptr[x + 1];
        if (lo == hi) { pred[idx] = kNoPred; continue; }
        const uint32_t key = s << 1;  // first version with sample >= s: vals < key  <=>  its sample < s
        uint64_t g = lo + (uint64_t)((float)(hi - lo) * ((float)s / (float)S));
        g = min(g, hi - 1);
        // 16 entries around g, the window aligned to 16 bytes (entries outside [lo, hi) belong to other nodes: masked; the
        // buffer is padded on both sides)
        int64_t b = (int64_t)g - 8;
        b -= (int64_t)((reinterpret_cast<uintptr_t>(vals + b) >> 2) & 3u);
        uint4 w4[4];
#pragma unroll
        for (int r = 0; r < 4; r++) w4[r] = *reinterpret_cast<const uint4*>(vals + b + 4 * r);
        const uint32_t w[16] = {w4[0].x, w4[0].y, w4[0].z, w4[0].w, w4[1].x, w4[1].y, w4[1].z, w4[1].w,
                                w4[2].x, w4[2].y, w4[2].z, w4[2].w, w4[3].x, w4[3].y, w4[3].z, w4[3].w};
        uint32_t cnt = 0, nvalid = 0;
#pragma unroll
        for (int e = 0; e < 16; e++) {
            const int64_t q = b + e;
            const bool valid = q >= (int64_t)lo && q < (int64_t)hi;
            nvalid += valid ? 1u : 0u;
            cnt += (valid && w[e] < key) ? 1u : 0u;
        }
        const uint64_t first = (uint64_t)max(b, (int64_t)lo), end = min((uint64_t)(b + 16), hi);
        uint64_t L, H;  // every index < L holds a value < key, every index >= H a value >= key
        if (cnt == 0 && first > lo) {  // everything in the window >= key: the answer lies at or before its first entry
            H = first;
            L = lo;
            for (uint64_t step = 16; H >= lo + step; step *= 2) {
                const uint64_t q = H - step;
                if (vals[q] < key) { L = q + 1; break; }
                H = q;
            }
        } else if (cnt == nvalid && end < hi) {  // everything < key: at or after its end
            L = end;
            H = hi;
            for (uint64_t step = 16; L + step < hi; step *= 2) {
                const uint64_t q = L + step;
                if (vals[q] < key) L = q + 1;
                else { H = q; break; }
            }
        } else {
            L = H = first + cnt;
        }
        while (L < H) {
            const uint64_t mid = (L + H) >> 1;
            if (vals[mid] < key) L = mid + 1;
            else H = mid;
        }
        pred[idx] = L > lo ? vals[L - 1] : kNoPred;
    }
}

// A version row is published by its stores alone: the buffer is filled with an all-ones pattern (a NaN no arithmetic
// produces: hardware NaNs are the canonical 0x7FC00000) before the kernel, a reader polls the row itself and takes it
// once every part differs from the pattern -- one memory round trip per dependency hop instead of flag + data.
// RELAXED (AE_CE_ORDERED): only the two END POINTS of a sample are dependencies -- the attraction is applied to the rows the previous
// writers of i and j published, exactly as in the sequential order -- while the five negatives are read as the memory system has
// them: every sample also stores its new rows IN PLACE (write-through), and a negative is one coherent load of that array, never a
// wait.  That is what the reference's own threaded loop guarantees (rows under a lock for the update, negatives through try_read,
// embedder.rs:1257-1265); the dependency depth of a C2 batch drops from 4 305 levels to 1 565 (tools/dependency_depth.py).
template <int DIM, bool RELAXED>
__global__ void __launch_bounds__(256) ce_dataflow_kernel(CeDev c, uint64_t S, const uint32_t* __restrict__ plan_nodes,
                                                          const float* __restrict__ plan_w, const uint32_t* __restrict__ pred,
                                                          float* __restrict__ ver, double grad_step, unsigned int* __restrict__ err, uint32_t lane_stride_arg) {
    // (bit 31 of the stride argument: the relaxed form computes its repulsion coefficients in f32)
    const uint32_t lane_stride = lane_stride_arg & 0x7FFFFFFFu;
    const bool f32_repulsion = (lane_stride_arg >> 31) != 0u;
    (void)f32_repulsion;
    // rows of more than 16 columns (asked_dim 17 ... 64): only the two end points are held in registers, a negative's row is
    // taken when its repulsion is due (7 x 64 registers do not exist); same arithmetic, same order
    constexpr bool WIDE = DIM > 16 || RELAXED;
    constexpr int NR = WIDE ? 2 : 7;
    // only every lane_stride-th lane carries samples: a wave's trip through the loop below costs the poll round trip plus
    // the arithmetic of whichever lanes advance, and a blocked lane moves once per trip -- fewer passengers, shorter trips
    const uint64_t gtid = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    const bool carrier = gtid % lane_stride == 0;
    const uint64_t nthreads = ((uint64_t)gridDim.x * blockDim.x) / lane_stride;
    const uint64_t tid = gtid / lane_stride;
    const uint64_t sweeps = (S + nthreads - 1) / nthreads;
    for (uint64_t k = 0; k < sweeps; k++) {  // wave-uniform trip count; sample indices increase with k
        const uint64_t s = k * nthreads + tid;
        bool finished = !carrier || s >= S;
        uint32_t node[7], pr[7];
        float rows[NR][DIM];
        float w = 0.f;
        uint32_t pending = 0;  // bit t: row t not gathered yet
        if (!finished) {
#pragma unroll
            for (int t = 0; t < 7; t++) {
                node[t] = plan_nodes[s * 7 + t];
                pr[t] = (RELAXED && t >= 2) ? kNoPred : pred[s * 7 + t];
                if (t < NR) {
                    if (pr[t] == kNoPred) {  // the batch's initial row (exact mode: c.y is read-only here; relaxed: nobody has written this row yet)
                        if constexpr (RELAXED) load_row_coherent<DIM>(c.y, node[t], rows[t]);
                        else load_row<DIM>(c.y, node[t], rows[t]);
                    } else {
                        pending |= 1u << t;
                    }
                }
            }
            w = plan_w[s];
        }
        // The sample advances step by step as its rows arrive: the attraction needs rows 0 and 1 and publishes y_j at once
        // (a successor that only reads y_j does not wait for the five repulsions), repulsion g needs row 1 + g; y_i is
        // published after the last one.  A dependency therefore delays only the steps that really use it.
        uint32_t polls = 0;
        int stage = 0;  // 0: attraction pending; 1..5: repulsion `stage` pending; 6: done
        float grad[DIM];
        float negs[(RELAXED && DIM <= 16) ? 5 : 1][DIM];  // relaxed form: the negatives' rows as of the current trip
        double scale = 1.;
        if (!finished) scale = (double)c.emb_scale[node[0]];
        // (Round 3, read off the ISA: under `if (pending bit)` every slot's poll is load - s_waitcnt vmcnt(0) - use, up to NR round trips
        // in a row per trip, and the per-sample set-up above is six dependent round trips.  Issuing all polls of a trip together
        // (unconditionally, a dropped result for slots that are not pending) and the set-up in two batches was built and measured: C2
        // ordered launch 4.42 against 4.48 ms, exact 8.87 against 9.15 -- nothing -- and the C3 shape, which is throughput-bound, LOST
        // 8-15 % to the extra requests (ordered 41.1 -> 44.4 ms, exact 68.1 -> 78.2).  Skipping the negatives' re-reads while both end
        // points are outstanding: 4.61 against 4.64 ms.  A level of the chain costs the store's way to the memory side plus the poll
        // that finds it, whatever else the trip does; the conditional form stays.)
        while (!__all(finished)) {
            if (!finished) {
#pragma unroll
                for (int t = 0; t < NR; t++) {
                    if ((pending >> t) & 1u) {
                        float tmp[DIM];
                        if (df_try_load_version<DIM>(ver, pr[t], tmp)) {
#pragma unroll
                            for (int q = 0; q < DIM; q++) rows[t][q] = tmp[q];
                            pending &= ~(1u << t);
                        }
                    }
                }
                if constexpr (RELAXED && DIM <= 16) {
                    // the negatives as the memory system has them NOW, re-read on every trip of a waiting sample (they travel with the
                    // polls: no round trip of their own): when the end points arrive the whole sample completes in that trip
#pragma unroll
                    for (int g = 0; g < 5; g++) load_row_coherent<DIM>(c.y, node[2 + g], negs[g]);
                }
                if (stage == 0 && (pending & 3u) == 0u) {
                    // (the ordered form: f64 scalars with one division per interaction, ce_sample_math.h; the exact form: the reference's
                    // operation order, bit for bit)
                    if constexpr (RELAXED) attract_f64<DIM>(rows[0], rows[1], grad, w, rcp_f64(scale * scale), c.b, grad_step);
                    else sample_attract<DIM>(rows[0], rows[1], grad, w, scale, c.b, grad_step);
                    if constexpr (RELAXED) df_store_version<DIM>(c.y, node[1], rows[1]);  // in place, for the negatives of others (:1239)
                    df_store_version<DIM>(ver, s * 2 + 1, rows[1]);
                    stage = 1;
                }
                if constexpr (!WIDE) {
                    // one code path for the five repulsions: the row is selected by the stage, lanes leave when they block
                    while (stage >= 1 && stage <= 5 && ((pending >> (1 + stage)) & 1u) == 0u) {
                        float yk[DIM];
#pragma unroll
                        for (int q = 0; q < DIM; q++) {
                            float v = rows[NR > 2 ? 2 : 0][q];
#pragma unroll
                            for (int g = 2; g <= 5; g++) v = stage == g ? rows[NR > 2 ? 1 + g : 0][q] : v;
                            yk[q] = v;
                        }
                        sample_repulse<DIM>(rows[0], yk, grad, scale, c.b, grad_step);
                        stage++;
                    }
                } else if constexpr (RELAXED && DIM <= 16) {
                    if (stage == 1) {  // (the negatives were read in this very trip, next to the polls: see below)
                        if (f32_repulsion) {
                            const float sf = (float)scale, inv_s2 = __builtin_amdgcn_rcpf(sf * sf);
#pragma unroll
                            for (int g = 0; g < 5; g++) sample_repulse_f32<DIM>(rows[0], negs[g], grad, inv_s2, (float)c.b, (float)grad_step);
                        } else {
                            const double inv_s2 = rcp_f64(scale * scale);
#pragma unroll
                            for (int g = 0; g < 5; g++) repulse_f64<DIM>(rows[0], negs[g], grad, inv_s2, c.b, grad_step);
                        }
                        stage = 6;
                    }
                } else {
                    while (stage >= 1 && stage <= 5) {
                        uint32_t nk = node[2], pk = pr[2];
#pragma unroll
                        for (int g = 2; g <= 5; g++) { nk = stage == g ? node[1 + g] : nk; pk = stage == g ? pr[1 + g] : pk; }
                        float yk[DIM];
                        if constexpr (RELAXED) load_row_coherent<DIM>(c.y, nk, yk);  // as the memory system has it now
                        else if (pk == kNoPred) load_row<DIM>(c.y, nk, yk);
                        else if (!df_try_load_version<DIM>(ver, pk, yk)) break;  // not published yet: next trip
                        if constexpr (RELAXED) repulse_f64<DIM>(rows[0], yk, grad, rcp_f64(scale * scale), c.b, grad_step);
                        else sample_repulse<DIM>(rows[0], yk, grad, scale, c.b, grad_step);
                        stage++;
                    }
                }
                if (stage == 6) {
                    if constexpr (RELAXED) df_store_version<DIM>(c.y, node[0], rows[0]);  // :1301
                    df_store_version<DIM>(ver, s * 2, rows[0]);
                    finished = true;
                } else if (++polls > (1u << 24)) {  // cannot happen (see above): fail instead of hanging
                    atomicOr(err, 8u);
                    float zero[DIM];
#pragma unroll
                    for (int q = 0; q < DIM; q++) zero[q] = 0.f;
                    df_store_version<DIM>(ver, s * 2, zero);  // unblock the successors; the host reports the error
                    df_store_version<DIM>(ver, s * 2 + 1, zero);
                    finished = true;
                }
            }
            if (!__all(finished)) __builtin_amdgcn_s_sleep(1);
        }
    }
}

// the last version of every node written in the batch becomes its row in the coordinate array
template <int DIM>
__global__ void df_commit_kernel(uint64_t n, const uint64_t* __restrict__ rowptr, const uint32_t* __restrict__ vals,
                                 const float* __restrict__ ver, float* __restrict__ y, const unsigned int* __restrict__ err) {
    const uint64_t x = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (*err & 8u) return;  // the dataflow kernel gave up (poll budget): its rows are not results, the coordinates stay as they were
    if (x >= n || rowptr[x + 1] == rowptr[x]) return;
    const uint32_t pv = vals[rowptr[x + 1] - 1];
#pragma unroll
    for (int t = 0; t < DIM; t++) y[x * DIM + t] = ver[(uint64_t)pv * DIM + t];
}

// cauchy_edge_weight (:1322-1345) + ce_compute_threaded (:1127-1163): per-block partial sums in f64
__global__ void __launch_bounds__(256) ce_value_kernel(CeDev c, double* __restrict__ partial) {
    __shared__ double red[256];
    double local = 0.;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = c.node_lo + blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < c.node_hi; i += stride) {
        const double scale = (double)c.emb_scale[i];
        uint64_t b;
        uint32_t len;
        row_bounds(c, (uint32_t)i, b, len);
        for (uint32_t m = 0; m < len; m++) {
            const float* a = c.y + i * c.dim;
            const float* o = c.y + (uint64_t)c.nbr[b + m] * c.dim;
            float acc = 0.f;
            for (uint32_t t = 0; t < c.dim; t++) { float df = a[t] - o[t]; acc += df * df; }  // :1326-1330
            double d = (double)acc / (scale * scale);  // :1331
            d = pow(d, c.b);                           // :1333
            const double weight = 1. / (1. + d);       // :1336
            float wf = (float)weight;                  // :1337
            if (!(wf < 1.0f)) wf = 1.0f - 1.1920929e-07f;  // :1338-1341
            const double we = (double)wf;
            const double wij = (double)c.proba[b + m];
            double term = 0.;
            if (we > 0.) term += -wij * log(we);              // :1150-1152
            if (we < 1.) term += -(1. - wij) * log(1. - we);  // :1153-1155
            local += term;
        }
    }
    red[threadIdx.x] = local;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

// estimate_embedded_scales_from_initial_scales, embedder.rs:1356-1373 (mean computed by the host
// driver with the reference's sequential f32 sum)
__global__ void embedded_scales_kernel(uint64_t n, const float* __restrict__ scale, float mean_scale, float* __restrict__ out) {
    uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = 0.2f * fmaxf(fminf(scale[i] / mean_scale, 4.0f), 0.25f);
}

__global__ void edge_src_kernel(uint64_t node_lo, uint64_t node_hi, const uint64_t* __restrict__ indptr, uint64_t edge_lo,
                                uint32_t* __restrict__ src) {
    uint64_t i = node_lo + blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i >= node_hi) return;
    for (uint64_t e = indptr[i]; e < indptr[i + 1]; e++) src[e - edge_lo] = (uint32_t)i;
}

}  // namespace

namespace ae {

// Walker/Vose alias table (host, setup only): stands in for rand_distr::WeightedAliasIndex::new
// (src/embedder.rs:919,987).  LIFO small/large stacks filled in index order.
void alias_build_host(const float* w, uint64_t n, std::vector<float>& odds, std::vector<uint32_t>& alias) {
    odds.assign(n, 1.0f);
    alias.resize(n);
    double sum = 0.;
    for (uint64_t i = 0; i < n; i++) sum += (double)w[i];
    std::vector<double> q(n);
    std::vector<uint32_t> small, large;
    small.reserve(n);
    large.reserve(n);
    for (uint64_t i = 0; i < n; i++) {
        q[i] = (double)w[i] * (double)n / sum;
        if (q[i] < 1.0) small.push_back((uint32_t)i);
        else large.push_back((uint32_t)i);
    }
    while (!small.empty() && !large.empty()) {
        uint32_t s = small.back(); small.pop_back();
        uint32_t l = large.back(); large.pop_back();
        odds[s] = (float)q[s];
        alias[s] = l;
        q[l] = (q[l] + q[s]) - 1.0;
        if (q[l] < 1.0) small.push_back(l);
        else large.push_back(l);
    }
    for (uint32_t l : large) { odds[l] = 1.0f; alias[l] = l; }
    for (uint32_t s : small) { odds[s] = 1.0f; alias[s] = s; }
}

}  // namespace ae


template <int DIM>
static void launch_hogwild(ae_entropy_optim* o, uint64_t nb_sample, double step, uint32_t iter) {
    unsigned grid = grid_cap(nb_sample, 256, 256 * 32);
    hipLaunchKernelGGL((ce_sgd_hogwild_kernel<DIM>), dim3(grid), dim3(256), 0, stream(), o->dev, o->sample_offset, nb_sample, step,
                       iter, o->err.p);
}
template <int DIM>
static void launch_planned(ae_entropy_optim* o, const uint32_t* order, uint64_t count, double step) {
    hipLaunchKernelGGL((ce_sgd_planned_kernel<DIM>), dim3(blocks_for(count, 256)), dim3(256), 0, stream(), o->dev, order, count,
                       o->plan_nodes.p, o->plan_w.p, step);
}

static void check_err_flag(ae_entropy_optim* o) {
    unsigned int h = 0;
    o->err.download(&h, 1);
    if (h) { o->err.zero(); }  // reported once: the flag does not poison later calls on the handle
    if (h & 16u) fail(AE_ERR_STATE, "event-ordered kernel: a wave's event lists exceeded its LDS pool (window sizing violated)");
    if (h & 32u) fail(AE_ERR_STATE, "event-ordered kernel: poll budget exceeded (a lane's partner never arrived: scheduling invariant violated)");
    if (h & 2u) fail(AE_ERR_INVALID_ARG, "sample plan capacity exceeded (edge probabilities of a row sum to more than 1?)");
    if (h) fail(AE_ERR_INVALID_ARG, "negative sampling could not find 5 admissible nodes (graph too small for its neighbourhood size?)");
}

static int device_cus() {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        AE_HIP(hipGetDevice(&dev));
        hipDeviceProp_t prop;
        AE_HIP(hipGetDeviceProperties(&prop, dev));
        cus = prop.multiProcessorCount;
    }
    return cus;
}

// `run`: the stream of the version fill and the dataflow kernel (null: the library stream) and the CUs it may use; the commit
// follows on the library stream (the caller has ordered it after `run`)
template <int DIM, bool RELAXED>
static void launch_dataflow2(ae_entropy_optim* o, ae_entropy_optim::DfSet& st, uint64_t S, double step, const uint64_t* rowptr, const uint32_t* keys,
                             hipStream_t run, int run_cus) {
    {
        int blocks_per_cu = 0;
        const int cus = run ? run_cus : device_cus();
        StreamScope on_run(run ? run : stream());
        const unsigned bs = debug_knob("AE_DF_BLOCK") ? (unsigned)atoi(debug_knob("AE_DF_BLOCK")) : 128u;
        int bpc = 0;
        AE_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpc, ce_dataflow_kernel<DIM, RELAXED>, (int)bs, 0));
        // The progress argument needs EVERY block resident (a sample of sweep k + 1 in block 0 may wait on a sample of sweep k
        // in the last block).  The occupancy query can be one block per CU higher than what the hardware admits (SGPR
        // granularity, MI355X_MICROARCH.md "Residency"): the grid is capped one full block per CU below it.
        blocks_per_cu = std::max(1, std::min(bpc, 8) - 1);
        const uint64_t blocks_cap = (uint64_t)blocks_per_cu * cus;
        // The run is bound by the dependency chain: per hop one trip of the carrying wave through its loop (poll round
        // trip + the f64 arithmetic of the lanes that advance).  Enough carrier lanes to keep every chain moving
        // (~512 samples per carrier, at least 16 K carriers), spread thinly over the waves (every 8th lane) while the
        // resident grid allows it; big batches are throughput-bound and use every lane.  Measured on MI355X, C2 batch:
        // 256 x 128 lanes all carrying 10.0 ms; 1024 x 128 with every 4th 8.8 ms, every 8th 8.2 ms, every 16th 10.1 ms;
        // C3 shape (100 M samples, full grid): all lanes 17 ms, every 2nd 22 ms, every 8th 53 ms.
        const uint64_t carriers_want = std::min<uint64_t>(S, std::max<uint64_t>(16384, S / 512));
        uint32_t lane_stride = 8;
        while (lane_stride > 1 && blocks_cap * bs / lane_stride < carriers_want) lane_stride /= 2;
        if (debug_knob("AE_DF_LANE_STRIDE")) lane_stride = (uint32_t)atoi(debug_knob("AE_DF_LANE_STRIDE"));
        if (lane_stride == 0 || (bs % lane_stride) != 0) lane_stride = 1;
        unsigned grid = (unsigned)std::min<uint64_t>(blocks_cap, std::max<uint64_t>(1, (carriers_want * lane_stride + bs - 1) / bs));
        if (debug_knob("AE_DF_GRID")) grid = (unsigned)std::min<uint64_t>(blocks_cap, std::max(1, atoi(debug_knob("AE_DF_GRID"))));
        CeDev dev = o->dev;
        const uint32_t* pn = st.plan_nodes.p;
        const float* pw = st.plan_w.p;
        const uint32_t* pred = st.pred.p;
        float* ver = o->df_ver.p;
        unsigned int* err = o->err.p;
        void* args[] = {&dev, &S, &pn, &pw, &pred, &ver, &step, &err, &lane_stride};
        {   // every version "unpublished"; in pieces of 1 GiB (one call for tens of GB was seen to leave part of the buffer unset)
            const size_t bytes = sizeof(float) * S * 2 * DIM, piece = size_t(1) << 30;
            for (size_t off = 0; off < bytes; off += piece)
                AE_HIP(hipMemsetAsync(reinterpret_cast<char*>(ver) + off, 0xFF, std::min(piece, bytes - off), stream()));
        }
        // A plain launch: same residency as a cooperative one without its +15-19 us (the cap above is the check).
        if (o->df_events.size() >= 64) {  // nobody asks for the timings: keep the list short
            (void)hipEventDestroy(o->df_events.front().first);
            (void)hipEventDestroy(o->df_events.front().second);
            o->df_events.erase(o->df_events.begin());
        }
        hipEvent_t e0, e1;
        AE_HIP(hipEventCreate(&e0));
        AE_HIP(hipEventCreate(&e1));
        AE_HIP(hipEventRecord(e0, stream()));
        (void)args;
        // f32 repulsion coefficients (debug knob AE_DF_F32_REPULSION): C2 batch 5.8 -> 5.4 ms, but the faster kernel's wave skew shows
        // in the fidelity on the stiff k = 6 graph (final CE 1.010 x the sequential mode's over three seeds, f64: 1.000): not the default
        const uint32_t stride_arg = lane_stride | ((RELAXED && debug_knob("AE_DF_F32_REPULSION")) ? 0x80000000u : 0u);
        hipLaunchKernelGGL((ce_dataflow_kernel<DIM, RELAXED>), dim3(grid), dim3(bs), 0, stream(), dev, S, pn, pw, pred, ver, step, err, stride_arg);
        AE_HIP(hipEventRecord(e1, stream()));
        o->df_events.emplace_back(e0, e1);
    }
    if (run) {  // back on the library stream, after the kernel
        AE_HIP(hipEventRecord(o->df_ahead.ran, run));
        AE_HIP(hipStreamWaitEvent(stream(), o->df_ahead.ran, 0));
    }
    hipLaunchKernelGGL((df_commit_kernel<DIM>), dim3(blocks_for(o->dev.n, 256)), dim3(256), 0, stream(), o->dev.n, rowptr, keys,
                       (const float*)o->df_ver.p, o->dev.y, (const unsigned int*)o->err.p);
}

template <int DIM>
static void launch_dataflow(ae_entropy_optim* o, ae_entropy_optim::DfSet& st, uint64_t S, double step, const uint64_t* rowptr, const uint32_t* keys, bool relaxed,
                            hipStream_t run, int run_cus) {
    if (relaxed) launch_dataflow2<DIM, true>(o, st, S, step, rowptr, keys, run, run_cus);
    else launch_dataflow2<DIM, false>(o, st, S, step, rowptr, keys, run, run_cus);
}

// everything of a sequential batch that depends only on (graph, RNG stream, batch index): the plan of its samples, their write
// events sorted by node, every read's predecessor.  Runs on whatever stream() is current.
static void df_reserve_set(ae_entropy_optim* o, ae_entropy_optim::DfSet& st, uint64_t S) {
    if (st.plan_nodes.n < S * 7) st.plan_nodes.alloc(S * 7);
    if (st.plan_w.n < S) st.plan_w.alloc(S);
    if (st.pred.n < S * 7) st.pred.alloc(S * 7);
    if (st.keys0.n < 2 * S + 16) { st.keys0.alloc(2 * S + 16); st.keys1.alloc(2 * S + 16); }  // (+16: df_pred_reads_kernel's window may overhang)
    if (st.rowptr.n < o->dev.n + 1) st.rowptr.alloc(o->dev.n + 1);
}
static void df_prepare_set(ae_entropy_optim* o, ae_entropy_optim::DfSet& st, uint64_t S, uint32_t iter, bool relaxed) {
    df_reserve_set(o, st, S);
    launch_plan(o, o->sample_offset, S, iter, st.plan_nodes.p, st.plan_w.p);
    // each key buffer holds 2 S node keys followed by 2 S version ids
    uint32_t* k0 = reinterpret_cast<uint32_t*>(st.keys0.p);
    uint32_t* k1 = reinterpret_cast<uint32_t*>(st.keys1.p);
    uint32_t *v0 = k0 + 2 * S, *v1 = k1 + 2 * S;
    hipLaunchKernelGGL(df_write_keys_kernel, dim3(blocks_for(S, 256)), dim3(256), 0, stream(), S, (const uint32_t*)st.plan_nodes.p, k0, v0);
    unsigned node_bits = 1;
    while (node_bits < 32 && (o->dev.n >> node_bits)) node_bits++;
    sort_pairs_u32_u32(k0, k1, v0, v1, 2 * S, node_bits);
    hipLaunchKernelGGL(df_rowptr_kernel, dim3(blocks_for(o->dev.n + 1, 256)), dim3(256), 0, stream(), (const uint32_t*)k1, 2 * S,
                       (uint64_t)o->dev.n, st.rowptr.p);
    hipLaunchKernelGGL(df_pred_writes_kernel, dim3(grid_cap(2 * S, 256, 1u << 20)), dim3(256), 0, stream(), S, (const uint32_t*)k1, (const uint32_t*)v1,
                       st.pred.p);
    if (!relaxed)  // (the relaxed form never waits for a negative: no predecessor to find)
        hipLaunchKernelGGL(df_pred_reads_kernel, dim3(grid_cap(S * 5, 256, 1u << 22)), dim3(256), 0, stream(), S, (const uint32_t*)st.plan_nodes.p,
                           (const uint32_t*)v1, (const uint64_t*)st.rowptr.p, st.pred.p);
    check_launch("df_pred");
}

// Two CU-masked streams for the overlap of a batch's dataflow kernel with the preparation of the next one: the first quarter of the
// CU mask bits (tools/probe_cumask.hip: bit b = one CU of XCD b % 8, so every XCD gives the same share) prepares, the rest runs.
static bool df_ahead_streams(ae_entropy_optim* o) {
    auto& a = o->df_ahead;
    if (a.tried) return a.run != nullptr;
    a.tried = 1;
    const int cus = device_cus();
    int prep_cus = cus / 4;
    if (debug_knob("AE_DF_PREP_CUS")) prep_cus = std::max(8, std::min(cus - 8, atoi(debug_knob("AE_DF_PREP_CUS"))));
    std::vector<uint32_t> m_prep((cus + 31) / 32, 0u), m_run((cus + 31) / 32, 0u);
    for (int b = 0; b < cus; b++) (b < prep_cus ? m_prep : m_run)[b / 32] |= 1u << (b % 32);
    hipStream_t run = nullptr, prep = nullptr;
    if (hipExtStreamCreateWithCUMask(&run, (uint32_t)m_run.size(), m_run.data()) != hipSuccess ||
        hipExtStreamCreateWithCUMask(&prep, (uint32_t)m_prep.size(), m_prep.data()) != hipSuccess) {
        (void)hipGetLastError();
        if (run) (void)hipStreamDestroy(run);
        return false;  // no CU masks on this runtime: batches run one after the other on the library stream
    }
    AE_HIP(hipEventCreateWithFlags(&a.start, hipEventDisableTiming));
    AE_HIP(hipEventCreateWithFlags(&a.ran, hipEventDisableTiming));
    AE_HIP(hipEventCreateWithFlags(&a.prepared, hipEventDisableTiming));
    a.run = run;
    a.prep = prep;
    a.run_cus = cus - prep_cus;
    return true;
}

// AE_CE_SEQUENTIAL / AE_CE_ORDERED scheduled on the device (see the dataflow kernels above).
// The preparation of a batch depends only on graph, RNG stream and batch index: while the dataflow kernel of batch b runs, the set of
// batch b + 1 (same size, next index -- what every caller's schedule asks next; anything else is prepared afresh) is built on a second
// stream.  On the SAME CUs that was tried and dropped, twice: the kernels do overlap, but the latency-bound dataflow slows by what the
// overlap saves (exact form C2 11.3 -> 11.5 ms per batch; ordered form 5.79 -> 5.73 ms) and the co-running planner skews the ordered
// kernel's waves enough to show in its fidelity on the stiff k = 6 graph (final CE 1.000 -> 1.010 of the sequential mode's).  A hop's
// price sits in the memory queue of the CU that polls, so the two get DISJOINT CUs (CU-masked streams): a quarter of the CUs prepares.
// Measured (MI355X, C2 batch): exact form 10.76 -> 9.28 ms (its kernel 8.25 -> 9.15 ms on 192 CUs -- more waves per CU, longer
// hops -- but 2.4 ms of preparation gone from the critical path); ordered form 5.81 -> 5.68 ms at C2 and 2.86 -> 3.13 ms at 60 k x
// k 6 (kernel 4.48 -> 5.21 ms against 1.3 ms of preparation hidden; with 32 preparing CUs the preparation, 8.6 ms, becomes the
// critical path): the exact form overlaps, the ordered form does not.  Only where the dataflow is latency-bound (thin waves: batches
// of up to 2^24 samples); bigger batches need every CU for either part.
static void run_sequential_dataflow(ae_entropy_optim* o, uint64_t S, double step, uint32_t iter, bool relaxed) {
    if (S >= (1ull << 31)) fail(AE_ERR_INVALID_ARG, "sequential mode supports < 2^31 samples per batch");
    const uint32_t dim = o->dev.dim;
    if (o->df_ver.n < S * 2 * dim) o->df_ver.alloc(S * 2 * dim);
    const bool prof = debug_knob("AE_CE_PROF") != nullptr;
    auto now = [&] { if (prof) sync(); return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
    auto& ahead = o->df_ahead;
    bool overlap = !relaxed && S <= (1ull << 24);
    if (debug_knob("AE_DF_AHEAD")) overlap = atoi(debug_knob("AE_DF_AHEAD")) != 0;
    overlap = overlap && df_ahead_streams(o);
    uint32_t cur = 0;
    if (ahead.valid && overlap && ahead.S == S && ahead.iter == iter && ahead.relaxed == relaxed) {
        cur = ahead.set;
        AE_HIP(hipStreamWaitEvent(stream(), ahead.prepared, 0));
    } else {
        if (ahead.valid) AE_HIP(hipStreamSynchronize(ahead.prep));  // a set nobody asked for: let it finish before its buffers are reused
        df_prepare_set(o, o->df_sets[0], S, iter, relaxed);
    }
    ahead.valid = false;
    ae_entropy_optim::DfSet& st = o->df_sets[cur];
    const double t1 = now();
    if (overlap) {
        df_reserve_set(o, o->df_sets[cur ^ 1], S);
        AE_HIP(hipEventRecord(ahead.start, stream()));
        AE_HIP(hipStreamWaitEvent(ahead.run, ahead.start, 0));
        AE_HIP(hipStreamWaitEvent(ahead.prep, ahead.start, 0));
    }
    uint32_t* v1 = reinterpret_cast<uint32_t*>(st.keys1.p) + 2 * S;
    AE_DISPATCH_DIM(dim, launch_dataflow, o, st, S, step, (const uint64_t*)st.rowptr.p, (const uint32_t*)v1, relaxed, overlap ? ahead.run : nullptr,
                    ahead.run_cus);
    check_launch("ce_dataflow");
    if (overlap) {
        StreamScope on_prep(ahead.prep);
        df_prepare_set(o, o->df_sets[cur ^ 1], S, iter + 1, relaxed);
        AE_HIP(hipEventRecord(ahead.prepared, ahead.prep));
        ahead.valid = true;
        ahead.S = S;
        ahead.iter = iter + 1;
        ahead.relaxed = relaxed;
        ahead.set = cur ^ 1;
    }
    sync();
    if (prof) fprintf(stderr, "CESEQ dataflow samples=%llu: plan + sort + predecessors %.2f ms, dataflow + commit %.2f ms\n",
                      (unsigned long long)S, (t1 - t0) * 1e3, (now() - t1) * 1e3);
    unsigned int h = 0;
    o->err.download(&h, 1);
    if (h & 8u) {
        o->err.zero();  // reported once: the flag does not poison later calls on the handle
        sync();
        // AE_CE_SEQUENTIAL commits its row versions only after a clean run (df_commit_kernel skips on the flag): the coordinates are
        // the batch's start.  AE_CE_ORDERED has stored every finished sample's rows in place: what is left is a partial batch.
        fail(AE_ERR_STATE, "%s dataflow kernel: poll budget exceeded (not every workgroup was resident -- is another process using this GPU? -- or the scheduling invariant was violated); %s",
             relaxed ? "ordered" : "sequential",
             relaxed ? "AE_CE_ORDERED stores rows in place: the coordinates hold a PARTIAL batch (the samples that finished) -- restart from a saved embedding"
                     : "the coordinates are those of the batch's start");
    }
    check_err_flag(o);
}

static void run_sequential(ae_entropy_optim* o, uint64_t nb_sample, double step, uint32_t iter) {
    // device-scheduled form; AE_CE_SEQ_LEVELS=1 (debug knob) keeps the host level schedule for A/B
    const bool relaxed = o->params.ce_mode == AE_CE_ORDERED;
    if (relaxed || !debug_knob("AE_CE_SEQ_LEVELS")) {
        run_sequential_dataflow(o, nb_sample, step, iter, relaxed);
        return;
    }
    if (nb_sample >= 0xFFFFFFFFull) fail(AE_ERR_INVALID_ARG, "sequential mode supports < 2^32 samples per batch");
    if (o->plan_nodes.n < nb_sample * 7) o->plan_nodes.alloc(nb_sample * 7);
    if (o->plan_w.n < nb_sample) o->plan_w.alloc(nb_sample);
    if (o->order.n < nb_sample) o->order.alloc(nb_sample);
    launch_plan(o, o->sample_offset, nb_sample, iter, o->plan_nodes.p, o->plan_w.p);
    const bool prof = debug_knob("AE_CE_PROF") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
    std::vector<uint32_t> nodes(nb_sample * 7);
    o->plan_nodes.download(nodes.data(), nb_sample * 7);
    const double t1 = now();
    check_err_flag(o);
    // level schedule: sample s runs after every earlier sample that wrote a node it touches and after
    // every earlier sample that read a node it writes  =>  identical to sequential execution.
    const uint64_t n = o->dev.n;
    std::vector<uint32_t> last_w(n, 0), last_r(n, 0), level(nb_sample);
    uint32_t max_level = 0;
    for (uint64_t s = 0; s < nb_sample; s++) {
        const uint32_t* p = &nodes[s * 7];
        uint32_t l = std::max(std::max(last_w[p[0]], last_r[p[0]]), std::max(last_w[p[1]], last_r[p[1]]));
        for (int g = 0; g < 5; g++) l = std::max(l, last_w[p[2 + g]]);
        l += 1;
        level[s] = l;
        last_w[p[0]] = l;
        last_w[p[1]] = l;
        for (int g = 0; g < 5; g++) last_r[p[2 + g]] = std::max(last_r[p[2 + g]], l);
        max_level = std::max(max_level, l);
    }
    uint32_t raw_depth = 0;
    if (prof) {  // depth of the true (read-after-write) dependencies alone: what a multi-version schedule would need
        std::vector<uint32_t> lw(n, 0);
        for (uint64_t s = 0; s < nb_sample; s++) {
            const uint32_t* p = &nodes[s * 7];
            uint32_t l = 0;
            for (int g = 0; g < 7; g++) l = std::max(l, lw[p[g]]);
            l += 1;
            lw[p[0]] = l;
            lw[p[1]] = l;
            raw_depth = std::max(raw_depth, l);
        }
    }
    std::vector<uint64_t> off(max_level + 2, 0);
    for (uint64_t s = 0; s < nb_sample; s++) off[level[s] + 1]++;
    for (uint32_t l = 0; l <= max_level; l++) off[l + 1] += off[l];
    std::vector<uint32_t> order(nb_sample);
    {
        std::vector<uint64_t> cur(off.begin(), off.end() - 1);
        for (uint64_t s = 0; s < nb_sample; s++) order[cur[level[s]]++] = (uint32_t)s;
    }
    const double t2 = now();
    o->order.upload(order.data(), nb_sample);
    for (uint32_t l = 1; l <= max_level; l++) {
        uint64_t cnt = off[l + 1] - off[l];
        if (!cnt) continue;
        AE_DISPATCH_DIM(o->dev.dim, launch_planned, o, o->order.p + off[l], cnt, step);
    }
    check_launch("ce_sgd_planned");
    sync();
    if (prof)
        fprintf(stderr, "CESEQ samples=%llu levels=%u (read-after-write depth %u) plan+download %.1f ms, host schedule %.1f ms, upload+%u launches %.1f ms\n",
                (unsigned long long)nb_sample, max_level, raw_depth, (t1 - t0) * 1e3, (t2 - t1) * 1e3, max_level, (now() - t2) * 1e3);
}



// AE_CE_AUTO resolves to the fastest mode whose output is the reference's: for batches of up to kAutoOrderedSamples
// samples the ordered dataflow (AE_CE_ORDERED: the sequential order, end points sequentially consistent, negatives as the memory
// system has them -- half the latency of the bit-exact AE_CE_SEQUENTIAL, which stays the parity mode, by name), beyond that the
// time-sliced mode (throughput-bound, a tenth of the memory).  The cross-over depends on the graph: ~30 M samples per batch on the
// node-permuted lattice of the scale benchmarks (uniform in-degree: few conflicts), ~250 M on the exact kNN graph of Higgs-shaped points
// with hubness weighting as round 3 measured it -- the rule follows the real graph.  Every asked_dim in [1, 64] has both (rows are stored
// zero-padded, ce_internal.h).  A SHARDED node range (multi-GPU) resolves to the time-sliced mode: it runs a shard's own events on
// current rows and reads the other shards' rows as of the last exchange -- faithful where few edges cross shards (a partition by
// connected components / locality; ce_slice_prepare refuses a shard with more than 10 % of its edge mass on cross-shard edges, and
// the caller then asks for the approximate rounds mode, AE_CE_HOGWILD, by name).
// Round 5: the time-sliced mode runs under-filled slices MERGED (one launch per slice, ce_slice_kernels.h) and takes over earlier: exact
// kNN graphs of Higgs-shaped points with hubness weighting, ordered / time-sliced ms per batch: 24 M samples 11.8 / 18.5, 48 M 23.4 /
// 21.4, 72 M 34.5 / 27.1, 99 M (configs[2]'s large graph) 49.5 / 33.8 -- break-even at ~48 M (tools/run_auto_crossover.py); with the
// dependency words in the rows' lines: 24 M 11.8 / 14.9, 36 M 19.9 / 14.9, 48 M 23.5 / 17.1, 72 M 36.1 / 22.4, 99 M 49.5 / 27.2 --
// break-even at ~29 M.
// The ~29 M break-even belongs to rows of <= 8 columns, whose dependency words ride in the rows' own lines; wider rows keep the words
// in an array of their own (the first set of figures above: break-even ~48 M samples per batch).
constexpr uint64_t kAutoOrderedSamples = 1ull << 25;        // 33.6 M: rows of <= 8 columns
constexpr uint64_t kAutoOrderedSamplesWide = 48ull << 20;   // 50.3 M: wider rows
uint32_t ae::resolve_ce_mode(uint32_t mode, uint64_t dim, bool sharded, uint64_t samples_per_batch, uint32_t max_nbng, uint64_t nnz) {
    if (mode > AE_CE_ORDERED) fail(AE_ERR_INVALID_ARG, "unknown ce_mode %u", mode);
    if (mode != AE_CE_AUTO) return mode;
    const bool sliced_ok = max_nbng <= 32 && nnz < 0xFFFFFFFFull;
    if (sharded) {
        if (!sliced_ok) fail(AE_ERR_INVALID_ARG, "AE_CE_AUTO on a sharded node range needs the time-sliced mode (rows of <= 32 neighbours, < 2^32 edges); "
                                                 "ask for the approximate rounds mode by name (ce_mode = AE_CE_HOGWILD)");
        return AE_CE_SLICED;
    }
    // (the ordered dataflow's scratch: 92 bytes of plan, sorted events and predecessors per sample + two published rows; long rows
    // at the top of the range would ask for more than a sixth of the device: 48 GB is the line)
    const uint64_t ordered_scratch = samples_per_batch * (92ull + 8ull * ae_pad_dim((uint32_t)dim));
    const uint64_t ordered_up_to = ae_pad_dim((uint32_t)dim) <= 8 ? kAutoOrderedSamples : kAutoOrderedSamplesWide;
    if ((samples_per_batch <= ordered_up_to && ordered_scratch <= (48ull << 30)) || !sliced_ok) {
        if (samples_per_batch >= (1ull << 31)) fail(AE_ERR_INVALID_ARG, "no faithful CE mode fits: rows of more than 32 neighbours or >= 2^32 edges with >= 2^31 samples per batch");
        return AE_CE_ORDERED;
    }
    return AE_CE_SLICED;
}

ae_entropy_optim* ae::entropy_optim_create_impl(const ae_kgraph* g, const ae_node_params* np, const ae_embedder_params* params,
                                                const float* y0, bool y0_on_device, const uint32_t* hub_counts, uint64_t node_lo,
                                                uint64_t node_hi) {
    {
        require_device();
        if (!g || !np || !params || !y0) fail(AE_ERR_INVALID_ARG, "null argument");
        if (np->g != g) fail(AE_ERR_INVALID_ARG, "node params were not computed from this graph");
        if (params->asked_dim == 0 || params->asked_dim > 64) fail(AE_ERR_INVALID_ARG, "asked_dim must be in [1,64]");
        if (node_lo >= node_hi || node_hi > g->n) fail(AE_ERR_INVALID_ARG, "bad node range");
        if (params->ce_precision != AE_PRECISION_F64 && params->ce_precision != AE_PRECISION_F32) fail(AE_ERR_INVALID_ARG, "unknown ce_precision %u", params->ce_precision);
        if (g->n < (uint64_t)g->max_nbng + 8) fail(AE_ERR_INVALID_ARG, "graph too small for negative sampling");
        std::unique_ptr<ae_entropy_optim> o(new ae_entropy_optim);
        o->g = g;
        o->np = np;
        o->params = *params;
        const uint64_t n = g->n, adim = params->asked_dim, dim = ae_pad_dim(adim);  // dim: the row stride (ce_internal.h)
        o->y.alloc(n * dim);
        if (dim != adim) o->y.zero();
        AE_HIP(hipMemcpy2DAsync(o->y.p, sizeof(float) * dim, y0, sizeof(float) * adim, sizeof(float) * adim, n,
                                y0_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, stream()));
        // embedded scales: mean of the initial scales as the reference's sequential f32 sum (:1358)
        // (on the device: the reference's sequential f32 order as a single-lane chain -- bit parity of the scales with the oracle --, a
        // tree reduction while the embedder has switched the summation order for a mode that is not the bit-exact one: linalg.h)
        // (AE_CE_AUTO never resolves to the bit-exact mode: it runs by name only)
        const bool bit_exact_follows = params->ce_mode == AE_CE_SEQUENTIAL;
        // (an enclosing scope -- the embedder's, which applied the same rule to the mode it resolved -- is kept; a stage-level call decides
        // by its own mode, whatever the process-wide default says)
        TreeSums sums(tree_sums_scope() >= 0 ? tree_sums() : !bit_exact_follows);
        const float mean_scale = seq_sum_f32(np->scale.p, n) / (float)n;
        o->emb_scale.alloc(n);
        hipLaunchKernelGGL(embedded_scales_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, stream(), n, np->scale.p, mean_scale,
                           o->emb_scale.p);
        check_launch("embedded_scales");
        o->err.alloc(1);
        o->err.zero();
        o->partial.alloc(1024);
        std::vector<uint64_t> hindptr;  // only needed for shard edge bounds
        uint64_t edge_lo, edge_hi;
        if (g->uniform_k) {
            edge_lo = node_lo * g->uniform_k;
            edge_hi = node_hi * g->uniform_k;
        } else {
            uint64_t two[2];
            AE_HIP(hipMemcpy(&two[0], g->indptr.p + node_lo, sizeof(uint64_t), hipMemcpyDeviceToHost));
            AE_HIP(hipMemcpy(&two[1], g->indptr.p + node_hi, sizeof(uint64_t), hipMemcpyDeviceToHost));
            edge_lo = two[0];
            edge_hi = two[1];
        }
        CeDev& d = o->dev;
        memset(&d, 0, sizeof(d));
        d.n = n; d.nnz = g->nnz; d.dim = (uint32_t)dim; d.ystride = (uint32_t)dim; d.uniform_k = g->uniform_k;
        d.indptr = g->indptr.p; d.nbr = g->nbr.p; d.proba = np->proba.p; d.emb_scale = o->emb_scale.p; d.y = o->y.p;
        d.b = params->b; d.seed = params->seed; d.sampler = params->ce_sampler;
        d.node_lo = node_lo; d.node_hi = node_hi; d.edge_lo = edge_lo; d.shard_edges = edge_hi - edge_lo;
        o->sample_offset = node_lo << 24;
        if (params->ce_sampler == AE_SAMPLER_ALIAS) {
            // WeightedAliasIndex over the edge probabilities (embedder.rs:987); host build, setup only
            std::vector<float> hp(d.shard_edges);
            AE_HIP(hipMemcpy(hp.data(), np->proba.p + edge_lo, sizeof(float) * d.shard_edges, hipMemcpyDeviceToHost));
            std::vector<float> odds;
            std::vector<uint32_t> alias;
            alias_build_host(hp.data(), d.shard_edges, odds, alias);
            o->edge_odds.alloc(d.shard_edges); o->edge_odds.upload(odds.data(), d.shard_edges);
            o->edge_alias.alloc(d.shard_edges); o->edge_alias.upload(alias.data(), d.shard_edges);
            o->edge_src.alloc(d.shard_edges);
            hipLaunchKernelGGL(edge_src_kernel, dim3(blocks_for(node_hi - node_lo, 256)), dim3(256), 0, stream(), node_lo, node_hi,
                               g->indptr.p, edge_lo, o->edge_src.p);
            check_launch("edge_src");
            sync();
            d.edge_odds = o->edge_odds.p; d.edge_alias = o->edge_alias.p; d.edge_src = o->edge_src.p;
        } else if (params->ce_sampler != AE_SAMPLER_ROWCDF) {
            fail(AE_ERR_INVALID_ARG, "unknown ce_sampler %u", params->ce_sampler);
        }
        if (params->hubness_weighting) {
            if (!hub_counts) fail(AE_ERR_INVALID_ARG, "hubness_weighting needs hub_counts");
            // NodeSampler::new, embedder.rs:826-833, 915-919
            std::vector<float> w(n);
            const float upper = (float)n;
            float s = 0.f;
            for (uint64_t i = 0; i < n; i++) {
                w[i] = std::min(std::max((float)hub_counts[i], 1.f), upper);
                s += w[i];
            }
            const float mean = s / (float)n;
            for (uint64_t i = 0; i < n; i++) w[i] = w[i] / mean;
            std::vector<float> odds;
            std::vector<uint32_t> alias;
            alias_build_host(w.data(), n, odds, alias);
            o->hub_odds.alloc(n); o->hub_odds.upload(odds.data(), n);
            o->hub_alias.alloc(n); o->hub_alias.upload(alias.data(), n);
            sync();
            std::vector<uint2> tab(n);
            for (uint64_t i = 0; i < n; i++) { uint32_t bits; memcpy(&bits, &odds[i], 4); tab[i] = make_uint2(bits, alias[i]); }
            o->hub_tab.alloc(n); o->hub_tab.upload(tab.data(), n);
            sync();
            d.hub_odds = o->hub_odds.p; d.hub_alias = o->hub_alias.p; d.hub_tab = o->hub_tab.p;
        }
        sync();
        const bool sharded = node_lo != 0 || node_hi != n;
        const uint32_t mode = resolve_ce_mode(params->ce_mode, dim, sharded, params->nb_sampling_by_edge * (edge_hi - edge_lo), g->max_nbng, g->nnz);
        uint32_t mode_final = mode;
        if (params->ce_mode == AE_CE_AUTO && mode == AE_CE_SLICED && ce_slice_unsupported(o.get())) {  // (e.g. more than 2^27 nodes)
            if (params->nb_sampling_by_edge * (edge_hi - edge_lo) >= (1ull << 31))
                fail(AE_ERR_INVALID_ARG, "no faithful CE mode fits this problem: AE_CE_SLICED: %s; AE_CE_ORDERED: >= 2^31 samples per batch", ce_slice_unsupported(o.get()));
            mode_final = AE_CE_ORDERED;
        }
        if (mode_final == AE_CE_EVENT || mode_final == AE_CE_HOGWILD) ce_node_build_transpose(o.get());
        if (mode_final == AE_CE_EVENT) ce_event_prepare(o.get());
        o->params.ce_mode = mode_final;  // (ce_slice_prepare reads it: the cost model of the matching cut belongs to this mode only)
        // (a sharded range prepares once its communicator is attached and every rank's range known -- the internal numbering depends
        // on them --, or with its first batch: not twice)
        if (mode_final == AE_CE_SLICED && !sharded) ce_slice_prepare(o.get());
        return o.release();
    }
}

extern "C" {

int32_t ae_entropy_optim_create(const ae_kgraph* g, const ae_node_params* np, const ae_embedder_params* params, const float* y0,
                                const uint32_t* hub_counts, uint64_t node_lo, uint64_t node_hi, ae_entropy_optim** out) {
    return guard([&] {
        if (!out) fail(AE_ERR_INVALID_ARG, "null argument");
        *out = ae::entropy_optim_create_impl(g, np, params, y0, false, hub_counts, node_lo, node_hi);
    });
}

int32_t ae_entropy_optim_destroy(ae_entropy_optim* o) {
    return guard([&] { delete o; });
}
int32_t ae_entropy_optim_get_nb_edges(const ae_entropy_optim* o, uint64_t* nnz) {
    return guard([&] {
        if (!o || !nnz) fail(AE_ERR_INVALID_ARG, "null argument");
        *nnz = o->dev.shard_edges;
    });
}

int32_t ae_entropy_optim_get_ce_mode(const ae_entropy_optim* o, uint32_t* ce_mode) {
    return guard([&] {
        if (!o || !ce_mode) fail(AE_ERR_INVALID_ARG, "null argument");
        *ce_mode = o->params.ce_mode;
    });
}

int32_t ae_entropy_optim_slice_info(const ae_entropy_optim* o, uint32_t* classes, double* overflow_fraction, uint32_t* colouring_rounds,
                                    uint32_t* slices_last_batch) {
    return guard([&] {
        if (!o) fail(AE_ERR_INVALID_ARG, "null argument");
        if (o->params.ce_mode != AE_CE_SLICED) fail(AE_ERR_STATE, "the handle does not run AE_CE_SLICED");
        if (!o->sl_prepared && !ce_slice_unsupported(o)) ce_slice_prepare(const_cast<ae_entropy_optim*>(o));
        if (classes) *classes = o->sl_classes;
        if (overflow_fraction) *overflow_fraction = o->sl_ov_frac;
        if (colouring_rounds) *colouring_rounds = o->sl_color_rounds;
        if (slices_last_batch) *slices_last_batch = o->rounds;
    });
}

int32_t ae_entropy_optim_slice_hub_info(const ae_entropy_optim* o, uint32_t* max_in_degree, double* busiest_row_events_per_step) {
    return guard([&] {
        if (!o) fail(AE_ERR_INVALID_ARG, "null argument");
        if (o->params.ce_mode != AE_CE_SLICED) fail(AE_ERR_STATE, "the handle does not run AE_CE_SLICED");
        if (!o->sl_prepared && !ce_slice_unsupported(o)) ce_slice_prepare(const_cast<ae_entropy_optim*>(o));
        if (max_in_degree) *max_in_degree = o->sl_max_in_degree;
        // (half an event per node and slice on average: a row of degree D sees 0.5 D / mean degree of them, spread over the classes)
        if (busiest_row_events_per_step)
            *busiest_row_events_per_step = o->sl_classes ? 0.5 * (double)(o->sl_max_in_degree + o->g->max_nbng) * (double)o->dev.n / (double)(2 * o->dev.nnz) / (double)o->sl_classes : 0.;
    });
}

int32_t ae_entropy_optim_slice_form(const ae_entropy_optim* o, uint32_t* form) {
    return guard([&] {
        if (!o || !form) fail(AE_ERR_INVALID_ARG, "null argument");
        if (o->params.ce_mode != AE_CE_SLICED) fail(AE_ERR_STATE, "the handle does not run AE_CE_SLICED");
        *form = o->sl_last_form;
    });
}

int32_t ae_entropy_optim_ce(ae_entropy_optim* o, double* ce) {
    return guard([&] {
        require_device();
        if (!o || !ce) fail(AE_ERR_INVALID_ARG, "null argument");
        const unsigned grid = std::min<unsigned>(1024, blocks_for(o->dev.node_hi - o->dev.node_lo, 256));
        hipLaunchKernelGGL(ce_value_kernel, dim3(grid), dim3(256), 0, stream(), o->dev, o->partial.p);
        check_launch("ce_value");
        std::vector<double> h(grid);
        o->partial.download(h.data(), grid);
        double s = 0.;
        for (unsigned i = 0; i < grid; i++) s += h[i];
        *ce = s;
    });
}

int32_t ae_entropy_optim_gradient_iteration(ae_entropy_optim* o, uint64_t nb_sample, double grad_step, uint64_t iter) {
    return guard([&] {
        require_device();
        if (!o) fail(AE_ERR_INVALID_ARG, "null argument");
        if (nb_sample == 0) return;
        if (nb_sample >= (1ull << 56)) fail(AE_ERR_INVALID_ARG, "too many samples");
        if (o->params.ce_mode == AE_CE_SLICED) {
            if (const char* why = ce_slice_unsupported(o)) fail(AE_ERR_INVALID_ARG, "AE_CE_SLICED: %s", why);
        } else if (o->params.ce_mode == AE_CE_EVENT) {
            if (const char* why = ce_event_unsupported(o)) fail(AE_ERR_INVALID_ARG, "AE_CE_EVENT: %s; use AE_CE_SEQUENTIAL or AE_CE_HOGWILD", why);
        } else if (o->params.ce_mode != AE_CE_SAMPLE_RACY && o->params.ce_mode != AE_CE_SEQUENTIAL && o->params.ce_mode != AE_CE_ORDERED && !ce_node_supports(o)) {
            fail(AE_ERR_INVALID_ARG, "AE_CE_HOGWILD needs asked_dim <= 32 and rows of <= 32 neighbours (longer rows: asked_dim in {2,3,4,8,16}); use AE_CE_SEQUENTIAL");
        }
        // one event pair per timed batch, created after validation; ae_entropy_optim_kernel_time drains the list -- a caller that
        // never asks keeps at most kMaxTimedBatches pairs (older batches are folded into a running sum)
        constexpr size_t kMaxTimedBatches = 64;
        if (o->events.size() >= kMaxTimedBatches) {
            AE_HIP(hipEventSynchronize(o->events.front().second));
            float ms = 0.f;
            AE_HIP(hipEventElapsedTime(&ms, o->events.front().first, o->events.front().second));
            o->events_folded_ms += ms;
            o->events_folded++;
            (void)hipEventDestroy(o->events.front().first);
            (void)hipEventDestroy(o->events.front().second);
            o->events.erase(o->events.begin());
        }
        hipEvent_t e0, e1;
        AE_HIP(hipEventCreate(&e0));
        AE_HIP(hipEventCreate(&e1));
        AE_HIP(hipEventRecord(e0, stream()));
        if (o->params.ce_mode == AE_CE_SEQUENTIAL || o->params.ce_mode == AE_CE_ORDERED) {
            run_sequential(o, nb_sample, grad_step, (uint32_t)iter);
            AE_HIP(hipEventRecord(e1, stream()));
            o->events.emplace_back(e0, e1);
            return;
        }
        if (o->params.ce_mode == AE_CE_SLICED) {
            ce_slice_gradient_iteration(o, nb_sample, grad_step, (uint32_t)iter);
            AE_HIP(hipEventRecord(e1, stream()));
            o->events.emplace_back(e0, e1);
            return;
        }
        if (o->params.ce_mode == AE_CE_EVENT) {
            ce_event_gradient_iteration(o, nb_sample, grad_step, (uint32_t)iter);
            AE_HIP(hipEventRecord(e1, stream()));
            o->events.emplace_back(e0, e1);
            return;
        }
        if (o->params.ce_mode == AE_CE_SAMPLE_RACY) {
            AE_DISPATCH_DIM(o->dev.dim, launch_hogwild, o, nb_sample, grad_step, (uint32_t)iter);
            check_launch("ce_sgd_hogwild");
        } else {
            ce_node_gradient_iteration(o, nb_sample, grad_step, (uint32_t)iter);
        }
        AE_HIP(hipEventRecord(e1, stream()));
        o->events.emplace_back(e0, e1);
    });
}

int32_t ae_entropy_optim_gradient_iteration_lockstep(ae_entropy_optim* const* shards, uint32_t world, const uint64_t* nb_sample, double grad_step,
                                                     uint64_t iter, uint32_t exchanges_per_batch) {
    return guard([&] {
        require_device();
        if (!shards || !nb_sample || world == 0) fail(AE_ERR_INVALID_ARG, "null argument");
        uint64_t expect = 0;
        for (uint32_t q = 0; q < world; q++) {
            ae_entropy_optim* o = shards[q];
            if (!o) fail(AE_ERR_INVALID_ARG, "null shard");
            if (o->params.ce_mode != AE_CE_HOGWILD) fail(AE_ERR_INVALID_ARG, "lockstep: only the rounds mode (AE_CE_HOGWILD) shards");
            if (o->comm) fail(AE_ERR_INVALID_ARG, "lockstep: shard %u has a communicator attached", q);
            if (o->dev.n != shards[0]->dev.n || o->dev.dim != shards[0]->dev.dim) fail(AE_ERR_INVALID_ARG, "lockstep: shards of different graphs");
            if (o->dev.node_lo != expect || o->dev.node_hi <= o->dev.node_lo) fail(AE_ERR_INVALID_ARG, "lockstep: the node ranges must tile [0, n) in order");
            expect = o->dev.node_hi;
        }
        if (expect != shards[0]->dev.n) fail(AE_ERR_INVALID_ARG, "lockstep: the node ranges must tile [0, n) in order");
        ce_node_gradient_iteration_lockstep(shards, world, nb_sample, grad_step, (uint32_t)iter, exchanges_per_batch);
        for (uint32_t q = 0; q < world; q++) check_err_flag(shards[q]);
    });
}

int32_t ae_entropy_optim_plan(ae_entropy_optim* o, uint64_t s_begin, uint64_t count, uint64_t iter, uint32_t* nodes7, float* w) {
    return guard([&] {
        require_device();
        if (!o || !nodes7 || count == 0) fail(AE_ERR_INVALID_ARG, "bad argument");
        DevBuf<uint32_t> dn(count * 7);
        DevBuf<float> dw(count);
        launch_plan(o, o->sample_offset + s_begin, count, (uint32_t)iter, dn.p, dw.p);
        dn.download(nodes7, count * 7);
        if (w) dw.download(w, count);
        check_err_flag(o);
    });
}

int32_t ae_entropy_optim_samples_drawn(ae_entropy_optim* o, uint64_t* samples, uint32_t* rounds) {
    return guard([&] {
        require_device();
        if (!o) fail(AE_ERR_INVALID_ARG, "null argument");
        unsigned long long h = 0;
        if (o->sample_counter.n) {
            std::vector<unsigned long long> hc = o->sample_counter.to_host();
            for (auto x : hc) h += x;
        }
        if (samples) *samples = h;
        if (rounds) *rounds = o->rounds;
    });
}

int32_t ae_entropy_optim_kernel_time(ae_entropy_optim* o, double* avg_ms, uint64_t* launches) {
    return guard([&] {
        require_device();
        if (!o || !avg_ms || !launches) fail(AE_ERR_INVALID_ARG, "null argument");
        sync();
        double tot = o->events_folded_ms;
        const uint64_t folded = o->events_folded;
        o->events_folded_ms = 0.;
        o->events_folded = 0;
        for (auto& e : o->events) {
            float ms = 0.f;
            AE_HIP(hipEventElapsedTime(&ms, e.first, e.second));
            tot += ms;
            (void)hipEventDestroy(e.first);
            (void)hipEventDestroy(e.second);
        }
        *launches = o->events.size() + folded;
        *avg_ms = *launches ? tot / (double)*launches : 0.;
        o->events.clear();
        check_err_flag(o);
    });
}

int32_t ae_entropy_optim_dataflow_time(ae_entropy_optim* o, double* avg_ms, uint64_t* launches) {
    return guard([&] {
        require_device();
        if (!o || !avg_ms || !launches) fail(AE_ERR_INVALID_ARG, "null argument");
        sync();
        double tot = 0.;
        for (auto& e : o->df_events) {
            float ms = 0.f;
            AE_HIP(hipEventElapsedTime(&ms, e.first, e.second));
            tot += ms;
            (void)hipEventDestroy(e.first);
            (void)hipEventDestroy(e.second);
        }
        *launches = o->df_events.size();
        *avg_ms = o->df_events.empty() ? 0. : tot / (double)o->df_events.size();
        o->df_events.clear();
    });
}

int32_t ae_entropy_optim_get_scales(const ae_entropy_optim* o, float* emb_scale) {
    return guard([&] {
        if (!o || !emb_scale) fail(AE_ERR_INVALID_ARG, "null argument");
        o->emb_scale.download(emb_scale, o->dev.n);
    });
}
int32_t ae_entropy_optim_get_embedded(const ae_entropy_optim* o, float* y) {
    return guard([&] {
        if (!o || !y) fail(AE_ERR_INVALID_ARG, "null argument");
        AE_HIP(hipMemcpy2DAsync(y, sizeof(float) * o->params.asked_dim, o->y.p, sizeof(float) * o->dev.dim, sizeof(float) * o->params.asked_dim,
                                o->dev.n, hipMemcpyDeviceToHost, stream()));
        sync();
    });
}
int32_t ae_entropy_optim_device_coords(ae_entropy_optim* o, void** d_y, uint64_t* n, uint64_t* dim) {
    return guard([&] {
        if (!o || !d_y) fail(AE_ERR_INVALID_ARG, "null argument");
        *d_y = (void*)o->y.p;
        if (n) *n = o->dev.n;
        if (dim) *dim = o->dev.dim;
    });
}

// entropy_optimize, embedder.rs:794-904
int32_t ae_entropy_optimize(const ae_kgraph* g, const ae_node_params* np, const ae_embedder_params* params, const float* y0,
                            float* y, double* ce_before, double* ce_after) {
    ae_entropy_optim* o = nullptr;
    int32_t rc;
    std::vector<uint32_t> hub;
    if (params && params->hubness_weighting && g) {
        hub.resize(g->n);
        rc = ae_kgraph_hubness(g, hub.data());
        if (rc) return rc;
    }
    rc = ae_entropy_optim_create(g, np, params, y0, hub.empty() ? nullptr : hub.data(), 0, g ? g->n : 0, &o);
    if (rc) return rc;
    auto cleanup = [&](int32_t code) { ae_entropy_optim_destroy(o); return code; };
    double ce = 0.;
    if ((rc = ae_entropy_optim_ce(o, &ce))) return cleanup(rc);  // :846
    if (ce_before) *ce_before = ce;
    uint64_t nnz = 0;
    ae_entropy_optim_get_nb_edges(o, &nnz);
    const uint64_t nb_sample = params->nb_sampling_by_edge * nnz;  // :858
    for (uint64_t iter = 1; iter <= params->nb_grad_batch; iter++) {  // :873
        const double step = params->grad_step * (1. - (double)iter / (double)params->nb_grad_batch);  // :875
        if ((rc = ae_entropy_optim_gradient_iteration(o, nb_sample, step, iter))) return cleanup(rc);
    }
    if ((rc = ae_entropy_optim_ce(o, &ce))) return cleanup(rc);  // :885
    if (ce_after) *ce_after = ce;
    if (y && (rc = ae_entropy_optim_get_embedded(o, y))) return cleanup(rc);
    double ms; uint64_t cnt;
    if ((rc = ae_entropy_optim_kernel_time(o, &ms, &cnt))) return cleanup(rc);
    return cleanup(AE_OK);
}

}  // extern "C"
