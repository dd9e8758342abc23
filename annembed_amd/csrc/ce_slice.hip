// ce_slice.hip -- AE_CE_SLICED: the CE gradient batch (gradient_iteration_threaded, src/embedder.rs:1311-1315) as a
// TIME-SLICED execution on CONFLICT-FREE MATCHINGS -- the faithful mode for graphs of any size (throughput-bound).
//
// What it keeps of the reference: every sample is applied to the CURRENT rows of both its end points with one
// gradient (embedder.rs:1228-1239), the samples come in an i.i.d. order, the reference's f64 scalars; the five negatives are
// read as the memory system has them (at most one launch old).  What it gives up: reproducibility sample by sample.
//
// How.  The i.i.d. edge draws of a batch are a Poisson process per edge (as in ce_event.hip): edge e fires c_e ~ Poisson(mu_e)
// times at i.i.d. uniform times.  The events of a batch are generated edge by edge (count -> scan -> fill), their times cut into
// thin SLICES (about half an event per node and slice).  Which events of a slice may run side by side is a property of the
// GRAPH, not of the draws: two samples conflict when their edges share a node.  So the edges are coloured ONCE per graph
// (sl_color_*: a proper edge colouring of the kNN graph seen as an undirected multigraph, greedy in parallel rounds) -- every
// colour class is a matching.  The events are bucketed by (slice, class) with one radix sort, classes in an order drawn afresh
// for every slice, and a STEP = the events of one class in one slice is one launch of sl_direct_kernel: no two of its events
// share a row, so every lane reads y_i and y_j, applies the attraction to both and the five repulsions to y_i exactly as
// embedder.rs:1207-1301 and writes both -- no ownership marks, no retries, no pending lists.  Within a slice the order of two
// events that share a node is the order of their classes, i.e. uniformly random; events of one class never share a node.
//
// Edges that find no colour below the cap (the in-edges of hubs beyond ~the cap: a node with d incident edges needs d classes)
// form the OVERFLOW class of every slice, executed optimistically after the slice's matchings (the previous form of this mode):
//   * every pending event marks its two rows in an owner array with its own id (plain stores: the last writer wins);
//   * the events that find their id on BOTH rows run, the others are deferred to the next pass (later passes: with probability
//     1/2 per pass, which breaks repeating stand-offs), what is left after the slice's passes joins the next slice, the batch
//     ends with passes until nothing is pending (hubs serialise as they do in the reference: one event per row at a time).
#include "ce_node_common.h"
#include "ce_sample_math.h"

#include <rocprim/rocprim.hpp>

#include <chrono>
#include <random>

using namespace ae;

namespace ae {
void sort_pairs_u32_u32(uint32_t* d_keys_in, uint32_t* d_keys_out, uint32_t* d_vals_in, uint32_t* d_vals_out, uint64_t count, unsigned end_bit);
}

namespace {

constexpr uint32_t kTagSlCount = 0xFFFF0031u, kTagSlTime = 0xFFFF0032u, kTagSlNeg = 0xFFFF0033u, kTagSlCoin = 0xFFFF0034u, kTagSlColor = 0xFFFF0035u;
// the pending list is kept as kSub sub-lists with a counter each: appends (one atomic per WORKGROUP) spread over kSub addresses --
// one counter serialises at ~12 ns per atomic, which with one atomic per wave was 2/3 of a pass at the C3 shape
constexpr int kSub = 16;
constexpr uint32_t kMaxClasses = 64;        // colours are bits of a 64-bit mask per node
constexpr uint8_t kNoColor = 0xFFu, kOverflowColor = 0xFEu;

struct EdgeRec {       // per edge, 16 bytes (colouring, event generation)
    uint32_t j;        // target
    float w;           // probability
    float s_src;       // embedded scale of the source
    uint32_t src;      // source
};
// An event in the sorted arrays: 8 bytes {source << 5 | slot of the edge in the source's row, target} -- the rows of both end
// points and the source's static record are requested in ONE hop after the (coalesced) event load.
struct Event {
    uint32_t im, j;
};
struct Pending {       // a pending event of the overflow class: 16 bytes, read and written coalesced
    uint32_t idx, im, j, pad;
};

struct SliceArgs {
    CeDev c;
    const float* srec;          // per node: static record of SREC floats {embedded scale, neighbour ids, edge probabilities}
    const Event* ev;            // the batch segment's events sorted by (slice, class)
    uint32_t f0, f1;            // the slice's overflow events = [f0, f1) of ev
    uint32_t* owner;            // [2][n]
    Pending* lists;             // [3][kSub][cap]: pending events, in kSub independent sub-lists (cap entries each)
    int tile;                   // 1: the negatives of this pass are drawn from a tile of rows staged in LDS (see sl_exec_kernel)
    uint32_t* counts;           // [3][kSub]
    uint64_t cap;
    uint32_t key;               // (batch << 12) | segment
    uint32_t pass_seq;          // running pass number of the batch (RNG key of the back-off coin)
    int src_list, dst_list, zero_list, owner_chk, owner_mark;
    int backoff;                // 1: a deferred event marks only with probability 1/2
    double step;
    unsigned long long* done_counter;   // [1024] spread counters of executed samples; [1024] = overflow flag
};

struct DirectArgs {
    CeDev c;
    const float* srec;
    const Event* ev;
    uint32_t begin, end;        // the step's events = [begin, end) of ev, sorted by edge (repeats of an edge adjacent)
    uint32_t ept;               // events per thread
    uint32_t key;               // (batch << 12) | segment
    uint32_t step_seq;          // running step number of the batch (RNG key of the tile windows)
    int tile;
    int dbg;                    // debug knob AE_SL_DBG (measurement only): 1 no arithmetic, 2 no stores, 4 no negatives, 8 no static record
    double step;
    unsigned long long* done_counter;
};

// ------------------------------------------------------------------------------------------------------------------
// edge colouring (once per graph)
// ------------------------------------------------------------------------------------------------------------------
// One round: every uncoloured edge proposes the lowest colour free at both its end points and bids for both nodes with a
// random priority; the edge that holds the minimum bid on BOTH nodes commits.  At most one edge commits per node and round, so
// the masks an edge read are still valid when it commits: the result is a proper colouring, equal to a sequential greedy
// colouring in some order (<= deg(i) + deg(j) - 1 colours, in practice max degree + 1 ... + 3).  A bid carries the round in
// its top byte (later rounds bid lower), so the bid array is never cleared.
__global__ void __launch_bounds__(256) sl_color_propose_kernel(uint64_t nnz, const EdgeRec* __restrict__ erec, const uint8_t* __restrict__ color,
                                                               const unsigned long long* __restrict__ used, unsigned long long* __restrict__ bid,
                                                               uint8_t* __restrict__ prop, uint32_t round, uint32_t hkey, unsigned long long capmask) {
    const uint64_t e = blockIdx.x * 256ull + threadIdx.x;
    if (e >= nnz || color[e] != kNoColor) return;
    const EdgeRec r = erec[e];
    const unsigned long long avail = ~(used[r.src] | used[r.j]) & capmask;
    if (!avail || r.src == r.j) { prop[e] = kOverflowColor; return; }
    prop[e] = (uint8_t)__builtin_ctzll(avail);
    const unsigned long long v = ((unsigned long long)(0xFFu - round) << 56) | ((unsigned long long)(pcg_hash((uint32_t)e ^ hkey) >> 8) << 32) | (uint32_t)e;
    atomicMin(&bid[r.src], v);
    atomicMin(&bid[r.j], v);
}
__global__ void __launch_bounds__(256) sl_color_commit_kernel(uint64_t nnz, const EdgeRec* __restrict__ erec, uint8_t* __restrict__ color,
                                                              unsigned long long* __restrict__ used, const unsigned long long* __restrict__ bid,
                                                              const uint8_t* __restrict__ prop, unsigned long long* __restrict__ remaining) {
    const uint64_t e = blockIdx.x * 256ull + threadIdx.x;
    bool left = false;
    if (e < nnz && color[e] == kNoColor) {
        const uint8_t p = prop[e];
        if (p == kOverflowColor) {
            color[e] = kOverflowColor;
        } else {
            const EdgeRec r = erec[e];
            if ((uint32_t)bid[r.src] == (uint32_t)e && (uint32_t)bid[r.j] == (uint32_t)e) {
                color[e] = p;
                atomicOr(&used[r.src], 1ull << p);
                atomicOr(&used[r.j], 1ull << p);
            } else {
                left = true;
            }
        }
    }
    // (one counter took a million same-address atomics per round: 8.8 ms of the commit kernel at the C4 shape, 0.3 s per handle)
    const unsigned long long m = __ballot(left);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(&remaining[(blockIdx.x * 4u + (threadIdx.x >> 6)) & 1023u], (unsigned long long)__popcll(m));
}
__global__ void __launch_bounds__(256) sl_color_giveup_kernel(uint64_t nnz, uint8_t* __restrict__ color) {
    const uint64_t e = blockIdx.x * 256ull + threadIdx.x;
    if (e < nnz && color[e] == kNoColor) color[e] = kOverflowColor;
}
// probability mass of every class ([kMaxClasses] = overflow), f64 atomics on LDS partials
__global__ void __launch_bounds__(256) sl_class_mass_kernel(uint64_t nnz, const EdgeRec* __restrict__ erec, const uint8_t* __restrict__ color, double* __restrict__ mass) {
    __shared__ double s_m[kMaxClasses + 1];
    for (int t = threadIdx.x; t <= (int)kMaxClasses; t += 256) s_m[t] = 0.;
    __syncthreads();
    for (uint64_t e = blockIdx.x * 256ull + threadIdx.x; e < nnz; e += (uint64_t)gridDim.x * 256ull) {
        const uint8_t cl = color[e];
        atomicAdd(&s_m[cl < kMaxClasses ? cl : kMaxClasses], (double)erec[e].w);
    }
    __syncthreads();
    for (int t = threadIdx.x; t <= (int)kMaxClasses; t += 256)
        if (s_m[t] != 0.) atomicAdd(&mass[t], s_m[t]);
}
// classes >= cut become overflow; per node: the probability mass of its overflow edges (sizes the pending lists)
__global__ void __launch_bounds__(256) sl_color_cut_kernel(uint64_t nnz, const EdgeRec* __restrict__ erec, uint8_t* __restrict__ color, uint32_t cut,
                                                           float* __restrict__ node_ov) {
    const uint64_t e = blockIdx.x * 256ull + threadIdx.x;
    if (e >= nnz) return;
    uint8_t cl = color[e];
    if (cl >= cut) { cl = kOverflowColor; color[e] = cl; }
    if (cl == kOverflowColor) {
        const EdgeRec r = erec[e];
        atomicAdd(&node_ov[r.src], r.w);
        atomicAdd(&node_ov[r.j], r.w);
    }
}
// expected backlog of the overflow class: what the rows with more than `per_slice_cap` overflow events per slice cannot run
__global__ void __launch_bounds__(256) sl_backlog_kernel(uint64_t n, const float* __restrict__ node_ov, float per_node, float capacity, double* __restrict__ out) {
    double local = 0.;
    for (uint64_t v = blockIdx.x * 256ull + threadIdx.x; v < n; v += (uint64_t)gridDim.x * 256ull) {
        const float ex = node_ov[v] * per_node - capacity;
        if (ex > 0.f) local += (double)ex;
    }
    for (int off = 32; off > 0; off >>= 1) local += __shfl_xor(local, off);
    if ((threadIdx.x & 63) == 0 && local != 0.) atomicAdd(out, local);
}

// ------------------------------------------------------------------------------------------------------------------
// events of a batch segment
// ------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) sl_count_kernel(CeDev c, float unit, uint32_t key, uint32_t* __restrict__ cnt) {
    const uint64_t e = blockIdx.x * 256ull + threadIdx.x;
    if (e >= c.nnz) return;
    const uint32_t ck = round_hash_key(key, c.seed) ^ kTagSlCount;
    const float mu = unit * c.proba[e];
    const float u = edge_uniform(e, ck);
    float p = __expf(-mu), cdf = p;
    uint32_t k = 0;
    while (u >= cdf && k < 255u) {
        k++;
        p *= mu * (1.0f / (float)k);
        cdf += p;
    }
    cnt[e] = k;
}
// (step key, event) of every event, at the edge's offset.  step key = slice * (classes + 1) + position of the edge's class in the
// slice's class order (class_pos[slice][class]; overflow last).  The slices of an edge's events are i.i.d. uniform; `spread`: an
// edge fires at most once per slice -- a second event of the edge in a slice moves to the next one (its repeats would otherwise
// run back to back inside the step, with no other event of their end points in between; with thin slices this touches < 1 %
// of the events).
__global__ void __launch_bounds__(256) sl_fill_kernel(CeDev c, uint32_t key, const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ offs,
                                                      uint32_t n_slices, const EdgeRec* __restrict__ erec, const uint8_t* __restrict__ color,
                                                      const uint8_t* __restrict__ class_pos, uint32_t classes, int spread,
                                                      uint32_t* __restrict__ keys, Event* __restrict__ vals) {
    const uint64_t e = blockIdx.x * 256ull + threadIdx.x;
    if (e >= c.nnz) return;
    const uint32_t k = cnt[e], o = offs[e];
    if (!k) return;
    const uint32_t tk = pcg_hash(round_hash_key(key, c.seed) ^ kTagSlTime);
    const uint8_t cl = color[e];
    const EdgeRec er = erec[e];
    uint64_t ib = c.uniform_k ? (uint64_t)er.src * c.uniform_k : c.indptr[er.src];
    const Event evv{(er.src << 5) | (uint32_t)(e - ib), er.j};
    constexpr uint32_t kSortMax = 24;
    __shared__ uint32_t s_sl[kSortMax * 256];  // [r][thread]: the slices of this thread's edge (dynamic indexing: LDS, not scratch)
    uint32_t* sl = s_sl + threadIdx.x;
#define SL(r) sl[(r) * 256u]
    const bool sorted = spread && k > 1 && k <= kSortMax && k <= n_slices;
    if (sorted) {  // the k slices in ascending order (insertion sort), then made strictly increasing
        for (uint32_t r = 0; r < k; r++) {
            const uint32_t s = __umulhi(pcg_hash((pcg_hash((uint32_t)e) + r * 0x9E3779B9u) ^ tk), n_slices);
            uint32_t q = r;
            while (q > 0 && SL(q - 1) > s) { SL(q) = SL(q - 1); q--; }
            SL(q) = s;
        }
        for (uint32_t r = 1; r < k; r++) SL(r) = max(SL(r), SL(r - 1) + 1u);
        // what ran past the end of the segment is pulled back from the top
        if (SL(k - 1) >= n_slices) {
            SL(k - 1) = n_slices - 1u;
            for (uint32_t r = k - 1; r > 0 && SL(r - 1) >= SL(r); r--) SL(r - 1) = SL(r) - 1u;
        }
    }
    for (uint32_t r = 0; r < k; r++) {
        const uint32_t s = sorted ? SL(r) : __umulhi(pcg_hash((pcg_hash((uint32_t)e) + r * 0x9E3779B9u) ^ tk), n_slices);
        const uint32_t pos = cl < classes ? (uint32_t)class_pos[s * classes + cl] : classes;
        keys[o + r] = s * (classes + 1u) + pos;
        vals[o + r] = evv;
    }
#undef SL
}
__global__ void sl_sptr_kernel(const uint32_t* __restrict__ keys, uint32_t total, uint32_t n_keys, uint32_t* __restrict__ sptr) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s > n_keys) return;
    uint32_t lo = 0, hi = total;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (keys[mid] < s) lo = mid + 1; else hi = mid;
    }
    sptr[s] = lo;
}
__global__ void sl_edge_rec_kernel(CeDev c, EdgeRec* __restrict__ out) {
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i >= c.n) return;
    uint64_t b, e1;
    if (c.uniform_k) { b = i * c.uniform_k; e1 = b + c.uniform_k; }
    else { b = c.indptr[i]; e1 = c.indptr[i + 1]; }
    const float s = c.emb_scale[i];
    for (uint64_t e = b; e < e1; e++) out[e] = EdgeRec{c.nbr[e], c.proba[e], s, (uint32_t)i};
}
// static record of a node: SREC floats = {embedded scale, KP neighbour ids (padded with ~0), KP edge probabilities}, KP = (SREC - 1) / 2
__global__ void sl_static_rec_kernel(CeDev c, uint32_t srec, float* __restrict__ out) {
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i >= c.n) return;
    uint64_t b;
    uint32_t k;
    if (c.uniform_k) { b = i * c.uniform_k; k = c.uniform_k; }
    else { b = c.indptr[i]; k = (uint32_t)(c.indptr[i + 1] - b); }
    const uint32_t kp = (srec - 1u) / 2u;
    float* r = out + i * srec;
    r[0] = c.emb_scale[i];
    for (uint32_t m = 0; m < kp; m++) {
        r[1 + m] = __uint_as_float(m < k ? c.nbr[b + m] : 0xFFFFFFFFu);
        r[1 + kp + m] = m < k ? c.proba[b + m] : 0.f;
    }
    for (uint32_t m = 1 + 2 * kp; m < srec; m++) r[m] = 0.f;
}

// ------------------------------------------------------------------------------------------------------------------
// memory access by lane groups
// ------------------------------------------------------------------------------------------------------------------
// The kernels below are bound by the NUMBER of memory requests, not by bytes (tools/ubench_rowgather.hip: ~55 G random requests/s
// whatever their width up to 64 bytes; a lane that loads a 32-byte row with two 16-byte instructions issues two).  A record of NF
// floats (a coordinate row of >= 8 columns, a static record) is therefore fetched by a GROUP of G = NF / 4 adjacent lanes: in
// step t every lane of the group loads its 16-byte piece of the record wanted by the group's lane t -- one request per record --
// and the pieces are handed to their owner through a wave-private LDS stage (a wave's LDS operations execute in program order:
// no workgroup barrier).  Stage rows are NF + 4 floats apart (bank spread).
using f4 = __attribute__((ext_vector_type(4))) float;
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
// issue: the group's loads into registers (pc[t] = this lane's piece of the record wanted by the group's lane t); land: through
// the stage to the owners.  Issue everything a sample needs first, land afterwards: one memory round trip, not one per record.
// value of the group's lane t (t is a constant after unrolling): DPP quad permutes for groups of 2 and 4 lanes (one VALU
// instruction), the LDS crossbar otherwise
template <int CTRL>
__device__ __forceinline__ uint32_t quad_perm(uint32_t x) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, CTRL, 0xF, 0xF, true);
}
template <int G>
__device__ __forceinline__ uint32_t group_bcast(uint32_t x, int t) {
    if constexpr (G == 1) {
        return x;
    } else if constexpr (G == 2) {
        return t == 0 ? quad_perm<0xA0>(x) : quad_perm<0xF5>(x);
    } else if constexpr (G == 4) {
        switch (t) {
            case 0: return quad_perm<0x00>(x);
            case 1: return quad_perm<0x55>(x);
            case 2: return quad_perm<0xAA>(x);
            default: return quad_perm<0xFF>(x);
        }
    } else {
        return (uint32_t)__shfl((int)x, ((int)(threadIdx.x & 63) & ~(G - 1)) + t);
    }
}
// issue: the group's loads into registers (pc[t] = this lane's piece of the record wanted by the group's lane t); land: through
// the stage to the owners.  Issue everything a sample needs first, land afterwards: one memory round trip, not one per record.
// idx < 2^31; the top bit of the broadcast word carries `want`.
template <int NF>
__device__ __forceinline__ void coop_issue(const float* __restrict__ base, uint32_t idx, bool want, f4 (&pc)[NF / 4]) {
    constexpr int G = NF / 4;
    const uint32_t sub = (threadIdx.x & 63) & (G - 1);
    // UNCONDITIONAL loads (a lane that wants nothing asks for record 0: one shared line): a load under `if (want)` makes the register
    // allocator merge the two paths with a copy behind the load -- an `s_waitcnt vmcnt(0)` in the middle of the issue phase, which
    // also drains the tile's loads (seen in the ISA of sl_direct_kernel<8,16>)
    const uint32_t word = want ? idx : 0u;
#pragma unroll
    for (int t = 0; t < G; t++) {
        const uint32_t wt = group_bcast<G>(word, t);
        pc[t] = *reinterpret_cast<const f4*>(base + (uint64_t)wt * NF + sub * 4u);
    }
}
template <int NF>
__device__ __forceinline__ void coop_land(const f4 (&pc)[NF / 4], float* stage) {
    constexpr int G = NF / 4, RS = NF + 4;
    const int lane = threadIdx.x & 63, sub = lane & (G - 1), gb = lane & ~(G - 1);
#pragma unroll
    for (int t = 0; t < G; t++) *reinterpret_cast<f4*>(stage + (gb + t) * RS + sub * 4) = pc[t];
    wave_lds_sync();
}
template <int NF>
__device__ __forceinline__ void coop_store(float* __restrict__ base, uint32_t idx, bool want, const float* stage) {
    constexpr int G = NF / 4, RS = NF + 4;
    const int lane = threadIdx.x & 63, sub = lane & (G - 1), gb = lane & ~(G - 1);
    const uint32_t word = idx | (want ? 0x80000000u : 0u);
    wave_lds_sync();
#pragma unroll
    for (int t = 0; t < G; t++) {
        const uint32_t wt = group_bcast<G>(word, t);
        if (wt & 0x80000000u)
            *reinterpret_cast<f4*>(base + (uint64_t)(wt & 0x7FFFFFFFu) * NF + (uint32_t)sub * 4u) = *reinterpret_cast<const f4*>(stage + (gb + t) * RS + sub * 4);
    }
    wave_lds_sync();
}
// rows: cooperative for 8 and 16 columns (2 / 4 lanes per row), one lane per row otherwise (<= 4 columns: one request anyway;
// 32 / 64 columns: rare, kept simple)
template <int DIM>
constexpr bool kCoopRow = DIM == 8 || DIM == 16;
template <int SREC>
constexpr bool kCoopRec = SREC <= 32;
template <int DIM, int SREC>
constexpr int kStageFloats = (kCoopRow<DIM> || kCoopRec<SREC>) ? 64 * ((((kCoopRow<DIM> ? DIM : 0) > (kCoopRec<SREC> ? SREC : 0)) ? DIM : SREC) + 4) : 1;

template <int DIM>
struct RowFetch {  // a coordinate row on its way to its lane
    f4 pc[kCoopRow<DIM> ? DIM / 4 : 1];
    __device__ __forceinline__ void issue(const float* __restrict__ y, uint32_t node, bool want, float* out) {
        if constexpr (kCoopRow<DIM>) coop_issue<DIM>(y, node, want, pc);
        else load_row<DIM>(y, want ? node : 0u, out);
    }
    __device__ __forceinline__ void land(float* stage, float* out) {
        if constexpr (kCoopRow<DIM>) {
            coop_land<DIM>(pc, stage);
            const float* p = stage + (threadIdx.x & 63) * (DIM + 4);
#pragma unroll
            for (int q = 0; q < DIM / 4; q++) {
                const f4 v = *reinterpret_cast<const f4*>(p + 4 * q);
                out[4 * q] = v.x; out[4 * q + 1] = v.y; out[4 * q + 2] = v.z; out[4 * q + 3] = v.w;
            }
            wave_lds_sync();
        }
    }
};
template <int DIM>
__device__ __forceinline__ void row_store(float* __restrict__ y, uint32_t node, bool want, float* stage, const float* in) {
    if constexpr (kCoopRow<DIM>) {
        float* p = stage + (threadIdx.x & 63) * (DIM + 4);
#pragma unroll
        for (int q = 0; q < DIM / 4; q++) {
            f4 v; v.x = in[4 * q]; v.y = in[4 * q + 1]; v.z = in[4 * q + 2]; v.w = in[4 * q + 3];
            *reinterpret_cast<f4*>(p + 4 * q) = v;
        }
        coop_store<DIM>(y, node, want, stage);
    } else {
        if (want) store_row<DIM>(y, node, in);
    }
}
// the source's static record: embedded scale, the neighbour ids (rejection test of the negatives), the sampled edge's probability
template <int SREC, int KREG>
struct RecFetch {
    static constexpr int KP = (SREC - 1) / 2;
    f4 pc[kCoopRec<SREC> ? SREC / 4 : 1];
    __device__ __forceinline__ void issue(const float* __restrict__ srec, uint32_t node, uint32_t m, bool want, float& scale, float& w, uint32_t (&nbr_reg)[KREG]) {
        if constexpr (kCoopRec<SREC>) {
            coop_issue<SREC>(srec, node, want, pc);
        } else {
            const float* p = srec + (uint64_t)(want ? node : 0u) * SREC;
            scale = p[0];
#pragma unroll
            for (int q = 0; q < KREG; q++) nbr_reg[q] = __float_as_uint(p[1 + q]);
            w = p[1 + KP + (want ? m : 0u)];
        }
    }
    __device__ __forceinline__ void land(float* stage, uint32_t m, bool want, float& scale, float& w, uint32_t (&nbr_reg)[KREG]) {
        if constexpr (kCoopRec<SREC>) {
            coop_land<SREC>(pc, stage);
            const float* p = stage + (threadIdx.x & 63) * (SREC + 4);
            scale = p[0];
#pragma unroll
            for (int q = 0; q < KREG; q++) nbr_reg[q] = __float_as_uint(p[1 + q]);
            w = p[1 + KP + (want ? m : 0u)];
            wave_lds_sync();
        }
    }
};

// ------------------------------------------------------------------------------------------------------------------
// one sample on rows held in registers
// ------------------------------------------------------------------------------------------------------------------
// tile of coordinate rows for the negatives of a crowded launch: kW windows of kL consecutive rows each, window starts uniform
// over the nodes (wrapping) and fresh per workgroup and launch, staged in LDS with coalesced loads.  A negative is then "window
// uniform, row uniform": every node has the same probability 1/n, as in embedder.rs:1121; the rows are as fresh as the launch
// (it started after every earlier launch's writes).  What differs from the reference: the negatives of the samples a workgroup
// runs in a launch come from the same kW windows (the marginals are exact, the joint law is not).
template <int DIM>
struct TileShape {
    // 256 rows (16 windows of 16 consecutive rows) serve the 5 x 256 draws of a workgroup: staging costs one coalesced row read per
    // sample instead of five random ones (a tile of 1024 rows costs as much as the gathers it replaces)
    static constexpr int kRows = DIM <= 16 ? 256 : 128;
    static constexpr int kL = 16;
    static constexpr int kW = kRows / kL;
    static constexpr int kRowBits = DIM <= 16 ? 8 : 7;
    static constexpr int kPieces = kRows * (DIM % 4 == 0 ? DIM / 4 : DIM) / 256;  // loads per thread: 16-byte pieces (single floats for 3 columns)
};
__device__ __forceinline__ uint32_t tile_window_start(uint32_t wkey, uint32_t w, uint32_t n) { return __umulhi(pcg_hash(wkey + w * 0x9E3779B9u), n); }

// The tile's loads are ISSUED at the start of the kernel (next to the event load) and LANDED in LDS when the sample's own loads
// are in flight: staging is off the critical path.  Hubness-weighted sampling (NodeSampler, embedder.rs:915-930: what
// examples/higgs.rs switches on) cannot use runs of consecutive rows: there every tile row is an independent draw of the alias
// table (one 8-byte look-up, then the row) -- a slot picked uniformly afterwards is again a draw of the reference's law; two
// requests per tile row instead of ten per sample.
template <int DIM>
struct TileFetch {
    using T = TileShape<DIM>;
    static constexpr int Q = DIM % 4 == 0 ? DIM / 4 : DIM;  // pieces per row
    f4 pc[T::kPieces];
    uint32_t node[T::kPieces];
    __device__ __forceinline__ void issue(const CeDev& c, uint32_t wkey, bool hub) {
#pragma unroll
        for (int z = 0; z < T::kPieces; z++) {
            const uint32_t x = (uint32_t)z * 256u + threadIdx.x, r = x / Q;
            if (hub) {
                const uint32_t w0 = pcg_hash(wkey + r * 0x9E3779B9u);
                const uint32_t xs = __umulhi(w0, (uint32_t)c.n);
                const float uu = (float)(pcg_hash(w0 ^ 0x9E3779B9u) >> 8) * (1.0f / 16777216.0f);
                const uint2 he = c.hub_tab[xs];
                node[z] = (uu < __uint_as_float(he.x)) ? xs : he.y;
            } else {
                node[z] = tile_window_start(wkey, r / T::kL, (uint32_t)c.n) + r % T::kL;
                node[z] -= node[z] >= (uint32_t)c.n ? (uint32_t)c.n : 0u;
            }
        }
#pragma unroll
        for (int z = 0; z < T::kPieces; z++) {
            const uint32_t x = (uint32_t)z * 256u + threadIdx.x, q = x % Q;
            if constexpr (DIM % 4 == 0) pc[z] = *reinterpret_cast<const f4*>(c.y + (uint64_t)node[z] * DIM + 4u * q);
            else pc[z].x = c.y[(uint64_t)node[z] * DIM + q];
        }
    }
    __device__ __forceinline__ void land(float* s_tile, uint32_t* s_tnode) {
#pragma unroll
        for (int z = 0; z < T::kPieces; z++) {
            const uint32_t x = (uint32_t)z * 256u + threadIdx.x;
            if constexpr (DIM % 4 == 0) *reinterpret_cast<f4*>(s_tile + 4u * x) = pc[z];
            else s_tile[x] = pc[z].x;
            if (x % Q == 0) s_tnode[x / Q] = node[z];
        }
        __syncthreads();
    }
};

// the five negatives (embedder.rs:1241-1253): uniform / NodeSampler (:927-930) draws, rejected when k = i, k = j or k in N(i)
// (nodeparam.rs:83-85; j is in N(i)).  TILE: a draw is a slot of the staged tile (one hash: window and row from its top bits), `out`
// receives the slots; otherwise node ids (uniform, or hubness-weighted through the alias table), eight candidates at a time so that
// the alias look-ups overlap.  Returns the number accepted (5 unless the graph is tiny).
template <int DIM, int KMAX, bool TILE>
__device__ __forceinline__ uint32_t draw_negatives(const CeDev& c, bool hub, const uint32_t* s_tnode, uint32_t nb, uint32_t i,
                                                   const uint32_t (&nbr_reg)[KMAX], uint32_t (&out)[5]) {
    using T = TileShape<DIM>;
    uint32_t got = 0;
#pragma unroll
    for (int g = 0; g < 5; g++) out[g] = TILE ? 0u : i;
    for (uint32_t round = 0; round < 8u && got < 5u; round++) {
        uint32_t cand[8], slot[8];
        if constexpr (TILE) {
#pragma unroll
            for (int z = 0; z < 8; z++) {
                const uint32_t row = pcg_hash(nb + (round * 8u + (uint32_t)z) * 0x9E3779B9u) >> (32 - T::kRowBits);
                cand[z] = s_tnode[row];
                slot[z] = row;
            }
        } else if (hub) {
            uint32_t xs[8], al[8];
            float od[8], uu[8];
#pragma unroll
            for (int z = 0; z < 8; z++) {
                const uint32_t w0 = pcg_hash(nb + (round * 8u + (uint32_t)z) * 0x9E3779B9u);
                xs[z] = __umulhi(w0, (uint32_t)c.n);
                uu[z] = (float)(pcg_hash(w0 ^ 0x9E3779B9u) >> 8) * (1.0f / 16777216.0f);
                const uint2 he = c.hub_tab[xs[z]];
                od[z] = __uint_as_float(he.x);
                al[z] = he.y;
            }
#pragma unroll
            for (int z = 0; z < 8; z++) { cand[z] = (uu[z] < od[z]) ? xs[z] : al[z]; slot[z] = cand[z]; }
        } else {
#pragma unroll
            for (int z = 0; z < 8; z++) { cand[z] = __umulhi(pcg_hash(nb + (round * 8u + (uint32_t)z) * 0x9E3779B9u), (uint32_t)c.n); slot[z] = cand[z]; }  // :1121
        }
#pragma unroll
        for (int z = 0; z < 8; z++) {
            uint32_t acc = cand[z] ^ i;
#pragma unroll
            for (int m = 0; m < KMAX; m++) { const uint32_t x = nbr_reg[m] ^ cand[z]; acc = x < acc ? x : acc; }
            const bool ok = acc != 0u && got < 5u;
#pragma unroll
            for (int g = 0; g < 5; g++) out[g] = (ok && got == (uint32_t)g) ? slot[z] : out[g];  // (static indexing keeps `out` in registers)
            got += ok ? 1u : 0u;
        }
    }
    return got;
}

// The sample's scalar arithmetic in f32 (this mode's default; AE_SL_F64 = the reference's f64 scalars, :1207-1229): the same
// formulas with hardware reciprocals.  This mode is validated statistically -- the rounding of a scalar coefficient (1e-7) is six
// orders of magnitude below the sampling noise -- and the 24 dependent f64 divisions of a sample were a quarter of a batch.
template <int DIM>
__device__ __forceinline__ void attract_f32(float* yi, float* yj, float* grad, float w, float inv_s2, float b, float step) {
    float acc = 0.f;
#pragma unroll
    for (int t = 0; t < DIM; t++) { grad[t] = 0.f; const float df = yi[t] - yj[t]; acc += df * df; }
    const float d = acc * inv_s2;
    if (d > 0.f) {
        const float coeff = b == 1.f ? 2.0f * inv_s2 * rcp(1.0f + d) : 2.0f * b * rcp(1.0f + __powf(d, b)) * __powf(d, b - 1.0f) * inv_s2;
        const float rep = rcp(fmaxf(d * d, 1.0f / kProbaMin));
        const float cf = fmaxf(step * coeff * (-w + (1.f - w) * rep), -0.49f);
#pragma unroll
        for (int t = 0; t < DIM; t++) grad[t] = (yj[t] - yi[t]) * cf;
    }
#pragma unroll
    for (int t = 0; t < DIM; t++) { yi[t] -= grad[t]; yj[t] += grad[t]; }
}
template <int DIM>
__device__ __forceinline__ void repulse_f32(float* yi, const float* yk, float* grad, float inv_s2, float b, float step) {
    float ak = 0.f;
#pragma unroll
    for (int t = 0; t < DIM; t++) { const float df = yi[t] - yk[t]; ak += df * df; }
    const float d = ak * inv_s2;
    if (ak > 0.f) {
        const float coeff = b == 1.f ? 2.0f * inv_s2 * rcp(1.0f + d) : 2.0f * b * rcp(1.0f + __powf(d, b)) * __powf(d, b - 1.0f) * inv_s2;
        const float cf = fminf(step * coeff * rcp(fmaxf(d * d, 1.0f / 16.0f)), 2.0f);
#pragma unroll
        for (int t = 0; t < DIM; t++) grad[t] = (yk[t] - yi[t]) * cf;
    }  // else: `gradient` keeps its previous value, as in the reference
#pragma unroll
    for (int t = 0; t < DIM; t++) yi[t] -= grad[t];
}

// ce_optim_edge_shannon (embedder.rs:1167-1302) on yi / yj in registers: the attraction (one gradient, both ends) and the
// repulsions from the `got` drawn negatives (tile slots or node ids in `neg`); a negative's row may be rewritten during this launch
// by its owner: at most one launch old
template <int DIM, bool F64, bool TILE>
__device__ __forceinline__ void run_sample(const CeDev& c, const float* s_tile, float* yi, float* yj, float w, float scale_f,
                                           double step, const uint32_t (&neg)[5], uint32_t got) {
    float grad[DIM];
    const double scale = (double)scale_f;
    const float inv_s2 = rcp(scale_f * scale_f), bf = (float)c.b, stepf = (float)step;
    auto att = [&]() {
        if constexpr (F64) sample_attract<DIM>(yi, yj, grad, w, scale, c.b, step);  // :1207-1238
        else attract_f32<DIM>(yi, yj, grad, w, inv_s2, bf, stepf);
    };
    auto rep = [&](const float* yk) {
        if constexpr (F64) sample_repulse<DIM>(yi, yk, grad, scale, c.b, step);  // :1267-1297
        else repulse_f32<DIM>(yi, yk, grad, inv_s2, bf, stepf);
    };
    auto fetch = [&](uint32_t x, float* row) {
        if constexpr (TILE) {
            if constexpr (DIM % 4 == 0) {
#pragma unroll
                for (int q = 0; q < DIM / 4; q++) {
                    const f4 v = *reinterpret_cast<const f4*>(s_tile + x * DIM + 4 * q);
                    row[4 * q] = v.x; row[4 * q + 1] = v.y; row[4 * q + 2] = v.z; row[4 * q + 3] = v.w;
                }
            } else {
#pragma unroll
                for (int t2 = 0; t2 < DIM; t2++) row[t2] = s_tile[x * DIM + t2];
            }
        } else {
            load_row<DIM>(c.y, x, row);
        }
    };
    if constexpr (DIM <= 16) {
        float nrow[5][DIM];
#pragma unroll
        for (int g = 0; g < 5; g++) fetch(neg[g], nrow[g]);
        att();
#pragma unroll
        for (int g = 0; g < 5; g++)
            if ((uint32_t)g < got) rep(nrow[g]);
    } else {  // wide rows: one negative at a time (5 x 64 registers do not exist)
        att();
        for (uint32_t g = 0; g < got; g++) {
            uint32_t x = neg[0];
#pragma unroll
            for (int q = 1; q < 5; q++) x = g == (uint32_t)q ? neg[q] : x;
            float nrow[DIM];
            fetch(x, nrow);
            rep(nrow);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// a step: the events of one colour class in one slice -- no two of them share a row
// ------------------------------------------------------------------------------------------------------------------
template <int DIM, int SREC, bool F64, bool TILE>
__global__ void __launch_bounds__(256) sl_direct_kernel(DirectArgs a) {
    using T = TileShape<DIM>;
    constexpr int KREG = (SREC - 1) / 2 < 32 ? (SREC - 1) / 2 : 32;
    __shared__ __attribute__((aligned(16))) float s_tile[TILE ? T::kRows * DIM : 4];
    __shared__ __attribute__((aligned(16))) float s_stage[4 * kStageFloats<DIM, SREC>];
    __shared__ uint32_t s_tnode[TILE ? T::kRows : 1];
    const CeDev c = a.c;
    const bool hub = c.hub_odds != nullptr;
    float* stage = s_stage + (threadIdx.x >> 6) * kStageFloats<DIM, SREC>;
    const uint32_t nkey = pcg_hash((uint32_t)c.seed ^ a.key ^ kTagSlNeg);
    const uint32_t wkey = pcg_hash(nkey + a.step_seq * 0x85EBCA6Bu) + blockIdx.x * 64u;
    uint32_t done = 0;
    const uint32_t base = a.begin + blockIdx.x * 256u * a.ept;
    // one pass over 256 events; FIRST (a compile-time tag): the pass that also stages the tile.  Kept out of the loop below so that the
    // wait for the event load counts the tile's loads behind it (`vmcnt(pieces)`); with a run-time `r == 0` the compiler has to assume
    // the path without them and waits for everything: the tile's random reads became a hop of their own in front of the rows.
    auto pass = [&](uint32_t r, auto first_tag) {
        constexpr bool FIRST = decltype(first_tag)::value;
        const uint32_t p = base + r * 256u + threadIdx.x;
        // hop 1: the event (coalesced) and, beside it, this thread's share of the tile
        bool act = p < a.end;
        Event e{0u, 0u};
        uint32_t prev_im = 0xFFFFFFFFu, next_im = 0xFFFFFFFFu;
        if (act) {
            e = a.ev[p];
            if (p > a.begin) prev_im = a.ev[p - 1].im;
            if (p + 1 < a.end) next_im = a.ev[p + 1].im;
        }
        TileFetch<DIM> ft;
        if constexpr (TILE && FIRST) ft.issue(c, wkey, hub);
        uint32_t run = 1;
        if (act && prev_im == e.im) act = false;  // a repeat of the previous event's edge: its first lane runs the whole run
        else if (act && next_im == e.im) { run = 2; while (p + run < a.end && a.ev[p + run].im == e.im) run++; }
        const uint32_t i = e.im >> 5;
        float yi[DIM], yj[DIM], scale_f = 1.f, w = 0.f;
        uint32_t nbr_reg[KREG];
        // hop 2: the source's record and both rows are requested together, then handed to their lanes
        RecFetch<SREC, KREG> fr;
        RowFetch<DIM> fi, fj;
        const bool want_rec = act && !(a.dbg & 8);
        fr.issue(a.srec, i, e.im & 31u, want_rec, scale_f, w, nbr_reg);
        fi.issue(c.y, i, act, yi);     // :1185
        fj.issue(c.y, e.j, act, yj);   // :1186
        if constexpr (TILE && FIRST) ft.land(s_tile, s_tnode);
        fr.land(stage, e.im & 31u, want_rec, scale_f, w, nbr_reg);
        fi.land(stage, yi);
        fj.land(stage, yj);
        if (a.dbg & 8) { for (int q = 0; q < KREG; q++) nbr_reg[q] = 0xFFFFFFFFu; w = 0.5f; scale_f = 1.f; }
        if (act) {
            for (uint32_t q = 0; q < run; q++) {
                uint32_t neg[5];
                uint32_t got = draw_negatives<DIM, KREG, TILE>(c, hub, s_tnode, pcg_hash(nkey + (p + q)), i, nbr_reg, neg);
                if (a.dbg & 4) { got = 0; for (int g = 0; g < 5; g++) neg[g] = TILE ? 0u : i; }
                if (!(a.dbg & 1)) run_sample<DIM, F64, TILE>(c, s_tile, yi, yj, w, scale_f, a.step, neg, got);
                else if (got == 77u) yi[0] += (float)neg[0];
            }
            done += run;
        }
        if (!(a.dbg & 2)) {
            row_store<DIM>(c.y, e.j, act, stage, yj);  // :1239
            row_store<DIM>(c.y, i, act, stage, yi);    // :1301
        } else if (yi[0] == 1.2345e-30f && yj[0] == 3.4e-30f) {
            row_store<DIM>(c.y, i, act, stage, yi);
        }
    };
    if (base < a.end) pass(0u, std::true_type{});  // (uniform over the workgroup)
    for (uint32_t r = 1; r < a.ept && base + r * 256u < a.end; r++) pass(r, std::false_type{});
    for (int off = 32; off > 0; off >>= 1) done += __shfl_xor(done, off);
    if ((threadIdx.x & 63) == 0 && done) atomicAdd(&a.done_counter[(blockIdx.x * 4u + (threadIdx.x >> 6)) & 1023u], (unsigned long long)done);
}

// ------------------------------------------------------------------------------------------------------------------
// the overflow class of a slice: optimistic passes
// ------------------------------------------------------------------------------------------------------------------
// start of a slice: pending list = what the previous slice left + the slice's own overflow events; every one marks its two rows.
// Sub-list s (blockIdx.y) takes the leftover sub-list s and every kSub-th event of the slice.
__global__ void __launch_bounds__(256) sl_mark_kernel(SliceArgs a) {
    const uint32_t sub = blockIdx.y;
    const uint32_t left = min(a.counts[a.src_list * kSub + sub], (uint32_t)a.cap);
    const uint32_t f0 = a.f0, f1 = a.f1;
    const uint32_t fresh = f1 > f0 + sub ? (f1 - f0 - sub + (uint32_t)kSub - 1u) / (uint32_t)kSub : 0u;
    const uint32_t total = left + fresh;
    const uint64_t so = ((uint64_t)a.src_list * kSub + sub) * a.cap, dof = ((uint64_t)a.dst_list * kSub + sub) * a.cap;
    for (uint64_t t = blockIdx.x * 256ull + threadIdx.x; t < total; t += (uint64_t)gridDim.x * 256ull) {
        Pending p;
        if (t < left) {
            p = a.lists[so + t];
        } else {
            p.idx = f0 + sub + (uint32_t)(t - left) * (uint32_t)kSub;
            const Event e = a.ev[p.idx];
            p.im = e.im; p.j = e.j; p.pad = 0;
        }
        if (t < a.cap) {
            a.lists[dof + t] = p;
            a.owner[(uint64_t)a.owner_mark * a.c.n + (p.im >> 5)] = p.idx;
            a.owner[(uint64_t)a.owner_mark * a.c.n + p.j] = p.idx;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        a.counts[a.dst_list * kSub + sub] = total < a.cap ? total : (uint32_t)a.cap;  // (cap is sized so that this never truncates; flagged otherwise)
        if (total > a.cap) atomicOr(reinterpret_cast<unsigned int*>(a.done_counter + 1024), 1u);
        a.counts[a.zero_list * kSub + sub] = 0;
    }
}

// one pass: the pending events that own both their rows run, the others go to the next list and mark for the next pass
template <int DIM, int SREC, bool F64, bool TILE>
__global__ void __launch_bounds__(256) sl_exec_kernel(SliceArgs a) {
    using T = TileShape<DIM>;
    constexpr int KREG = (SREC - 1) / 2 < 32 ? (SREC - 1) / 2 : 32;
    __shared__ __attribute__((aligned(16))) float s_tile[TILE ? T::kRows * DIM : 4];
    __shared__ __attribute__((aligned(16))) float s_stage[4 * kStageFloats<DIM, SREC>];
    __shared__ uint32_t s_tnode[TILE ? T::kRows : 1];
    __shared__ uint32_t s_wave_cnt[4], s_base;
    const CeDev c = a.c;
    const uint32_t sub = blockIdx.y, dsub = (blockIdx.x + blockIdx.y) % (uint32_t)kSub;
    const uint32_t total = min(a.counts[a.src_list * kSub + sub], (uint32_t)a.cap);  // (an overflowing append is flagged, never read back)
    const uint64_t so = ((uint64_t)a.src_list * kSub + sub) * a.cap, dof = ((uint64_t)a.dst_list * kSub + dsub) * a.cap;
    const uint32_t* own_chk = a.owner + (uint64_t)a.owner_chk * c.n;
    uint32_t* own_mark = a.owner + (uint64_t)a.owner_mark * c.n;
    const bool hub = c.hub_odds != nullptr;
    const uint32_t nkey = pcg_hash((uint32_t)c.seed ^ a.key ^ kTagSlNeg);
    const uint32_t wkey = pcg_hash(nkey + a.pass_seq * 0x9E3779B9u) + (blockIdx.x * (uint32_t)kSub + sub) * 64u;
    float* stage = s_stage + (threadIdx.x >> 6) * kStageFloats<DIM, SREC>;
    unsigned long long done = 0;
    // one trip over 256 pending events; FIRST (compile-time, see sl_direct_kernel): the trip that also stages the tile -- its loads are
    // issued behind the ownership checks and travel while the rows are requested
    auto trip = [&](uint64_t t0, auto first_tag) {
        constexpr bool FIRST = decltype(first_tag)::value;
        const uint64_t t = t0 + threadIdx.x;
        const bool have = t < total;
        Pending p = a.lists[so + (have ? t : 0)];
        if (!have) p = Pending{0, 0, 0, 0};
        const uint32_t o1 = own_chk[p.im >> 5], o2 = own_chk[p.j];
        TileFetch<DIM> ft;
        if constexpr (TILE && FIRST) ft.issue(c, wkey, hub);
        const bool win = have && o1 == p.idx && o2 == p.idx;
        const uint32_t i = p.im >> 5, idx = p.idx;
        // everything that depends only on (i, j): both rows, the static record of i -- in flight together.  Plain (cached) loads: a
        // pass is a launch of its own, everything earlier passes wrote is visible, and the rows this event owns are touched by
        // nobody else during the pass
        float yi[DIM], yj[DIM], scale_f = 1.f, w = 0.f;
        uint32_t nbr_reg[KREG];
        RecFetch<SREC, KREG> fr;
        RowFetch<DIM> fi, fj;
        fr.issue(a.srec, i, p.im & 31u, win, scale_f, w, nbr_reg);
        fi.issue(c.y, i, win, yi);
        fj.issue(c.y, p.j, win, yj);
        if constexpr (TILE && FIRST) ft.land(s_tile, s_tnode);
        fr.land(stage, p.im & 31u, win, scale_f, w, nbr_reg);
        fi.land(stage, yi);
        fj.land(stage, yj);
        if (win) {
            uint32_t neg[5];
            const uint32_t got = draw_negatives<DIM, KREG, TILE>(c, hub, s_tnode, pcg_hash(nkey + idx), i, nbr_reg, neg);
            run_sample<DIM, F64, TILE>(c, s_tile, yi, yj, w, scale_f, a.step, neg, got);
            done++;
        }
        row_store<DIM>(c.y, p.j, win, stage, yj);  // :1239
        row_store<DIM>(c.y, i, win, stage, yi);    // :1301
        // deferred: append to the next list (one atomic per workgroup, on one of kSub counters), mark for the next pass
        const bool defer = have && !win;
        const unsigned long long m = __ballot(defer);
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        if (lane == 0) s_wave_cnt[wv] = (uint32_t)__popcll(m);
        __syncthreads();
        if (threadIdx.x == 0) {
            const uint32_t tot = s_wave_cnt[0] + s_wave_cnt[1] + s_wave_cnt[2] + s_wave_cnt[3];
            s_base = tot ? atomicAdd(&a.counts[a.dst_list * kSub + dsub], tot) : 0u;
        }
        __syncthreads();
        if (defer) {
            uint32_t before = 0;
            for (int q = 0; q < wv; q++) before += s_wave_cnt[q];
            const uint32_t pos = s_base + before + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            if (pos < a.cap) {
                a.lists[dof + pos] = p;
                const bool mark = !a.backoff || (pcg_hash(idx ^ pcg_hash(a.pass_seq ^ kTagSlCoin)) & 1u);
                if (mark) {
                    own_mark[i] = idx;
                    own_mark[p.j] = idx;
                }
            } else {
                atomicOr(reinterpret_cast<unsigned int*>(a.done_counter + 1024), 1u);
            }
        }
        __syncthreads();  // s_wave_cnt / s_base are reused by the next trip
    };
    // (a tile pass is a slice's first: its grid covers the list in one trip; a workgroup without events leaves at once)
    const uint64_t t_first = blockIdx.x * 256ull, t_step = (uint64_t)gridDim.x * 256ull;
    if (t_first < total) trip(t_first, std::true_type{});
    for (uint64_t t0 = t_first + t_step; t0 < total; t0 += t_step) trip(t0, std::false_type{});
    for (int off = 32; off > 0; off >>= 1) done += __shfl_xor(done, off);
    if ((threadIdx.x & 63) == 0 && done) atomicAdd(&a.done_counter[(blockIdx.x * 4u + (threadIdx.x >> 6) + blockIdx.y * 64u) & 1023u], done);
    if (blockIdx.x == 0 && threadIdx.x == 0) a.counts[a.zero_list * kSub + sub] = 0;
}

// ------------------------------------------------------------------------------------------------------------------
// chain rounds: what is still pending after a slice's passes belongs to a few busy rows (hubs: in-degrees in the thousands on kNN
// graphs of high-dimensional data), and a pass runs ONE event per row.  A chain round runs ALL pending events of a target row in one
// lane, the row in registers: (1) link: every pending event pushes itself on its target's list (atomicExch on a head word) and claims
// its source (owner word, last writer wins); (2) run: the event that ended up at the head of a list walks it -- an event runs if it
// owns its source and the source is not itself the target of a list of this round (its row would be in another lane's registers),
// otherwise it goes to the next pending list; (3) unlink: the heads are cleared.  ~2 us per event of a chain (three dependent
// round trips, the next link prefetched) instead of a pass of its own (a 6-8 us launch).
// ------------------------------------------------------------------------------------------------------------------
constexpr uint32_t kNil = 0xFFFFFFFFu;
__global__ void __launch_bounds__(256) sl_chain_link_kernel(SliceArgs a, uint32_t* __restrict__ head, uint32_t* __restrict__ next) {
    const uint32_t sub = blockIdx.y;
    const uint32_t total = min(a.counts[a.src_list * kSub + sub], (uint32_t)a.cap);
    const uint64_t so = ((uint64_t)a.src_list * kSub + sub) * a.cap;
    const uint32_t coin = pcg_hash(a.pass_seq ^ kTagSlCoin);
    for (uint64_t t = blockIdx.x * 256ull + threadIdx.x; t < total; t += (uint64_t)gridDim.x * 256ull) {
        const Pending p = a.lists[so + t];
        const uint32_t pos = (uint32_t)((uint64_t)sub * a.cap + t);
        // Half of the targets (a fresh coin per node and round) have their lists walked in a round; the events of the others go
        // straight to the next list.  Two pending events i -> j and j -> i would otherwise wait for each other's row for ever (each
        // source is the target of a walked list); with the coin one of the two rows is free in half of the rounds.  (Tossing the coin
        // only for targets that are also sources of pending events was tried: more rounds, not fewer.)
        if (pcg_hash(p.j ^ coin) & 1u) {
            next[pos] = atomicExch(&head[p.j], pos);
            a.owner[p.im >> 5] = p.idx;
        } else {
            next[pos] = kNil;
            const uint32_t dsub = (pos + blockIdx.x) % (uint32_t)kSub;
            const uint32_t at = atomicAdd(&a.counts[a.dst_list * kSub + dsub], 1u);
            if (at < a.cap) a.lists[((uint64_t)a.dst_list * kSub + dsub) * a.cap + at] = p;
            else atomicOr(reinterpret_cast<unsigned int*>(a.done_counter + 1024), 1u);
        }
    }
}
__global__ void __launch_bounds__(256) sl_chain_unlink_kernel(SliceArgs a, uint32_t* __restrict__ head) {
    const uint32_t sub = blockIdx.y;
    const uint32_t total = min(a.counts[a.src_list * kSub + sub], (uint32_t)a.cap);
    const uint64_t so = ((uint64_t)a.src_list * kSub + sub) * a.cap;
    for (uint64_t t = blockIdx.x * 256ull + threadIdx.x; t < total; t += (uint64_t)gridDim.x * 256ull) head[a.lists[so + t].j] = kNil;
}
template <int DIM, int SREC>
__global__ void __launch_bounds__(256) sl_chain_run_kernel(SliceArgs a, const uint32_t* __restrict__ head, const uint32_t* __restrict__ next) {
    constexpr int KP = (SREC - 1) / 2, KREG = KP < 32 ? KP : 32;
    const CeDev c = a.c;
    const uint32_t sub = blockIdx.y;
    const uint32_t total = min(a.counts[a.src_list * kSub + sub], (uint32_t)a.cap);
    const uint64_t so = ((uint64_t)a.src_list * kSub + sub) * a.cap;
    const Pending* src = a.lists + (uint64_t)a.src_list * kSub * a.cap;  // position = sub-list * cap + index
    const bool hub = c.hub_odds != nullptr;
    const uint32_t nkey = pcg_hash((uint32_t)c.seed ^ a.key ^ kTagSlNeg);
    unsigned long long done = 0;
    for (uint64_t t = blockIdx.x * 256ull + threadIdx.x; t < total; t += (uint64_t)gridDim.x * 256ull) {
        const Pending p0 = a.lists[so + t];
        const uint32_t pos = (uint32_t)((uint64_t)sub * a.cap + t);
        if (head[p0.j] != pos) continue;  // the head of its target's list walks it
        float yj[DIM];
        load_row<DIM>(c.y, p0.j, yj);
        uint32_t cur = pos, nxt = next[pos];
        Pending e = p0;
        for (;;) {
            Pending en{0u, 0u, 0u, 0u};
            uint32_t nn = kNil;
            if (nxt != kNil) { en = src[nxt]; nn = next[nxt]; }  // the next link travels while this event runs
            const uint32_t i = e.im >> 5;
            if (a.owner[i] == e.idx && head[i] == kNil) {
                float yi[DIM];
                load_row<DIM>(c.y, i, yi);
                const float* r = a.srec + (uint64_t)i * SREC;
                const float scale_f = r[0], w = r[1 + KP + (e.im & 31u)];
                uint32_t nbr_reg[KREG];
#pragma unroll
                for (int q = 0; q < KREG; q++) nbr_reg[q] = __float_as_uint(r[1 + q]);
                uint32_t neg[5];
                const uint32_t got = draw_negatives<DIM, KREG, false>(c, hub, nullptr, pcg_hash(nkey + e.idx), i, nbr_reg, neg);
                run_sample<DIM, false, false>(c, nullptr, yi, yj, w, scale_f, a.step, neg, got);
                store_row<DIM>(c.y, i, yi);  // :1301
                done++;
            } else {  // the source is claimed by another event or is a target of this round: next round
                const uint32_t dsub = (cur + blockIdx.x) % (uint32_t)kSub;
                const uint32_t at = atomicAdd(&a.counts[a.dst_list * kSub + dsub], 1u);
                if (at < a.cap) a.lists[((uint64_t)a.dst_list * kSub + dsub) * a.cap + at] = e;
                else atomicOr(reinterpret_cast<unsigned int*>(a.done_counter + 1024), 1u);
            }
            if (nxt == kNil) break;
            cur = nxt; e = en; nxt = nn;
        }
        store_row<DIM>(c.y, p0.j, yj);  // :1239
    }
    for (int off = 32; off > 0; off >>= 1) done += __shfl_xor(done, off);
    if ((threadIdx.x & 63) == 0 && done) atomicAdd(&a.done_counter[(blockIdx.x * 4u + (threadIdx.x >> 6) + blockIdx.y * 64u) & 1023u], done);
    if (blockIdx.x == 0 && threadIdx.x == 0) a.counts[a.zero_list * kSub + sub] = 0;
}
template <int DIM>
void launch_chain_run(const SliceArgs& a, unsigned grid, uint32_t srec, const uint32_t* head, const uint32_t* next) {
    if (srec == 16) hipLaunchKernelGGL((sl_chain_run_kernel<DIM, 16>), dim3(grid, kSub), dim3(256), 0, stream(), a, head, next);
    else if (srec == 32) hipLaunchKernelGGL((sl_chain_run_kernel<DIM, 32>), dim3(grid, kSub), dim3(256), 0, stream(), a, head, next);
    else if (srec == 64) hipLaunchKernelGGL((sl_chain_run_kernel<DIM, 64>), dim3(grid, kSub), dim3(256), 0, stream(), a, head, next);
    else hipLaunchKernelGGL((sl_chain_run_kernel<DIM, 128>), dim3(grid, kSub), dim3(256), 0, stream(), a, head, next);
}

template <int DIM, bool F64, bool TILE>
void launch_exec3(const SliceArgs& a, unsigned grid, uint32_t srec) {
    if (srec == 16) hipLaunchKernelGGL((sl_exec_kernel<DIM, 16, F64, TILE>), dim3(grid, kSub), dim3(256), 0, stream(), a);
    else if (srec == 32) hipLaunchKernelGGL((sl_exec_kernel<DIM, 32, F64, TILE>), dim3(grid, kSub), dim3(256), 0, stream(), a);
    else if (srec == 64) hipLaunchKernelGGL((sl_exec_kernel<DIM, 64, F64, TILE>), dim3(grid, kSub), dim3(256), 0, stream(), a);
    else hipLaunchKernelGGL((sl_exec_kernel<DIM, 128, F64, TILE>), dim3(grid, kSub), dim3(256), 0, stream(), a);
}
template <int DIM>
void launch_exec(const SliceArgs& a, unsigned grid, uint32_t srec, bool f64) {
    const bool tile = a.tile && a.c.n > (uint64_t)TileShape<DIM>::kRows * 4ull;
    if (f64) { if (tile) launch_exec3<DIM, true, true>(a, grid, srec); else launch_exec3<DIM, true, false>(a, grid, srec); }
    else { if (tile) launch_exec3<DIM, false, true>(a, grid, srec); else launch_exec3<DIM, false, false>(a, grid, srec); }
}
template <int DIM, bool F64, bool TILE>
void launch_direct3(const DirectArgs& a, uint32_t srec) {
    const unsigned grid = (a.end - a.begin + 256u * a.ept - 1u) / (256u * a.ept);
    if (srec == 16) hipLaunchKernelGGL((sl_direct_kernel<DIM, 16, F64, TILE>), dim3(grid), dim3(256), 0, stream(), a);
    else if (srec == 32) hipLaunchKernelGGL((sl_direct_kernel<DIM, 32, F64, TILE>), dim3(grid), dim3(256), 0, stream(), a);
    else if (srec == 64) hipLaunchKernelGGL((sl_direct_kernel<DIM, 64, F64, TILE>), dim3(grid), dim3(256), 0, stream(), a);
    else hipLaunchKernelGGL((sl_direct_kernel<DIM, 128, F64, TILE>), dim3(grid), dim3(256), 0, stream(), a);
}
template <int DIM>
void launch_direct(const DirectArgs& a, uint32_t srec, bool f64) {
    const bool tile = a.tile && a.c.n > (uint64_t)TileShape<DIM>::kRows * 4ull;
    if (f64) { if (tile) launch_direct3<DIM, true, true>(a, srec); else launch_direct3<DIM, true, false>(a, srec); }
    else { if (tile) launch_direct3<DIM, false, true>(a, srec); else launch_direct3<DIM, false, false>(a, srec); }
}

// stable (LSD radix) sort of (step key, event) pairs on the key bits [0, end_bit)
// Sorts the events by their (slice, class) key between the two buffer pairs; returns true if the result sits in the second pair.
// rocPRIM's double-buffer form: the caller's two pairs ARE the ping-pong buffers, the temporary storage is histograms only.  (The
// in/out form asks for a full copy of keys and values as temporary storage -- 7.6 GB at the C4 shape -- and its size follows the
// batch's event count: whenever a batch set a new record the stream-ordered pool had to get a fresh block from the driver, 1.5-2 s,
// a few times per run.  Found in round 3 as C4-shape batches of 300-900 ms among batches of 121 ms.)
bool sort_events(ae_entropy_optim* o, uint32_t* keys_a, uint32_t* keys_b, Event* vals_a, Event* vals_b, uint64_t count, unsigned end_bit) {
    static_assert(sizeof(Event) == 8, "events are sorted as 64-bit values");
    rocprim::double_buffer<uint32_t> dk(keys_a, keys_b);
    rocprim::double_buffer<unsigned long long> dv(reinterpret_cast<unsigned long long*>(vals_a), reinterpret_cast<unsigned long long*>(vals_b));
    size_t tmp_bytes = 0;
    if (rocprim::radix_sort_pairs(nullptr, tmp_bytes, dk, dv, count, 0, end_bit, stream()) != hipSuccess)
        fail(AE_ERR_NO_DEVICE, "rocprim radix_sort_pairs (size query) failed");
    if (o->sl_sort_tmp.n < tmp_bytes + 1) o->sl_sort_tmp.alloc(2 * tmp_bytes + 4096);  // (kept with the handle: no allocation in the batch)
    if (rocprim::radix_sort_pairs(o->sl_sort_tmp.p, tmp_bytes, dk, dv, count, 0, end_bit, stream()) != hipSuccess)
        fail(AE_ERR_NO_DEVICE, "rocprim radix_sort_pairs failed");
    return dk.current() == keys_b;
}

}  // namespace

namespace ae {

const char* ce_slice_unsupported(const ae_entropy_optim* o) {
    if (o->dev.node_lo != 0 || o->dev.node_hi != o->dev.n) return "a sharded node range (both rows of a sample must be on the device)";
    if (o->dev.nnz >= 0xFFFFFFFFull) return "more than 2^32 edges";
    if (o->dev.n > (1ull << 27)) return "more than 2^27 nodes";
    if (o->g->max_nbng > 32) return "rows of more than 32 neighbours";
    return nullptr;
}

// the edge colouring of the graph: sl_color[e] = class of edge e (a matching), kOverflowColor for the edges of the overflow class
static void slice_color_edges(ae_entropy_optim* o) {
    const uint64_t n = o->dev.n, nnz = o->dev.nnz;
    const EdgeRec* erec = reinterpret_cast<const EdgeRec*>(o->sl_erec.p);
    o->sl_color.alloc(nnz);
    AE_HIP(hipMemsetAsync(o->sl_color.p, kNoColor, nnz, stream()));
    o->sl_node_ov.alloc(n);
    o->sl_node_ov.zero();
    o->sl_classes = 0;
    o->sl_ov_frac = 1.0;
    if (debug_knob("AE_SL_NO_MATCH")) {  // A/B: everything through the optimistic passes (the previous form of the mode)
        hipLaunchKernelGGL(sl_color_giveup_kernel, dim3(blocks_for(nnz, 256)), dim3(256), 0, stream(), nnz, o->sl_color.p);
        hipLaunchKernelGGL(sl_color_cut_kernel, dim3(blocks_for(nnz, 256)), dim3(256), 0, stream(), nnz, erec, o->sl_color.p, 0u, o->sl_node_ov.p);
        check_launch("sl_color");
        return;
    }
    uint32_t cap = kMaxClasses;
    if (debug_knob("AE_SL_CLASS_CAP")) cap = std::min<uint32_t>(kMaxClasses, std::max(1, atoi(debug_knob("AE_SL_CLASS_CAP"))));
    const unsigned long long capmask = cap >= 64 ? ~0ull : ((1ull << cap) - 1ull);
    DevBuf<unsigned long long> used, bid, remaining;
    DevBuf<uint8_t> prop;
    used.alloc_pooled(n); bid.alloc_pooled(n); remaining.alloc_pooled(1024); prop.alloc_pooled(nnz);
    used.zero();
    AE_HIP(hipMemsetAsync(bid.p, 0xFF, sizeof(unsigned long long) * n, stream()));
    const unsigned grid = blocks_for(nnz, 256);
    uint32_t round = 0;
    for (; round < 250; round++) {
        remaining.zero();
        hipLaunchKernelGGL(sl_color_propose_kernel, dim3(grid), dim3(256), 0, stream(), nnz, erec, (const uint8_t*)o->sl_color.p,
                           (const unsigned long long*)used.p, bid.p, prop.p, round, pcg_hash(round ^ kTagSlColor ^ (uint32_t)o->dev.seed), capmask);
        hipLaunchKernelGGL(sl_color_commit_kernel, dim3(grid), dim3(256), 0, stream(), nnz, erec, o->sl_color.p, used.p,
                           (const unsigned long long*)bid.p, (const uint8_t*)prop.p, remaining.p);
        if ((round & 3u) == 3u || round < 2) {
            unsigned long long left = 0;
            for (unsigned long long v : remaining.to_host()) left += v;
            if (!left) { round++; break; }
        }
    }
    hipLaunchKernelGGL(sl_color_giveup_kernel, dim3(grid), dim3(256), 0, stream(), nnz, o->sl_color.p);
    check_launch("sl_color");
    // class masses; the thin tail of classes (only hubs reach them) joins the overflow class: a step is a launch
    DevBuf<double> mass;
    mass.alloc_pooled(kMaxClasses + 1);
    mass.zero();
    hipLaunchKernelGGL(sl_class_mass_kernel, dim3(grid_cap(nnz, 256, 2048)), dim3(256), 0, stream(), nnz, erec, (const uint8_t*)o->sl_color.p, mass.p);
    std::vector<double> hm = mass.to_host();
    double total = 0.;
    for (double v : hm) total += v;
    // How many classes run as matchings.  A step is a launch: it costs ~9 us of latency whatever it holds, and an event in it
    // 0.14 ns (rows of <= 8 columns: ~5.6 random requests at the ~55 G requests/s the memory system serves) to 0.24 ns (wider
    // rows: one wave per SIMD); an event of the overflow class costs 0.26 ns on a lattice, 0.41 ns on an exact kNN graph with hubs
    // (two owner marks, two checks, a pending-list trip and 1.4 - 2 attempts: priced at 0.30 / 0.28 ns -- on the kNN graph of 11 M
    // Higgs-shaped points 0.24 cut at 12 classes, 180 ms per batch; 0.30 at 16, 169 ms; 0.36 at 19, 171 ms) and the class two to four
    // launches per slice.  The cut that minimises the batch time -- 0 =
    // everything optimistic (graphs of a few million edges: their steps would hold a few thousand events), all classes = no
    // overflow (large regular graphs).  Constants measured on MI355X at the C3 / C4 / C5-shard shapes.
    uint32_t top = kMaxClasses;
    while (top > 0 && hm[top - 1] == 0.) top--;
    const double events = (double)o->params.nb_sampling_by_edge * (double)nnz;
    const double slices = std::max(1.0, 4.0 * events / (double)n);
    uint32_t cut = top;
    double tail = hm[kMaxClasses], best = 1e300;
    {
        double t = hm[kMaxClasses];
        for (uint32_t c = top + 1; c-- > 0;) {  // c = number of classes kept
            if (c < top) t += hm[c];
            const double frac = total > 0. ? t / total : 0.;
            const double launches = (double)c + (t > 0. ? (frac < 0.05 ? 2.0 : 4.0) : 0.0);
            const double c_match = o->dev.dim <= 8 ? 0.14e-9 : 0.24e-9;
            const double c_opt = debug_knob("AE_SL_COPT") ? atof(debug_knob("AE_SL_COPT")) * 1e-9 : (o->dev.dim <= 8 ? 0.30e-9 : 0.28e-9);
            const double cost = slices * launches * 9e-6 + events * ((1.0 - frac) * c_match + frac * c_opt);
            if (cost < best) { best = cost; cut = c; tail = t; }
        }
    }
    if (debug_knob("AE_SL_TAIL")) {  // A/B: the tail of classes holding at most this share of the probability mass joins the overflow class
        const double tail_max = atof(debug_knob("AE_SL_TAIL")) * total;
        cut = top;
        tail = hm[kMaxClasses];
        while (cut > 1 && tail + hm[cut - 1] <= tail_max) { tail += hm[cut - 1]; cut--; }
    }
    hipLaunchKernelGGL(sl_color_cut_kernel, dim3(grid), dim3(256), 0, stream(), nnz, erec, o->sl_color.p, cut, o->sl_node_ov.p);
    check_launch("sl_color_cut");
    sync();
    o->sl_classes = cut;
    o->sl_ov_frac = total > 0. ? tail / total : 0.;
    o->sl_color_rounds = round;
    if (debug_knob("AE_CE_PROF"))
        fprintf(stderr, "CESLICE colouring: %u rounds, %u classes, overflow mass %.4f\n", round, cut, o->sl_ov_frac);
}

void ce_slice_prepare(ae_entropy_optim* o) {
    const ae_kgraph* g = o->g;
    if (ce_slice_unsupported(o)) return;  // reported by the first batch
    o->sl_erec.alloc(g->nnz * 4);  // EdgeRec as four words
    hipLaunchKernelGGL(sl_edge_rec_kernel, dim3(blocks_for(g->n, 256)), dim3(256), 0, stream(), o->dev, reinterpret_cast<EdgeRec*>(o->sl_erec.p));
    // static records: 16 floats serve rows of <= 7 neighbours, 32: <= 15, 64: <= 31, 128: 32
    o->sl_srec_floats = g->max_nbng <= 7 ? 16u : (g->max_nbng <= 15 ? 32u : (g->max_nbng <= 31 ? 64u : 128u));
    o->sl_srec.alloc(g->n * o->sl_srec_floats);
    hipLaunchKernelGGL(sl_static_rec_kernel, dim3(blocks_for(g->n, 256)), dim3(256), 0, stream(), o->dev, o->sl_srec_floats, o->sl_srec.p);
    check_launch("sl_prepare");
    // largest edge probability (segments keep the per-edge Poisson mean below 64)
    std::vector<float> hp = o->np->proba.to_host();
    float pmax = 0.f;
    for (float v : hp) pmax = std::max(pmax, v);
    o->sl_pmax = pmax;
    slice_color_edges(o);
    o->sl_owner.alloc(2 * g->n);
    AE_HIP(hipMemsetAsync(o->sl_owner.p, 0xFF, sizeof(uint32_t) * 2 * g->n, stream()));
    o->sl_counts.alloc(3 * kSub);
    o->sl_counts.zero();
    o->sl_done.alloc(1025);
    o->sl_done.zero();
    if (!o->sample_counter.n) { o->sample_counter.alloc(1024); o->sample_counter.zero(); }
}

void ce_slice_gradient_iteration(ae_entropy_optim* o, uint64_t nb_sample, double grad_step, uint32_t iter) {
    if (const char* why = ce_slice_unsupported(o)) fail(AE_ERR_INVALID_ARG, "AE_CE_SLICED: %s", why);
    const uint64_t n = o->dev.n, nnz = o->dev.nnz;
    const double per_node = (double)nb_sample / (double)n;
    // segments of the batch: per-edge Poisson mean <= 64 (f32 inversion: exp(-64) is a normal number, the count is capped at 255),
    // at most 2^30 events per segment
    uint32_t segments = (uint32_t)std::max(1.0, std::ceil(per_node * (double)o->sl_pmax / 64.0));
    segments = std::max(segments, (uint32_t)(nb_sample / (1ull << 30)) + 1u);
    if (iter >= (1u << 20) || segments >= 4096) fail(AE_ERR_INVALID_ARG, "AE_CE_SLICED: batch / segment index too large for the RNG key");
    const double seg_samples = (double)nb_sample / segments;
    // slices of a segment: about half an event per node and slice (a sample is an event at two nodes)
    const double lambda_s = debug_knob("AE_SL_LAMBDA") ? atof(debug_knob("AE_SL_LAMBDA")) : 0.5;
    const uint32_t n_slices = (uint32_t)std::max(1.0, std::ceil(2.0 * seg_samples / (double)n / lambda_s));
    // passes per slice of the overflow class: a thin one (a few per cent of the events: conflicts among them are rare) runs once and
    // carries its losers into the next slice
    int passes = debug_knob("AE_SL_PASSES") ? atoi(debug_knob("AE_SL_PASSES")) : (o->sl_ov_frac < 0.05 ? 1 : 3);
    const int spread = debug_knob("AE_SL_NO_SPREAD") ? 0 : 1;
    // scalar arithmetic: the reference's f64 (embedder.rs:1207-1229) unless the caller opted into f32 (ae_embedder_params.ce_precision)
    const bool f64 = o->params.ce_precision != AE_PRECISION_F32;
    const uint32_t classes = o->sl_classes;
    const bool has_overflow = o->sl_ov_frac > 0.;
    const uint64_t n_keys = (uint64_t)n_slices * (classes + 1u);
    if (n_keys >= (1ull << 31)) fail(AE_ERR_INVALID_ARG, "AE_CE_SLICED: too many steps in a batch");
    o->rounds = segments * n_slices;
    const uint64_t ev_cap = (uint64_t)(seg_samples + 8.0 * std::sqrt(seg_samples) + 1024.0);
    // pending lists of the overflow class: a slice's overflow events (+ 16 sigma) four times over, plus what the rows that receive
    // more overflow events than a slice's passes can run (hubs) accumulate until the drain
    const double per_slice_ov = seg_samples / n_slices * o->sl_ov_frac;
    double backlog = 0.;
    if (has_overflow && passes == 1 && !debug_knob("AE_SL_PASSES")) {
        // One pass per slice serves a thin overflow class only while no ROW is busy in it: a row that receives more than a quarter of
        // an overflow event per slice (a hub whose edges lie beyond the colour budget) queues its events behind one another, slices
        // late -- seen as a final CE 1.5-4.7 % off on the exact kNN graph of 11 M points when the cut was forced to a 2-5 % overflow
        // class.  Such a graph gets the three passes of a thick class.
        DevBuf<double> d_busy;
        d_busy.alloc_pooled(1);
        d_busy.zero();
        double busy = 0.;
        hipLaunchKernelGGL(sl_backlog_kernel, dim3(grid_cap(n, 256, 1024)), dim3(256), 0, stream(), n, (const float*)o->sl_node_ov.p,
                           (float)(seg_samples / (double)n), 0.25f * (float)n_slices, d_busy.p);
        d_busy.download(&busy, 1);
        if (busy > 0.) passes = 3;
    }
    if (has_overflow) {
        DevBuf<double> d_backlog;
        d_backlog.alloc_pooled(1);
        d_backlog.zero();
        hipLaunchKernelGGL(sl_backlog_kernel, dim3(grid_cap(n, 256, 1024)), dim3(256), 0, stream(), n, (const float*)o->sl_node_ov.p,
                           (float)(seg_samples / (double)n), (float)(passes * n_slices), d_backlog.p);
        d_backlog.download(&backlog, 1);
    }
    const uint64_t cap = (uint64_t)((4.0 * per_slice_ov + 16.0 * std::sqrt(per_slice_ov) + 2.0 * backlog) / kSub + 8192.0);  // per sub-list
    if (o->sl_cnt.n < nnz) { o->sl_cnt.alloc(nnz); o->sl_offs.alloc(nnz); }
    if (o->sl_keys0.n < ev_cap) { o->sl_keys0.alloc(ev_cap); o->sl_keys1.alloc(ev_cap); o->sl_vals0.alloc(2 * ev_cap); o->sl_vals1.alloc(2 * ev_cap); }
    if (o->sl_sptr.n < n_keys + 2) o->sl_sptr.alloc(n_keys + 2);
    if (has_overflow && o->sl_lists.n < 3 * (uint64_t)kSub * cap * 4) o->sl_lists.alloc(3 * (uint64_t)kSub * cap * 4);
    if ((uint64_t)kSub * cap >= 0xFFFFFFFFull) fail(AE_ERR_INVALID_ARG, "AE_CE_SLICED: pending lists beyond 2^32 entries");
    if (has_overflow && o->sl_chain_next.n < (uint64_t)kSub * cap) o->sl_chain_next.alloc((uint64_t)kSub * cap);
    if (has_overflow && o->sl_chain_head.n < n) {
        o->sl_chain_head.alloc(n);
        AE_HIP(hipMemsetAsync(o->sl_chain_head.p, 0xFF, sizeof(uint32_t) * n, stream()));
    }
    if (o->sl_class_pos.n < (uint64_t)n_slices * std::max(1u, classes)) o->sl_class_pos.alloc((uint64_t)n_slices * std::max(1u, classes));
    unsigned kbits = 1;
    while (kbits < 32 && (n_keys >> kbits)) kbits++;
    SliceArgs a;
    a.c = o->dev;
    a.srec = o->sl_srec.p;
    a.owner = o->sl_owner.p;
    a.lists = reinterpret_cast<Pending*>(o->sl_lists.p);
    const bool use_tile = !debug_knob("AE_SL_NO_TILE");
    // the LDS tile pays when a negative's row would come from beyond the L2s and the step has enough events to fill the chip anyway
    const uint64_t tile_min_events = debug_knob("AE_SL_TILE_MIN") ? (uint64_t)atoll(debug_knob("AE_SL_TILE_MIN")) : 65536ull;
    const bool y_in_cache = (uint64_t)n * o->dev.dim * 4ull <= (4ull << 20) && !debug_knob("AE_SL_TILE_ALWAYS");
    a.counts = o->sl_counts.p;
    a.cap = cap;
    a.step = grad_step;
    a.done_counter = o->sl_done.p;
    DirectArgs da;
    da.c = o->dev;
    da.srec = o->sl_srec.p;
    da.step = grad_step;
    da.dbg = debug_knob("AE_SL_DBG") ? atoi(debug_knob("AE_SL_DBG")) : 0;
    da.done_counter = o->sl_done.p;
    const unsigned grid_full = (unsigned)std::min<uint64_t>(blocks_for((uint64_t)(per_slice_ov * 1.5 / kSub) + 512, 256), 65535u);  // per sub-list
    const uint32_t ept_force = debug_knob("AE_SL_EPT") ? (uint32_t)std::max(1, atoi(debug_knob("AE_SL_EPT"))) : 0u;
    uint32_t pass_seq = 0, step_seq = 0;
    int cur = 0;  // list that holds what is pending
    const bool prof = debug_knob("AE_CE_PROF") != nullptr;
    auto wall = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_evgen = 0., t_enqueue = 0., t_drain = 0.;
    int drain_iterations = 0;
    const double t_begin = wall();
    o->sl_counts.zero();
    std::vector<uint8_t> class_pos((size_t)n_slices * std::max(1u, classes));
    std::vector<uint32_t> hptr(n_keys + 2);
    Event* ev0 = reinterpret_cast<Event*>(o->sl_vals0.p);
    Event* ev1 = reinterpret_cast<Event*>(o->sl_vals1.p);
    // one chain round over the pending list `cur` (see sl_chain_run_kernel); the f64-scalar debug variant keeps the passes
    const bool use_chains = !f64 && !debug_knob("AE_SL_NO_CHAIN");
    auto chain_round = [&](unsigned grid) {
        a.src_list = cur; a.dst_list = (cur + 1) % 3; a.zero_list = (cur + 2) % 3;
        a.pass_seq = pass_seq++;
        hipLaunchKernelGGL(sl_chain_link_kernel, dim3(grid, kSub), dim3(256), 0, stream(), a, o->sl_chain_head.p, o->sl_chain_next.p);
        AE_DISPATCH_DIM(o->dev.dim, launch_chain_run, a, grid, o->sl_srec_floats, (const uint32_t*)o->sl_chain_head.p, (const uint32_t*)o->sl_chain_next.p);
        hipLaunchKernelGGL(sl_chain_unlink_kernel, dim3(grid, kSub), dim3(256), 0, stream(), a, o->sl_chain_head.p);
        cur = (cur + 1) % 3;
    };
    for (uint32_t sg = 0; sg < segments; sg++) {
        const uint32_t key = (iter << 12) | sg;
        const double t_seg = wall();
        // the order of the classes inside every slice: a fresh uniform permutation (so the order of two events that share a node is
        // uniform, as in an i.i.d. sequence)
        {
            std::mt19937_64 rng(o->dev.seed * 0x9E3779B97F4A7C15ull + ((uint64_t)key << 20) + 0x5851F42D4C957F2Dull);
            for (uint32_t s = 0; s < n_slices && classes; s++) {
                uint8_t* row = class_pos.data() + (size_t)s * classes;
                for (uint32_t q = 0; q < classes; q++) row[q] = (uint8_t)q;
                for (uint32_t q = classes - 1; q > 0; q--) std::swap(row[q], row[rng() % (q + 1)]);
            }
            if (classes) o->sl_class_pos.upload(class_pos.data(), class_pos.size());
        }
        hipLaunchKernelGGL(sl_count_kernel, dim3(blocks_for(nnz, 256)), dim3(256), 0, stream(), o->dev, (float)(seg_samples / (double)n), key, o->sl_cnt.p);
        {
            size_t tmp_bytes = 0;
            if (rocprim::exclusive_scan(nullptr, tmp_bytes, o->sl_cnt.p, o->sl_offs.p, 0u, nnz, rocprim::plus<uint32_t>(), stream()) != hipSuccess)
                fail(AE_ERR_NO_DEVICE, "rocprim exclusive_scan (size query) failed");
            DevBuf<char> tmp;
            tmp.alloc_pooled(tmp_bytes ? tmp_bytes : 1);
            if (rocprim::exclusive_scan(tmp.p, tmp_bytes, o->sl_cnt.p, o->sl_offs.p, 0u, nnz, rocprim::plus<uint32_t>(), stream()) != hipSuccess)
                fail(AE_ERR_NO_DEVICE, "rocprim exclusive_scan failed");
        }
        uint32_t last[2];
        AE_HIP(hipMemcpyAsync(&last[0], o->sl_offs.p + (nnz - 1), 4, hipMemcpyDeviceToHost, stream()));
        AE_HIP(hipMemcpyAsync(&last[1], o->sl_cnt.p + (nnz - 1), 4, hipMemcpyDeviceToHost, stream()));
        sync();
        const double t_cnt = wall();
        const uint32_t total = last[0] + last[1];
        if (total > ev_cap) fail(AE_ERR_STATE, "AE_CE_SLICED: more events than the 8-sigma capacity");
        hipLaunchKernelGGL(sl_fill_kernel, dim3(blocks_for(nnz, 256)), dim3(256), 0, stream(), o->dev, key, (const uint32_t*)o->sl_cnt.p,
                           (const uint32_t*)o->sl_offs.p, n_slices, reinterpret_cast<const EdgeRec*>(o->sl_erec.p), (const uint8_t*)o->sl_color.p,
                           (const uint8_t*)o->sl_class_pos.p, classes, spread, o->sl_keys0.p, ev0);
        if (prof) sync();
        const double t_fill = wall();
        const bool in_second = sort_events(o, o->sl_keys0.p, o->sl_keys1.p, ev0, ev1, total, kbits);
        const uint32_t* sorted_keys = in_second ? o->sl_keys1.p : o->sl_keys0.p;
        Event* sorted_ev = in_second ? ev1 : ev0;
        if (prof) sync();
        const double t_sort = wall();
        hipLaunchKernelGGL(sl_sptr_kernel, dim3(blocks_for(n_keys + 1, 256)), dim3(256), 0, stream(), sorted_keys, total, (uint32_t)n_keys,
                           o->sl_sptr.p);
        check_launch("sl_events");
        o->sl_sptr.download(hptr.data(), n_keys + 1);
        const double t_ev = wall();
        t_evgen += t_ev - t_seg;
        if (prof && t_ev - t_seg > 0.1)
            fprintf(stderr, "CESLICE slow event generation: permutation + count + scan %.1f ms, fill %.1f ms, sort %.1f ms, slice pointers + download %.1f ms\n",
                    (t_cnt - t_seg) * 1e3, (t_fill - t_cnt) * 1e3, (t_sort - t_fill) * 1e3, (t_ev - t_sort) * 1e3);
        a.ev = sorted_ev;
        da.ev = sorted_ev;
        a.key = key;
        da.key = key;
        // NOTE: event indices are positions in this segment's sorted array; what is still pending when a segment ends is
        // finished (the drain below) before the next segment reuses the arrays
        for (uint32_t s = 0; s < n_slices; s++) {
            const uint32_t* sp = hptr.data() + (size_t)s * (classes + 1u);
            for (uint32_t q = 0; q < classes; q++) {  // the slice's matchings, in this slice's order
                if (sp[q + 1] == sp[q]) continue;
                da.begin = sp[q];
                da.end = sp[q + 1];
                const uint32_t cnt = da.end - da.begin;
                // events per thread: amortises the tile; only where the grid still covers the chip several times
                da.ept = ept_force ? ept_force : std::min(4u, std::max(1u, cnt / (256u * 384u)));
                da.tile = (use_tile && !y_in_cache && cnt >= tile_min_events) ? 1 : 0;
                da.step_seq = step_seq++;
                AE_DISPATCH_DIM(o->dev.dim, launch_direct, da, o->sl_srec_floats, f64);
            }
            if (!has_overflow) continue;
            a.f0 = sp[classes];
            a.f1 = sp[classes + 1u];
            a.src_list = cur; a.dst_list = (cur + 1) % 3; a.zero_list = (cur + 2) % 3;
            a.owner_mark = 0;
            hipLaunchKernelGGL(sl_mark_kernel, dim3(grid_full, kSub), dim3(256), 0, stream(), a);
            cur = (cur + 1) % 3;
            for (int p = 0; p < passes; p++) {
                a.src_list = cur; a.dst_list = (cur + 1) % 3; a.zero_list = (cur + 2) % 3;
                a.owner_chk = p & 1; a.owner_mark = (p + 1) & 1;
                a.backoff = p >= 1;
                a.tile = (p == 0 && use_tile && !y_in_cache && per_slice_ov >= (double)tile_min_events) ? 1 : 0;
                a.pass_seq = pass_seq++;
                const unsigned grid = p == 0 ? grid_full : std::max(4u, grid_full >> (2 * p));
                AE_DISPATCH_DIM(o->dev.dim, launch_exec, a, grid, o->sl_srec_floats, f64);
                cur = (cur + 1) % 3;
            }
            // Rows that receive more overflow events per slice than the passes above can run (hubs: one event per row and pass, as
            // the row's lock serialises them in the reference) would carry a growing backlog to the end of the batch -- their
            // events would all run AFTER everybody else's (measured on a graph whose hub is every node's neighbour: final CE
            // 0.61x).  Where the graph has such rows (expected backlog > 1 % of the events, known from the edge colouring) the slice is not left before its
            // pending list is back to the size conflicts alone explain: a look at the counters every 4 extra passes.
            if (backlog > 0.01 * seg_samples) {  // (below 1 % of the events the late ones do not show: blobs k = 6 with in-degrees up to 105: CE 1.002-1.014 either way)
                const uint64_t carry_ok = (uint64_t)(0.05 * per_slice_ov) + 16;
                uint64_t prev_left = 0;
                int prev_pass = 0;
                bool chain_mode = false;
                for (int extra = 0, look = 0; extra < 1000000; look++) {
                    uint32_t lefts[kSub];
                    AE_HIP(hipMemcpyAsync(lefts, o->sl_counts.p + cur * kSub, 4 * kSub, hipMemcpyDeviceToHost, stream()));
                    sync();
                    uint64_t left = 0;
                    for (int q = 0; q < kSub; q++) left += lefts[q];
                    if (left <= carry_ok) break;
                    const unsigned grid = (unsigned)std::min<uint64_t>(std::max<uint64_t>(4, blocks_for(left / kSub + 256, 256)), 65535u);
                    // Passes while they make progress; a chain round (all pending events of a row in one lane) once the last passes
                    // ran about one event per pass and row -- more than 48 further passes would be needed at that pace.
                    const bool slow = prev_pass > 0 && (double)left * prev_pass > 48.0 * (double)std::max<uint64_t>(1, prev_left > left ? prev_left - left : 0);
                    prev_left = left;
                    // (Chain rounds INSIDE the slices were measured and are off: a round walks a hub's few hundred events of the slice at
                    // ~3 us each and the coin halves that -- no better than as many passes; 6.25 M-node kNN graph of 128-D data: 1.37 s per
                    // batch in the slices against 0.88 s.  They pay in the drain, where thousands of events of a few rows are left.)
                    if (use_chains && debug_knob("AE_SL_CHAIN_IN_SLICE") && (slow || chain_mode)) { chain_round(grid); chain_round(grid); extra += 2; chain_mode = true; continue; }
                    // 4, 4, 8, 16, 32, 64 passes between looks: a look is a host round trip
                    const int n_pass = look < 2 ? 4 : std::min(64, 4 << (look - 1));
                    extra += n_pass;
                    prev_pass = n_pass;
                    for (int p = 0; p < n_pass; p++) {
                        a.src_list = cur; a.dst_list = (cur + 1) % 3; a.zero_list = (cur + 2) % 3;
                        a.owner_chk = (passes + p) & 1; a.owner_mark = (passes + p + 1) & 1;
                        a.backoff = 1;
                        a.tile = 0;
                        a.pass_seq = pass_seq++;
                        AE_DISPATCH_DIM(o->dev.dim, launch_exec, a, grid, o->sl_srec_floats, f64);
                        cur = (cur + 1) % 3;
                    }
                }
            }
            // the next slice's mark kernel marks in owner[0]; the last pass above marked in owner[(passes + extra) & 1]: the mark
            // kernel re-marks everything that is pending anyway
        }
        const double t_enq = wall();
        t_enqueue += t_enq - t_ev;
        // drain: passes until nothing is pending (a look at the counters every 8 passes)
        uint64_t drain_prev_left = 0;
        int drain_prev_pass = 0;
        bool drain_chain_mode = false;
        for (int guard = 0; has_overflow && guard < 1000000; guard++) {
            drain_iterations++;
            uint32_t lefts[kSub];
            unsigned long long flag = 0;
            AE_HIP(hipMemcpyAsync(lefts, o->sl_counts.p + cur * kSub, 4 * kSub, hipMemcpyDeviceToHost, stream()));
            AE_HIP(hipMemcpyAsync(&flag, o->sl_done.p + 1024, sizeof(flag), hipMemcpyDeviceToHost, stream()));
            sync();
            if (flag) break;  // pending list overflow: reported below
            uint64_t left = 0;
            for (int q = 0; q < kSub; q++) left += lefts[q];
            if (!left) break;
            if (guard == 999999) fail(AE_ERR_STATE, "AE_CE_SLICED: pending events did not drain");
            // re-mark all (owner[0]) then 8 passes
            a.f0 = a.f1 = 0;
            a.src_list = cur; a.dst_list = (cur + 1) % 3; a.zero_list = (cur + 2) % 3;
            a.owner_mark = 0;
            const unsigned grid = (unsigned)std::min<uint64_t>(std::max<uint64_t>(4, blocks_for(left / kSub + 256, 256)), 65535u);
            // Passes while they make progress; chain rounds (all pending events of a row in one lane, sl_chain_run_kernel) for the rest of
            // the drain once more than 512 further passes would be needed at the pace of the last ones -- thousands of events of a few rows
            // are left, one per row and pass.  (Measured: a 10 000-in-degree hub in 1 M nodes 56 -> 41 ms per batch; the kNN graph of 128-D
            // data, max in-degree 8 764: drain 320 -> 30 ms.  With the threshold at 48 the kNN graph of 28-D data -- in-degrees up to 133,
            // ~750 events on the busiest row -- lost 63 -> 67-75 ms: a round costs three launches, a look and a coin.)
            const bool slow = drain_prev_pass > 0 &&
                              (double)left * drain_prev_pass > 512.0 * (double)std::max<uint64_t>(1, drain_prev_left > left ? drain_prev_left - left : 0);
            drain_prev_left = left;
            if (use_chains && (slow || drain_chain_mode)) { chain_round(grid); chain_round(grid); drain_chain_mode = true; continue; }
            hipLaunchKernelGGL(sl_mark_kernel, dim3(grid, kSub), dim3(256), 0, stream(), a);
            cur = (cur + 1) % 3;
            // 8 passes before the next look, then 16, 32, 64: what is left after the first looks are the events of a few hubs, one per
            // hub and pass (a kNN graph with hubness weighting: ~750 passes at the end of a batch -- 94 looks of 8 were 4 ms of host
            // round trips; a pass over an empty list is a 6 us launch)
            const int n_pass = std::min(64, 8 << std::min(guard, 3));
            drain_prev_pass = n_pass;
            for (int p = 0; p < n_pass; p++) {
                a.src_list = cur; a.dst_list = (cur + 1) % 3; a.zero_list = (cur + 2) % 3;
                a.owner_chk = p & 1; a.owner_mark = (p + 1) & 1;
                a.backoff = 1;
                a.tile = 0;
                a.pass_seq = pass_seq++;
                AE_DISPATCH_DIM(o->dev.dim, launch_exec, a, grid, o->sl_srec_floats, f64);
                cur = (cur + 1) % 3;
            }
        }
        t_drain += wall() - t_enq;
    }
    if (prof) fprintf(stderr, "CESLICE batch %u: event generation %.1f ms, slices enqueued in %.1f ms, first look + drain %.1f ms (%d looks), total %.1f ms\n", iter,
                      t_evgen * 1e3, t_enqueue * 1e3, t_drain * 1e3, drain_iterations, (wall() - t_begin) * 1e3);
    check_launch("ce_slice");
    std::vector<unsigned long long> h = o->sl_done.to_host();
    o->sl_done.zero();
    if (h[1024]) {
        o->sl_counts.zero();
        sync();
        fail(AE_ERR_STATE, "AE_CE_SLICED: pending list overflow");
    }
    // samples executed, into the common counter
    unsigned long long d = 0;
    for (int q = 0; q < 1024; q++) d += h[q];
    std::vector<unsigned long long> hc = o->sample_counter.to_host();
    hc[0] += d;
    o->sample_counter.upload(hc.data(), hc.size());
    sync();
}

}  // namespace ae
