// ce_slice_kernels.h -- device side of AE_CE_SLICED (ce_slice.hip holds the host driver, the colouring and the event generation):
// lane-group memory access, the tile of negatives, one sample on rows in registers, the step kernel (a colour class of a time slice,
// with chains through hub rows), the optimistic pass kernel of the overflow class, chain rounds.  Everything here is a template over
// the row stride; ce_slice_dim*.hip instantiate one stride each (the translation units compile side by side).
#pragma once
#include "ce_node_common.h"
#include "ce_sample_math.h"

#include <type_traits>

namespace ae {
namespace sl {

constexpr uint32_t kTagSlCount = 0xFFFF0031u, kTagSlTime = 0xFFFF0032u, kTagSlNeg = 0xFFFF0033u, kTagSlCoin = 0xFFFF0034u, kTagSlColor = 0xFFFF0035u, kTagSlPool = 0xFFFF0036u;
// the pending list is kept as kSub sub-lists with a counter each: appends (one atomic per WORKGROUP) spread over kSub addresses --
// one counter serialises at ~12 ns per atomic, which with one atomic per wave was 2/3 of a pass at the C3 shape
constexpr int kSub = 16;
constexpr uint32_t kMaxClasses = 64;        // colours are bits of a 64-bit mask per node
constexpr uint8_t kNoColor = 0xFFu, kOverflowColor = 0xFEu;

struct EdgeRec {       // per edge, 16 bytes (colouring, event generation)
    uint32_t j;        // target
    float w;           // probability
    uint32_t flags;    // kHalfEvent: the source belongs to another shard (multi-GPU)
    uint32_t im;       // source << 5 | slot of the edge in the source's row
};
constexpr uint32_t kNoNode = 0xFFFFFFFFu;                // (<= 2^27 nodes: ce_slice_unsupported)
constexpr uint32_t kNodeMask = 0x07FFFFFFu;
// Multi-GPU (a sharded node range): an edge whose SOURCE belongs to another shard fires here as a HALF event -- the owner of its
// target applies the attraction to the target's row against its replica of the source's row (which it neither stores nor repulses:
// the owner of the source runs the other half, with the negatives, against ITS replica of the target).  Bit 31 of Event::j.
constexpr uint32_t kHalfEvent = 0x80000000u;
__host__ __device__ __forceinline__ uint32_t ev_node(uint32_t j) { return j & kNodeMask; }
__host__ __device__ __forceinline__ bool ev_half(uint32_t j) { return (j & kHalfEvent) != 0u; }
// An event in the sorted arrays: 12 bytes {source << 5 | slot of the edge in the source's row, target, the edge's probability} -- the
// rows of both end points and what is static about the source are requested in ONE hop after the (coalesced) event load.  The events
// of a step are sorted by target: those that share one are adjacent and run as a chain through the target's row (sl_step_body).
// (w rides in the event since round 6: with it a source's embedded scale and neighbour ids fit into the 64-byte LINE of its row --
// NodeLine below -- and the source costs one request where record and row were two.)
struct Event {
    uint32_t im, j;   // j: target, kHalfEvent in bit 31
    float w;          // probability of the edge (embedder.rs:1184)
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
    const uint32_t* hub_pool;   // (hubness weighting) the batch's pool of NodeSampler draws, see TileFetch
    uint32_t hub_pool_n;
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
    int dbg;                    // debug knob AE_SL_DBG (measurement only): 1 no arithmetic, 2 no stores, 4 no negatives, 8 no static record; 64: merged slices read their negatives late (see there)
    double step;
    unsigned long long* done_counter;   // [1024] spread counters of executed samples; [1024] = error flags (1: pending list overflow, 2: hand-over poll budget)
    uint32_t* chunk_flag;       // hand-over of a target's row between the 64-event chunks of a step: [chunk] = step token once the chunk's tail is through
    const uint32_t* hub_pool;   // (hubness weighting) the batch's pool of NodeSampler draws, see TileFetch
    uint32_t hub_pool_n;
};

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
__device__ __forceinline__ void coop_issue(const float* __restrict__ base, uint32_t idx, bool want, f4 (&pc)[NF / 4], uint32_t stride = NF) {   // (stride: floats from record to record)
    constexpr int G = NF / 4;
    const uint32_t sub = (threadIdx.x & 63) & (G - 1);
    // UNCONDITIONAL loads (a lane that wants nothing asks for record 0: one shared line): a load under `if (want)` makes the register
    // allocator merge the two paths with a copy behind the load -- an `s_waitcnt vmcnt(0)` in the middle of the issue phase, which
    // also drains the tile's loads (seen in the ISA of sl_direct_kernel<8,16>)
    const uint32_t word = want ? idx : 0u;
#pragma unroll
    for (int t = 0; t < G; t++) {
        const uint32_t wt = group_bcast<G>(word, t);
        pc[t] = *reinterpret_cast<const f4*>(base + (uint64_t)wt * stride + sub * 4u);
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
__device__ __forceinline__ void coop_store(float* __restrict__ base, uint32_t idx, bool want, const float* stage, uint32_t stride = NF, int nt = 0) {
    constexpr int G = NF / 4, RS = NF + 4;
    const int lane = threadIdx.x & 63, sub = lane & (G - 1), gb = lane & ~(G - 1);
    const uint32_t word = idx | (want ? 0x80000000u : 0u);
    wave_lds_sync();
#pragma unroll
    for (int t = 0; t < G; t++) {
        const uint32_t wt = group_bcast<G>(word, t);
        if (wt & 0x80000000u) {
            f4* dst = reinterpret_cast<f4*>(base + (uint64_t)(wt & 0x7FFFFFFFu) * stride + (uint32_t)sub * 4u);
            const f4 v = *reinterpret_cast<const f4*>(stage + (gb + t) * RS + sub * 4);
            // (A/B, AE_SL_DBG bit 128: a non-temporal store; bit 512: a write-through store (sc0 sc1) -- the row leaves the caches as it is written
            // instead of in the burst at the kernel's end)
            if (nt == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(dst), "v"(v) : "memory");
            else if (nt == 1) __builtin_nontemporal_store(v, dst);
            else *dst = v;
        }
    }
    wave_lds_sync();
}
// A node's LINE in a merged batch on rows of 8 or of 2 columns (ce_slice_gradient_iteration: words_in_rows): four pieces --
//   8 columns: 64 bytes = {word of the even slices, 8 spare bytes | the row, 2 pieces | word of the odd slices, 8 spare bytes}
//   2 columns: 32 bytes = {word of the even slices | the row | word of the odd slices | 8 spare bytes}
// c.y points at the ROW of node 0 (every other kernel of the mode sees rows kFloats apart).  One request of four lanes brings row
// and words; the row's store takes the slice's word along, zeroed (the event that stores a row the plain way is the node's last of
// the slice: it wipes the word) -- the pieces [0, kRowPieces] on an even slice, [1, kRowPieces + 1] on an odd one.
template <int DIM>
struct LineShape {
    static constexpr int kFloats = DIM == 8 ? 16 : 8;      // a node's line (= the stride of the rows)
    static constexpr int kPiece = kFloats / 4;             // floats per piece: one lane's share of a request
    static constexpr int kRowPieces = DIM == 8 ? 2 : 1;
    static constexpr int kRowAt = kPiece;                  // floats from the start of the line
    static constexpr int kOddWordAt = kPiece * (kRowPieces + 1);
    static constexpr int kStageRow = kFloats + 4;
};
template <int DIM>
constexpr bool kHasLines = DIM == 8 || DIM == 2;
template <int DIM>
struct LineFetch {
    using S = LineShape<DIM>;
    using piece_t = float __attribute__((ext_vector_type(S::kPiece)));
    piece_t pc[4];
    __device__ __forceinline__ void issue(const float* __restrict__ y, uint32_t node, bool want) {
        const uint32_t sub = (threadIdx.x & 63) & 3u, word = want ? node : 0u;   // (unconditional loads: coop_issue says why)
        const float* base = y - S::kRowAt;
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const uint32_t wt = group_bcast<4>(word, t);
            pc[t] = *reinterpret_cast<const piece_t*>(base + (uint64_t)wt * S::kFloats + sub * (uint32_t)S::kPiece);
        }
    }
    __device__ __forceinline__ void land(float* stage, float* row, unsigned long long& word, uint32_t set) {
        const int lane = threadIdx.x & 63, sub = lane & 3, gb = lane & ~3;
#pragma unroll
        for (int t = 0; t < 4; t++) *reinterpret_cast<piece_t*>(stage + (gb + t) * S::kStageRow + sub * S::kPiece) = pc[t];
        wave_lds_sync();
        const float* p = stage + lane * S::kStageRow;
#pragma unroll
        for (int q = 0; q < DIM; q++) row[q] = p[S::kRowAt + q];
        const float* wp = p + (set ? S::kOddWordAt : 0);
        word = ((unsigned long long)__float_as_uint(wp[1]) << 32) | __float_as_uint(wp[0]);
        wave_lds_sync();
    }
};
// a store that leaves the caches as it is written (sc0 sc1): the other XCDs' agent-scope loads see it without waiting for the launch's end
__device__ __forceinline__ void store_through(f4* dst, const f4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(dst), "v"(v) : "memory"); }
using f2v = __attribute__((ext_vector_type(2))) float;
__device__ __forceinline__ void store_through(f2v* dst, const f2v v) { asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" : : "v"(dst), "v"(v) : "memory"); }
template <int DIM>
__device__ __forceinline__ void line_store(float* __restrict__ y, uint32_t node, bool want, float* stage, const float* row, uint32_t set, bool through = false) {
    using S = LineShape<DIM>;
    using piece_t = float __attribute__((ext_vector_type(S::kPiece)));
    const int lane = threadIdx.x & 63, sub = lane & 3, gb = lane & ~3;
    float* p = stage + lane * S::kStageRow;
#pragma unroll
    for (int q = 0; q < S::kFloats; q++) p[q] = (q >= S::kRowAt && q < S::kRowAt + DIM) ? row[q - S::kRowAt] : 0.f;
    const uint32_t word = node | (want ? 0x80000000u : 0u);
    const int lo = set ? 1 : 0, hi = (set ? 1 : 0) + S::kRowPieces;
    float* base = y - S::kRowAt;
    wave_lds_sync();
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const uint32_t wt = group_bcast<4>(word, t);
        if ((wt & 0x80000000u) && sub >= lo && sub <= hi) {
            piece_t* dst = reinterpret_cast<piece_t*>(base + (uint64_t)(wt & 0x7FFFFFFFu) * S::kFloats + (uint32_t)sub * (uint32_t)S::kPiece);
            const piece_t v = *reinterpret_cast<const piece_t*>(stage + (gb + t) * S::kStageRow + sub * S::kPiece);
            if (through) store_through(dst, v); else *dst = v;
        }
    }
    wave_lds_sync();
}
// rows: cooperative for 8 and 16 columns (2 / 4 lanes per row), one lane per row otherwise (<= 4 columns: one request anyway;
// 32 / 64 columns: rare, kept simple)
template <int DIM>
constexpr bool kCoopRow = DIM == 8 || DIM == 16;
template <int SREC>
constexpr bool kCoopRec = SREC <= 32;
template <int DIM>
constexpr int kRowStage = kHasLines<DIM> ? LineShape<DIM>::kFloats : (kCoopRow<DIM> ? DIM : 0);   // (rows of 8 / 2 columns may travel as whole node lines: LineFetch)
template <int DIM, int SREC>
constexpr int kStageFloats = (kRowStage<DIM> > 0 || kCoopRec<SREC>) ? 64 * (((kRowStage<DIM> > (kCoopRec<SREC> ? SREC : 0)) ? kRowStage<DIM> : SREC) + 4) : 1;

template <int DIM>
struct RowFetch {  // a coordinate row on its way to its lane
    f4 pc[kCoopRow<DIM> ? DIM / 4 : 1];
    __device__ __forceinline__ void issue(const float* __restrict__ y, uint32_t node, bool want, float* out, uint32_t stride) {
        if constexpr (kCoopRow<DIM>) coop_issue<DIM>(y, node, want, pc, stride);
        else load_row<DIM>(y + (uint64_t)(want ? node : 0u) * stride, 0u, out);
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
// rows handed from one workgroup of a launch to another: past the caches (agent scope)
template <int DIM>
__device__ __forceinline__ void store_row_agent(float* __restrict__ y, uint32_t node, const float* in, uint32_t stride) {
    float* p = y + (uint64_t)node * stride;
    if constexpr (DIM % 2 == 0) {
#pragma unroll
        for (int q = 0; q < DIM / 2; q++) {
            const uint64_t bits = ((uint64_t)__float_as_uint(in[2 * q + 1]) << 32) | __float_as_uint(in[2 * q]);
            __hip_atomic_store(reinterpret_cast<uint64_t*>(p) + q, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    } else {
#pragma unroll
        for (int t = 0; t < DIM; t++) __hip_atomic_store(reinterpret_cast<uint32_t*>(p) + t, __float_as_uint(in[t]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
template <int DIM>
__device__ __forceinline__ void load_row_agent(const float* __restrict__ y, uint32_t node, float* out, uint32_t stride) {
    const float* p = y + (uint64_t)node * stride;
    if constexpr (DIM % 2 == 0) {
#pragma unroll
        for (int q = 0; q < DIM / 2; q++) {
            const uint64_t bits = __hip_atomic_load(reinterpret_cast<const uint64_t*>(p) + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            out[2 * q] = __uint_as_float((uint32_t)bits);
            out[2 * q + 1] = __uint_as_float((uint32_t)(bits >> 32));
        }
    } else {
#pragma unroll
        for (int t = 0; t < DIM; t++) out[t] = __uint_as_float(__hip_atomic_load(reinterpret_cast<const uint32_t*>(p) + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    }
}
template <int DIM>
__device__ __forceinline__ void row_store(float* __restrict__ y, uint32_t node, bool want, float* stage, const float* in, uint32_t stride, int nt = 0) {
    if constexpr (kCoopRow<DIM>) {
        float* p = stage + (threadIdx.x & 63) * (DIM + 4);
#pragma unroll
        for (int q = 0; q < DIM / 4; q++) {
            f4 v; v.x = in[4 * q]; v.y = in[4 * q + 1]; v.z = in[4 * q + 2]; v.w = in[4 * q + 3];
            *reinterpret_cast<f4*>(p + 4 * q) = v;
        }
        coop_store<DIM>(y, node, want, stage, stride, nt);
    } else {
        if (want) {
            if (nt == 2) store_row_agent<DIM>(y, node, in, stride);   // (rows of <= 4 columns: one 8- / 16-byte piece or two)
            else store_row<DIM>(y + (uint64_t)node * stride, 0u, in);
        }
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

// A node's LINE in a batch that runs one launch per class (ce_slice_gradient_iteration: `node_lines`): the batch's internal copy of the
// coordinates keeps, behind every row, what is static about the node as a SOURCE -- LINE floats = {the row, DIM | embedded scale |
// neighbour ids, LINE - DIM - 1 of them, padded with ~0} -- so that a source costs ONE request (a group of LINE / 4 lanes) where the
// static record and the row were two, and half the bytes (configs[3]: 8 columns, 6 neighbours: 15 floats in a 64-byte line against a
// 64-byte record + a 32-byte row out of another line).  The sampled edge's probability comes with the event.  A target's row and the
// rows of the tile of negatives are the first DIM floats of their nodes' lines (CeDev::ystride = LINE), the stores go there too.  The
// step kernel is bound by the NUMBER of requests that miss the L2 (tools/ubench_rowgather.hip): six per event became five.
constexpr int node_line_floats(int dim, int max_nbng) {   // 0: no line form for this shape (the static record stays)
    const int need = dim + 1 + max_nbng;
    return dim > 16 ? 0 : (need <= 8 ? 8 : (need <= 16 ? 16 : (need <= 32 ? 32 : 0)));
}
template <int DIM, int LINE>
struct LineRec {
    static constexpr int KP = LINE - DIM - 1;
    f4 pc[LINE / 4];
    __device__ __forceinline__ void issue(const float* __restrict__ y, uint32_t node, bool want) { coop_issue<LINE>(y, node, want, pc, (uint32_t)LINE); }
    __device__ __forceinline__ void land(float* stage, float* yi, float& scale, uint32_t (&nbr_reg)[KP]) {
        coop_land<LINE>(pc, stage);
        const float* p = stage + (threadIdx.x & 63) * (LINE + 4);
#pragma unroll
        for (int q = 0; q < DIM; q++) yi[q] = p[q];
        scale = p[DIM];
#pragma unroll
        for (int q = 0; q < KP; q++) nbr_reg[q] = __float_as_uint(p[DIM + 1 + q]);
        wave_lds_sync();
    }
};

// ------------------------------------------------------------------------------------------------------------------
// one sample on rows held in registers
// ------------------------------------------------------------------------------------------------------------------
// tile of coordinate rows for the negatives of a crowded launch: kW windows of kL consecutive rows each, window starts uniform
// over the nodes (wrapping) and fresh per workgroup and launch, staged in LDS with coalesced loads.  A negative is then "window
// uniform, row uniform": every node has the same probability 1/n, as in embedder.rs:1121; the rows are as fresh as the launch
// (it started after every earlier launch's writes).  What differs from the reference: the negatives of the samples a workgroup
// runs in a launch come from the same kW windows.  Across samples that is harmless (measured: samples sharing all their candidates
// four by four, or a pool of 256 nodes per workgroup, leave the result where independent draws put it).  INSIDE a sample it is not:
// the reference's five negatives are five different nodes at five unrelated places (a repeat has probability 25 / n), while five
// slots out of 256 repeat a row in 4% of the samples and fall into the same window -- neighbouring ids, on a graph whose ids carry
// locality neighbouring points -- in 62%: two coherent pushes on y_i.  Round 2-4's tile did exactly that and biased clustered graphs
// (1 M Higgs-shaped points, 8 columns: CE 0.90 of the exact mode's, the shortest edges 2.5x longer; with component-ordered ids CE
// 0.61), which lattices and single blobs never showed.  draw_negatives therefore takes no window twice in a sample (with
// hubness weighting: no row twice): the sample's joint law is again "five different, unrelated nodes", every node still has
// probability 1 / n by symmetry.  tools/run_tile_bias.py, DESIGN.md 4.3.
template <int DIM>
struct TileShape {
    // 256 rows (16 windows of 16 consecutive rows) serve the 5 x 256 draws of a workgroup: staging costs one coalesced row read per
    // sample instead of five random ones (a tile of 1024 rows costs as much as the gathers it replaces)
    static constexpr int kRows = DIM <= 16 ? 256 : 128;
    static constexpr int kL = 16;   // (draw_negatives: window = slot >> 4)
    static constexpr int kW = kRows / kL;
    static constexpr int kPieces = kRows * (DIM % 4 == 0 ? DIM / 4 : DIM) / 256;  // loads per thread: 16-byte pieces (single floats for 3 columns)
};
__device__ __forceinline__ uint32_t tile_window_start(uint32_t wkey, uint32_t w, uint32_t n) { return __umulhi(pcg_hash(wkey + w * 0x9E3779B9u), n); }

// The tile's loads are ISSUED at the start of the kernel (next to the event load) and LANDED in LDS when the sample's own loads
// are in flight: staging is off the critical path.  Hubness-weighted sampling (NodeSampler, embedder.rs:915-930: what
// examples/higgs.rs switches on) cannot use runs of consecutive rows: there every tile row is an independent draw of the alias
// table -- a slot picked uniformly afterwards is again a draw of the reference's law.  The draws themselves are made ahead, once
// per batch: a POOL of 2^24 (small graphs: 2^20) i.i.d. draws (sl_hub_pool_kernel), of which a tile takes kRows consecutive ones at
// a uniform offset -- again kRows independent draws, read with one coalesced request where every row used to cost a random 8-byte
// look-up of its own.
template <int DIM>
struct TileFetch {
    using T = TileShape<DIM>;
    static constexpr int Q = DIM % 4 == 0 ? DIM / 4 : DIM;  // pieces per row
    f4 pc[T::kPieces];
    uint32_t node[T::kPieces];
    // rows: the tile's first `rows` rows are distinct draws, the others repeat them (the step kernel's hub tile: 128 draws for 256 samples)
    // coherent: the rows are read past the caches (agent scope: what the other workgroups of this launch have written through)
    __device__ __forceinline__ void issue(const CeDev& c, uint32_t wkey, bool hub, const uint32_t* __restrict__ hub_pool, uint32_t hub_pool_n, uint32_t rows = (uint32_t)TileShape<DIM>::kRows,
                                          bool coherent = false) {
        const uint32_t pool_at = hub ? __umulhi(pcg_hash(wkey), hub_pool_n - (uint32_t)T::kRows) : 0u;
#pragma unroll
        for (int z = 0; z < T::kPieces; z++) {
            const uint32_t x = (uint32_t)z * 256u + threadIdx.x, r = (x / Q) & (rows - 1u);
            if (hub) {
                node[z] = hub_pool[pool_at + r];   // a run of the batch's pool: kRows independent draws of the alias table, one coalesced read
            } else {
                node[z] = tile_window_start(wkey, r / T::kL, (uint32_t)c.n) + r % T::kL;
                node[z] -= node[z] >= (uint32_t)c.n ? (uint32_t)c.n : 0u;
            }
        }
        const float* yb = c.yneg ? c.yneg : c.y;
#pragma unroll
        for (int z = 0; z < T::kPieces; z++) {
            const uint32_t x = (uint32_t)z * 256u + threadIdx.x, q = x % Q;
            if constexpr (DIM % 4 == 0) {
                const float* src = yb + (uint64_t)node[z] * c.ystride + 4u * q;
                if (coherent) {
                    const uint64_t lo = __hip_atomic_load(reinterpret_cast<const uint64_t*>(src), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const uint64_t hi = __hip_atomic_load(reinterpret_cast<const uint64_t*>(src) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    pc[z].x = __uint_as_float((uint32_t)lo); pc[z].y = __uint_as_float((uint32_t)(lo >> 32));
                    pc[z].z = __uint_as_float((uint32_t)hi); pc[z].w = __uint_as_float((uint32_t)(hi >> 32));
                } else {
                    pc[z] = *reinterpret_cast<const f4*>(src);
                }
            } else {
                const float* src = yb + (uint64_t)node[z] * c.ystride + q;
                pc[z].x = coherent ? __uint_as_float(__hip_atomic_load(reinterpret_cast<const uint32_t*>(src), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) : *src;
            }
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
// receives the slots, no window is used twice by a sample (TileShape); otherwise node ids (uniform, or hubness-weighted through the alias table), eight candidates at a time so that
// the alias look-ups overlap.  Returns the number accepted (5 unless the graph is tiny).
template <int DIM, int KMAX, bool TILE>
__device__ __forceinline__ uint32_t draw_negatives(const CeDev& c, bool hub, const uint32_t* s_tnode, uint32_t nb, uint32_t i,
                                                   const uint32_t (&nbr_reg)[KMAX], uint32_t (&out)[5], uint32_t tile_rows = (uint32_t)TileShape<DIM>::kRows) {
    using T = TileShape<DIM>;
    uint32_t got = 0;
    // the units a sample may use once: windows of T::kL = 16 rows; with hubness weighting single rows (every tile row is a draw of its own)
    const uint32_t wshift = hub ? 0u : 4u, units = hub ? tile_rows : tile_rows / (uint32_t)T::kL;
    static_assert(T::kL == 16 && (T::kW & (T::kW - 1)) == 0 && (T::kRows & (T::kRows - 1)) == 0, "draw_negatives: windows of 16 rows, power-of-two counts");
#pragma unroll
    for (int g = 0; g < 5; g++) out[g] = TILE ? 0u : i;
    for (uint32_t round = 0; round < 8u && got < 5u; round++) {
        uint32_t cand[8], slot[8];
        if constexpr (TILE) {
            // candidate number q of the sample sits in unit (u0 + q * stride) mod units, stride odd: a walk that visits every unit once
            // before it repeats -- consecutive candidates are in different windows by construction; the row inside the window is a
            // nibble of one more hash
            const uint32_t rows16 = pcg_hash(nb ^ (round * 0x9E3779B9u + 0x7F4A7C15u));
#pragma unroll
            for (int z = 0; z < 8; z++) {
                const uint32_t unit = ((nb >> 8) + (round * 8u + (uint32_t)z) * ((nb >> 16) | 1u)) & (units - 1u);
                const uint32_t row = hub ? unit : (unit << 4) | ((rows16 >> (4 * z)) & 15u);
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
            bool dup = false;
            if constexpr (TILE) {   // no window twice in a sample (TileShape): only once the walk has been round all units
                if (round * 8u + 8u > units) {
#pragma unroll
                    for (int g = 0; g < 5; g++) dup = dup || ((uint32_t)g < got && (out[g] >> wshift) == (slot[z] >> wshift));
                }
            }
            const bool ok = acc != 0u && got < 5u && !dup;
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

// (attract_f64 / repulse_f64 -- f64 scalars with ONE division per interaction -- live in ce_sample_math.h: the ordered dataflow uses them too)

// ce_optim_edge_shannon (embedder.rs:1167-1302) on yi / yj in registers, in its two phases: the attraction (one gradient, both ends
// -- the part of a sample that a chain through the target's row serialises) and the repulsions from the `got` drawn negatives (tile
// slots or node ids in `neg`; they move y_i only).  A negative's row may be rewritten during this launch by its owner: at most one
// launch old.  Gathered negatives of rows of <= 16 columns are requested ahead (`fetch`: a memory round trip); rows in the LDS tile and
// wider rows are read one at a time in `repulse` (40 / 80 registers less across a chain's turns: three waves per SIMD at 8 columns).
template <int DIM, bool F64, bool TILE>
struct SplitSample {
    static constexpr bool kAhead = DIM <= 16 && !TILE;
    float grad[DIM];
    float nrow[kAhead ? 5 : 1][DIM];
    __device__ __forceinline__ static void fetch_row(const CeDev& c, const float* s_tile, uint32_t x, float* row, bool coherent = false) {
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
            if (coherent) load_row_agent<DIM>(c.y, x, row, c.ystride);
            else load_row<DIM>((c.yneg ? c.yneg : c.y) + (uint64_t)x * c.ystride, 0u, row);
        }
    }
    __device__ __forceinline__ void fetch(const CeDev& c, const float* s_tile, const uint32_t (&neg)[5], bool coherent = false) {
        if constexpr (kAhead) {
#pragma unroll
            for (int g = 0; g < 5; g++) fetch_row(c, s_tile, neg[g], nrow[g], coherent);
        }
    }
    // (1 / s^2 in f64, once per call site: the compiler keeps it in registers across a sample's interactions)
    __device__ __forceinline__ static double inv_s2_f64(float scale_f) { return rcp_f64((double)scale_f * (double)scale_f); }
    __device__ __forceinline__ void attract(const CeDev& c, float* yi, float* yj, float w, float scale_f, double step) {
        if constexpr (F64) attract_f64<DIM>(yi, yj, grad, w, inv_s2_f64(scale_f), c.b, step);  // :1207-1238
        else attract_f32<DIM>(yi, yj, grad, w, rcp(scale_f * scale_f), (float)c.b, (float)step);
    }
    __device__ __forceinline__ void repulse_one(const CeDev& c, float* yi, const float* yk, float scale_f, double step) {
        if constexpr (F64) repulse_f64<DIM>(yi, yk, grad, inv_s2_f64(scale_f), c.b, step);  // :1267-1297
        else repulse_f32<DIM>(yi, yk, grad, rcp(scale_f * scale_f), (float)c.b, (float)step);
    }
    __device__ __forceinline__ void repulse(const CeDev& c, const float* s_tile, float* yi, float scale_f, double step, const uint32_t (&neg)[5], uint32_t got, bool coherent = false) {
        if constexpr (kAhead) {
#pragma unroll
            for (int g = 0; g < 5; g++)
                if ((uint32_t)g < got) repulse_one(c, yi, nrow[g], scale_f, step);
        } else {  // wide rows: one negative at a time (5 x 64 registers do not exist)
            for (uint32_t g = 0; g < got; g++) {
                uint32_t x = neg[0];
#pragma unroll
                for (int q = 1; q < 5; q++) x = g == (uint32_t)q ? neg[q] : x;
                fetch_row(c, s_tile, x, nrow[0], coherent);
                repulse_one(c, yi, nrow[0], scale_f, step);
            }
        }
    }
};
template <int DIM, bool F64, bool TILE>
__device__ __forceinline__ void run_sample(const CeDev& c, const float* s_tile, float* yi, float* yj, float w, float scale_f,
                                           double step, const uint32_t (&neg)[5], uint32_t got, bool coherent = false) {
    SplitSample<DIM, F64, TILE> sm;
    sm.fetch(c, s_tile, neg, coherent);
    sm.attract(c, yi, yj, w, scale_f, step);
    sm.repulse(c, s_tile, yi, scale_f, step, neg, got, coherent);
}

// ------------------------------------------------------------------------------------------------------------------
// a step: the events of one colour class in one slice
// ------------------------------------------------------------------------------------------------------------------
// A class is a forest of in-stars (slice_color_edges): no node is the source of two events of a step, none is source and target, but
// any number of events may share their TARGET.  Those are adjacent in the step's event range (the events are generated from an edge
// order sorted by target; the sort by (slice, class) is stable) and run as a CHAIN through the target's row:
//   * inside a 64-event chunk (one wave): every lane fetches its own event's data (the source's record and row, the negatives) in
//     parallel; the target's row is loaded by the chain's first lane and handed from lane to lane (one wave shuffle per turn): in turn
//     t the lanes at position t of their chains apply their attraction to the row they received.  A step without chains is turn 0
//     only -- the code path of a plain matching;
//   * across chunks (a chain that spans a chunk boundary, possibly a workgroup boundary): the last lane of the chunk writes the
//     target's row through to memory and publishes the step's token in the chunk's flag; lane 0 of the next chunk polls the flag and
//     reads the row with agent-scope loads before its turn.  Workgroups are dispatched in index order on every XCD and a chunk only
//     waits for the chunk before it, so the wait always ends (poll budget: error flag 2, never a hang).
// The reference serialises a hub's events through the row's lock at ~0.1 us per hand-over (embedder.rs:942,1185-1186,1239,1301); a
// turn here costs one attraction (the repulsions of a sample move y_i only: they run after the turns, all lanes side by side), instead of
// one launch per event and row as in the optimistic passes.
// Repeats of one edge inside a step (only without `spread`) are run by the first of their lanes, as before; their other lanes hand
// the target's row on untouched.
template <int DIM, int SREC, bool F64, bool TILE>
struct StepShared {
    using T = TileShape<DIM>;
    __attribute__((aligned(16))) float tile[TILE ? T::kRows * DIM : 4];
    __attribute__((aligned(16))) float stage[4 * kStageFloats<DIM, SREC>];
    uint32_t tnode[TILE ? T::kRows : 1];
};

// the events [a.begin, a.end) of one step, workgroup `block` of the step's grid (256 x a.ept events per workgroup)
// LINE > 0: the source's row, embedded scale and neighbour ids come as ONE line of LINE floats (NodeLine above; SREC == LINE then: it only
// sizes the stage), the edge's probability from the event
template <int DIM, int SREC, bool F64, bool TILE, int LINE = 0>
__device__ __forceinline__ void sl_step_body(const DirectArgs& a, uint32_t block, StepShared<DIM, SREC, F64, TILE>& sh, uint32_t& done) {
    constexpr int KREG = LINE > 0 ? LINE - DIM - 1 : ((SREC - 1) / 2 < 32 ? (SREC - 1) / 2 : 32);
    const CeDev c = a.c;
    const bool hub = c.hub_odds != nullptr;
    float* stage = sh.stage + (threadIdx.x >> 6) * kStageFloats<DIM, SREC>;
    const uint32_t nkey = pcg_hash((uint32_t)c.seed ^ a.key ^ kTagSlNeg);
    const uint32_t wkey = pcg_hash(nkey + a.step_seq * 0x85EBCA6Bu) + block * 64u;
    const uint32_t base = a.begin + block * 256u * a.ept;
    const int lane = threadIdx.x & 63;
    // one pass over 256 events; FIRST (a compile-time tag): the pass that also stages the tile.  Kept out of the loop below so that the
    // wait for the event load counts the tile's loads behind it (`vmcnt(pieces)`); with a run-time `r == 0` the compiler has to assume
    // the path without them and waits for everything: the tile's random reads became a hop of their own in front of the rows.
    auto pass = [&](uint32_t r, auto first_tag) {
        constexpr bool FIRST = decltype(first_tag)::value;
        const uint32_t p = base + r * 256u + threadIdx.x;
        // hop 1: the event and its two neighbours in the array (coalesced) and, beside them, this thread's share of the tile
        const bool act0 = p < a.end;
        Event e{0u, kNoNode}, pv{0u, kNoNode}, nx{0u, kNoNode};
        if (act0) {
            e = a.ev[p];
            if (p > a.begin) pv = a.ev[p - 1];
            if (p + 1 < a.end) nx = a.ev[p + 1];
        }
        TileFetch<DIM> ft;
        // Hubness-weighted tiles take 128 draws of the pool for the workgroup's 256 samples (round 6: half a request per event instead of
        // one; configs[3]'s graph 145 -> 139.5 ms per batch; ten seeds a side on 1 M Higgs-shaped points, 8 columns, tile forced: CE 0.9939
        // against 0.9938 with 256 draws, median edge 1.010 +- 0.004 against 1.006 +- 0.003 -- profiles/r06/r6_tile128_bias.txt).  Window
        // tiles keep their 16 windows (a sample takes no window twice: 8 would leave little choice).  AE_SL_DBG bit 256: 256 draws (A/B).
        const uint32_t tile_rows = (hub && !(a.dbg & 256) && TileShape<DIM>::kRows == 256) ? 128u : (uint32_t)TileShape<DIM>::kRows;
        if constexpr (TILE && FIRST) ft.issue(c, wkey, hub, a.hub_pool, a.hub_pool_n, tile_rows);
        const uint32_t i = e.im >> 5, j = act0 ? ev_node(e.j) : 0u;   // (an idle lane addresses row 0: the lane-group stores carry `want` in the index's top bit)
        const bool half = act0 && ev_half(e.j);   // (multi-GPU) the source is another shard's: attraction on the target's row only
        // chains: this event has the previous one's target / the next one has this one's
        const bool inrun = act0 && p > a.begin && ev_node(pv.j) == j;
        const bool next_inrun = act0 && p + 1u < a.end && ev_node(nx.j) == j;
        const bool absorbed = inrun && pv.im == e.im;   // a repeat of the previous event's edge: its first lane runs the repeats
        const bool cmp = act0 && !absorbed;
        uint32_t rep = 1;
        if (cmp && next_inrun && nx.im == e.im) { rep = 2; while (p + rep < a.end && a.ev[p + rep].im == e.im && ev_node(a.ev[p + rep].j) == j) rep++; }
        const unsigned long long run_mask = __ballot(inrun);
        const bool cont = (run_mask & 1ull) != 0ull;          // lane 0 continues a chain of the previous chunk (uniform over the wave)
        const unsigned long long heads = ~run_mask | 1ull;   // first lanes of the chunk's chain segments (a lone event is its own)
        const int head = 63 - __clzll(heads & ((2ull << lane) - 1ull));
        const uint32_t runpos = act0 ? (uint32_t)(lane - head) + ((cont && head == 0) ? 1u : 0u) : 0u;
        const bool last_in_seg = lane == 63 || !((run_mask >> (lane + 1)) & 1ull);
        const bool hand_over = act0 && lane == 63 && next_inrun;   // the chain goes on in the next chunk
        float yi[DIM], yj[DIM], scale_f = 1.f, w = 0.f;
        uint32_t nbr_reg[KREG];
        // hop 2: the source's record and both rows are requested together, then handed to their lanes (a target's row only by the
        // lane that starts its chain; a chain continued from the previous chunk receives it through memory below)
        RowFetch<DIM> fj;
        if constexpr (LINE > 0) {
            LineRec<DIM, LINE> fl;
            fl.issue(c.y, i, cmp);                            // :1185, and what RecFetch brings
            fj.issue(c.y, j, cmp && !inrun, yj, c.ystride);   // :1186
            if constexpr (TILE && FIRST) ft.land(sh.tile, sh.tnode);
            fl.land(stage, yi, scale_f, nbr_reg);
            fj.land(stage, yj);
            w = e.w;
        } else {
            RecFetch<SREC, KREG> fr;
            RowFetch<DIM> fi;
            const bool want_rec = cmp && !(a.dbg & 8);
            fr.issue(a.srec, i, e.im & 31u, want_rec, scale_f, w, nbr_reg);
            fi.issue(c.y, i, cmp, yi, c.ystride);             // :1185
            fj.issue(c.y, j, cmp && !inrun, yj, c.ystride);   // :1186
            if constexpr (TILE && FIRST) ft.land(sh.tile, sh.tnode);
            fr.land(stage, e.im & 31u, want_rec, scale_f, w, nbr_reg);
            fi.land(stage, yi);
            fj.land(stage, yj);
            if (a.dbg & 8) { for (int q = 0; q < KREG; q++) nbr_reg[q] = 0xFFFFFFFFu; w = 0.5f; scale_f = 1.f; }
        }
        const uint32_t chunk = (a.begin >> 6) + ((p - a.begin) >> 6);
        // the sample's last repetition (its only one unless the edge repeats inside the step): negatives drawn and their rows requested
        // now; its attraction runs in the lane's turn, its repulsions -- they move y_i only -- after the turns, all lanes side by side
        SplitSample<DIM, F64, TILE> sm;
        uint32_t neg[5], got = 0;
        if (cmp && !half) {
            got = draw_negatives<DIM, KREG, TILE>(c, hub, sh.tnode, pcg_hash(nkey + (p + rep - 1u)), i, nbr_reg, neg, tile_rows);
            if (a.dbg & 4) { got = 0; for (int g = 0; g < 5; g++) neg[g] = TILE ? 0u : i; }
            sm.fetch(c, sh.tile, neg);
        }
        for (uint32_t t = 0;; t++) {
            if (t >= 1u) {   // (only chunks with chains get here)
                float in[DIM];
#pragma unroll
                for (int q = 0; q < DIM; q++) in[q] = __shfl_up(yj[q], 1);
                if (t == 1u && cont && lane == 0) {   // the target's row as the previous chunk left it
                    const uint32_t token = a.step_seq + 1u;
                    uint32_t polls = 0;
                    while (__hip_atomic_load(&a.chunk_flag[chunk - 1u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != token) {
                        __builtin_amdgcn_s_sleep(1);
                        if (++polls > (1u << 24)) { atomicOr(reinterpret_cast<unsigned int*>(a.done_counter + 1024), 2u); break; }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    load_row_agent<DIM>(c.y, j, in, c.ystride);
                }
                if (act0 && runpos == t) {
#pragma unroll
                    for (int q = 0; q < DIM; q++) yj[q] = in[q];
                }
            }
            if (cmp && runpos == t && !(a.dbg & 1)) {
                for (uint32_t q = 0; q + 1u < rep; q++) {   // earlier repetitions of the edge: whole samples, one after the other
                    uint32_t ng[5] = {0u, 0u, 0u, 0u, 0u};
                    const uint32_t gt = half ? 0u : draw_negatives<DIM, KREG, TILE>(c, hub, sh.tnode, pcg_hash(nkey + (p + q)), i, nbr_reg, ng, tile_rows);
                    run_sample<DIM, F64, TILE>(c, sh.tile, yi, yj, w, scale_f, a.step, ng, gt);
                }
                sm.attract(c, yi, yj, w, scale_f, a.step);
            }
            if (!__ballot(act0 && runpos > t)) break;
        }
        if (cmp && !half) {
            if (!(a.dbg & 1)) sm.repulse(c, sh.tile, yi, scale_f, a.step, neg, got);
            else if (got == 77u) yi[0] += (float)neg[0];
            done += rep;   // (a half event is counted by the shard that owns its source)
        }
        // stores: the source's row by every lane that computed, the target's row by the last lane of its chain segment -- written through
        // and announced where the chain goes on in the next chunk
        const bool store_j = act0 && last_in_seg && !hand_over;
        if (!(a.dbg & 2)) {
            const int st_mode = (a.dbg & 512) ? 2 : ((a.dbg & 128) ? 1 : 0);
            row_store<DIM>(c.y, j, store_j, stage, yj, c.ystride, st_mode);  // :1239
            row_store<DIM>(c.y, i, cmp && !half, stage, yi, c.ystride, st_mode);      // :1301
        } else if (yi[0] == 1.2345e-30f && yj[0] == 3.4e-30f) {
            row_store<DIM>(c.y, i, cmp && !half, stage, yi, c.ystride);
        }
        if (hand_over) {
            store_row_agent<DIM>(c.y, j, yj, c.ystride);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __hip_atomic_store(&a.chunk_flag[chunk], a.step_seq + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    if (base < a.end) pass(0u, std::true_type{});  // (uniform over the workgroup)
    for (uint32_t r = 1; r < a.ept && base + r * 256u < a.end; r++) pass(r, std::false_type{});
}

// (rows of <= 8 columns with f64 scalars and the LDS tile -- the big-graph path --: three waves per SIMD -- 168 VGPRs -- are worth a spill or two; the one-division arithmetic
// sits at 171 without the bound, and a resident step of 190 k events instead of 127 k is what the slice count is sized by)
template <int DIM, int SREC, bool F64, bool TILE, int LINE = 0>
__global__ void __launch_bounds__(256, (F64 && TILE && DIM <= 8) ? 3 : 1) sl_direct_kernel(DirectArgs a) {
    __shared__ StepShared<DIM, SREC, F64, TILE> sh;
    uint32_t done = 0;
    sl_step_body<DIM, SREC, F64, TILE, LINE>(a, blockIdx.x, sh, done);
    for (int off = 32; off > 0; off >>= 1) done += __shfl_xor(done, off);
    if ((threadIdx.x & 63) == 0 && done) atomicAdd(&a.done_counter[(blockIdx.x * 4u + (threadIdx.x >> 6)) & 1023u], (unsigned long long)done);
}

// ------------------------------------------------------------------------------------------------------------------
// a whole SLICE in one launch: the classes of a slice side by side, ordered node by node instead of launch by launch
// ------------------------------------------------------------------------------------------------------------------
// A step with few events (a rank's share of a sharded batch, any graph of ~10^6 nodes) is a chain of latencies: 19-24 us however
// little it holds, and a slice is k + 5 of them.  But only ~17 % of a slice's events share a node with another event of the slice
// (half an event per node and slice): the launch boundary between two classes orders ALL events of the one before ALL of the other
// where the law only asks for an order between events that share a node.  Here every class of a slice runs in ONE launch, a workgroup
// holding events of one class only, and the order between two events on a node is kept node by node:
//   * a per-node 64-bit word holds the classes that have an event on the node (low half) and the classes that are through with it
//     (high half).  The low halves are entered by the slice BEFORE this one, from inside its launch (a wave that is through enters
//     the next slice's events: two fire-and-forget atomics per event, beside the other waves' work), into the other of two sets of
//     words; a slice without such a predecessor takes a pass of its own (sl_dep_mark_kernel, ce_slice.hip).
//   * LANE BY LANE: a lane whose node has an earlier class polls that class's bit (a budget instead of a hang), then reads the row
//     with agent-scope loads -- while the other lanes of its wave run (the body is a loop over the lanes that are ready); a lane whose
//     node has a later class writes the row through (agent-scope stores), waits for them and sets its bit: the target's row the
//     moment its chain's last lane has attracted, the source's row after its repulsions.  The node's last event of the slice wipes
//     the word.  The other ~83 % run as in sl_step_body.
//   * Workgroups are dispatched in index order and a class's workgroups come before the next class's: an event only ever waits for a
//     workgroup that is already running.  Chains through a shared target take their turns inside the same loop; the chain as a whole
//     takes the target's place in the node's order (its head waits, its last lane announces it).
//   * Where the words live: rows of 8 / 2 columns keep them in the node's own line of the batch's internal copy (LineShape: one
//     request brings row and word, the row's store takes the wiped word along), rows of 3 / 4 columns behind the row, longer rows
//     in an array of their own (SliceRunArgs::dep_stride).
constexpr uint32_t kDepBits = 32;                       // classes a merged slice can hold
constexpr uint32_t kErrDepPoll = 16u;                   // done_counter[1024] flag: a dependency inside a merged slice was not met within the poll budget
struct SliceRunArgs {
    DirectArgs d;                  // c, srec, ev, key, step, done_counter, chunk_flag, hub_pool (begin / end / step_seq / ept / tile: per class, below)
    const uint32_t* sptr;          // this slice's class pointers: class position q = [sptr[q], sptr[q + 1])
    uint32_t classes;
    uint32_t step_seq0;            // running step number of the slice's first class
    unsigned long long* dep;       // [n]: classes through with the node << 32 | classes with an event on the node; all zero between slices
    const uint32_t* next_sptr;     // the NEXT slice's class pointers (nullptr: nothing to prepare): its events enter `dep_next` while this slice runs
    unsigned long long* dep_next;  // [n]: the other set of words (a slice cleans its own set as it goes)
    uint32_t set;                  // 0 / 1: which of the two sets of words this slice runs on
    uint32_t lines;                // the words sit in the node's line around its row (LineShape: rows of 8 / 2 columns)
    uint32_t dep_stride;           // 64-bit words from one node's word to the next: 1 (an array of its own), or the coordinate rows' stride / 2 where the words
                                   // sit behind the node's row (rows of <= 8 columns: the word comes with the row's line and is wiped with its store)
    // THE CLASS WINDOW (round 6): a workgroup of class position q starts -- stages its tile, reads its rows, draws its negatives -- only once the
    // classes at positions <= q - window are through, every row is stored through the caches, and the negatives are read past them: the
    // negatives' rows are `window` steps old at most, where a merged launch without it reads them as the slice found them (the form's
    // published bias: DESIGN 4.3b).  0: no window.
    uint32_t window;
    uint32_t* class_done;          // [classes]: workgroups of every class position that are through (zeroed per batch)
};
constexpr uint32_t kErrWindowPoll = 32u;                // done_counter[1024] flag: a workgroup waited for the classes before its window beyond the poll budget
// an event of a slice enters its class in the words of its two nodes (fire-and-forget atomics on the words' low halves)
__device__ __forceinline__ void dep_mark_event(unsigned long long* dep, uint32_t dep_stride, const Event e, uint32_t q) {
    uint32_t* words = reinterpret_cast<uint32_t*>(dep);   // (little endian: the low word of a node's 64 bits holds the classes)
    if (!ev_half(e.j)) atomicOr(words + 2ull * dep_stride * (e.im >> 5), 1u << q);   // (a half event reads its source's row from the replica: nobody here writes it)
    atomicOr(words + 2ull * dep_stride * ev_node(e.j), 1u << q);
}
__device__ __forceinline__ uint32_t dep_classes(unsigned long long w) { return (uint32_t)w; }
__device__ __forceinline__ uint32_t dep_done(unsigned long long w) { return (uint32_t)(w >> 32); }
// (sl_dep_mark_kernel, the pass over a slice's events that fills the words before the launch: ce_slice.hip)
template <int DIM, int SREC, bool F64, bool TILE>
__global__ void __launch_bounds__(256) sl_slice_kernel(SliceRunArgs ra) {
    constexpr int KREG = (SREC - 1) / 2 < 32 ? (SREC - 1) / 2 : 32;
    __shared__ StepShared<DIM, SREC, F64, TILE> sh;
    const DirectArgs& a = ra.d;
    const CeDev c = a.c;
    const bool hub = c.hub_odds != nullptr;
    float* stage = sh.stage + (threadIdx.x >> 6) * kStageFloats<DIM, SREC>;
    const uint32_t nkey = pcg_hash((uint32_t)c.seed ^ a.key ^ kTagSlNeg);
    const int lane = threadIdx.x & 63;
    // this workgroup's class and its place in the class (uniform)
    uint32_t q = 0, block = blockIdx.x, s_begin = ra.sptr[0], s_end = ra.sptr[1];
    for (;;) {
        const uint32_t nb = (s_end - s_begin + 255u) >> 8;
        if (block < nb || q + 1u >= ra.classes) break;
        block -= nb;
        q++;
        s_begin = s_end;
        s_end = ra.sptr[q + 1u];
    }
    const uint32_t s_seq = ra.step_seq0 + q;
    const uint32_t wkey = pcg_hash(nkey + s_seq * 0x85EBCA6Bu) + block * 64u;
    const uint32_t p = s_begin + block * 256u + threadIdx.x;
    uint32_t done = 0;
    const bool through = ra.window != 0u;   // (uniform) every row store leaves the caches as it is written
    // the class window: before anything of this workgroup reads a negative's row, the class `window` positions before its own is through
    // (dispatched before this workgroup: it runs or is through; ONE counter is polled -- the classes before that one were held to
    // the same rule, and what the window bounds is the age of the negatives' rows, a statistical matter: the node-by-node order is the
    // dependency words' business)
    auto window_wait = [&]() {
        if (ra.window && q >= ra.window) {
            if (threadIdx.x == 0) {
                const uint32_t qq = q - ra.window, want = (ra.sptr[qq + 1u] - ra.sptr[qq] + 255u) >> 8;
                uint32_t polls = 0;
                while (__hip_atomic_load(&ra.class_done[qq], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++polls > (1u << 22)) { atomicOr(reinterpret_cast<unsigned int*>(a.done_counter + 1024), kErrWindowPoll); break; }
                }
            }
            __syncthreads();
        }
    };
    // hop 1: the event and its two neighbours in the array, the tile
    const bool act0 = p < s_end;
    Event e{0u, kNoNode}, pv{0u, kNoNode}, nx{0u, kNoNode};
    if (act0) {
        e = a.ev[p];
        if (p > s_begin) pv = a.ev[p - 1];
        if (p + 1 < s_end) nx = a.ev[p + 1];
    }
    TileFetch<DIM> ft;
    if constexpr (TILE) if (!through) ft.issue(c, wkey, hub, a.hub_pool, a.hub_pool_n);
    const uint32_t i = e.im >> 5, j = act0 ? ev_node(e.j) : 0u;
    const bool half = act0 && ev_half(e.j);
    const bool inrun = act0 && p > s_begin && ev_node(pv.j) == j;
    const bool next_inrun = act0 && p + 1u < s_end && ev_node(nx.j) == j;
    const bool absorbed = inrun && pv.im == e.im;
    const bool cmp = act0 && !absorbed;
    uint32_t rep = 1;
    if (cmp && next_inrun && nx.im == e.im) { rep = 2; while (p + rep < s_end && a.ev[p + rep].im == e.im && ev_node(a.ev[p + rep].j) == j) rep++; }
    const unsigned long long run_mask = __ballot(inrun);
    const bool last_in_seg = lane == 63 || !((run_mask >> (lane + 1)) & 1ull);
    const bool hand_over = act0 && lane == 63 && next_inrun;
    float yi[DIM], yj[DIM], scale_f = 1.f, w = 0.f;
    uint32_t nbr_reg[KREG];
    // hop 2: record, rows -- and the two dependency words
    RecFetch<SREC, KREG> fr;
    RowFetch<DIM> fi, fj;
    unsigned long long wi = 0ull, wj = 0ull;
    unsigned long long* const dep_i = ra.dep + (uint64_t)i * ra.dep_stride;
    unsigned long long* const dep_j = ra.dep + (uint64_t)j * ra.dep_stride;
    // (the classes were entered by the launch before this one: plain loads -- behind the row, the word comes with the row's line)
    const bool lines = kHasLines<DIM> && ra.lines != 0u;   // (uniform) the words in the node's line
    if (lines) {
        if constexpr (kHasLines<DIM>) {
            LineFetch<DIM> li, lj;
            fr.issue(a.srec, i, e.im & 31u, cmp, scale_f, w, nbr_reg);
            li.issue(c.y, i, cmp);
            lj.issue(c.y, j, act0);   // (a chain's followers want the word: the last of them announces the chain)
            if (through) {   // (the event's own requests are on their way while the workgroup waits for its window)
                window_wait();
                if constexpr (TILE) ft.issue(c, wkey, hub, a.hub_pool, a.hub_pool_n, (uint32_t)TileShape<DIM>::kRows, true);
            }
            if constexpr (TILE) ft.land(sh.tile, sh.tnode);
            fr.land(stage, e.im & 31u, cmp, scale_f, w, nbr_reg);
            li.land(stage, yi, wi, ra.set);
            lj.land(stage, yj, wj, ra.set);
            if (half) wi = 0ull;
        }
    } else {
        if (cmp && !half && !(a.dbg & 32)) wi = *dep_i;   // (dbg 16 / 32: timing experiments, wrong results)
        if (act0 && !(a.dbg & 32)) wj = *dep_j;
        fr.issue(a.srec, i, e.im & 31u, cmp, scale_f, w, nbr_reg);
        fi.issue(c.y, i, cmp, yi, c.ystride);
        fj.issue(c.y, j, cmp && !inrun, yj, c.ystride);
        if (through) {
            window_wait();
            if constexpr (TILE) ft.issue(c, wkey, hub, a.hub_pool, a.hub_pool_n, (uint32_t)TileShape<DIM>::kRows, true);
        }
        if constexpr (TILE) ft.land(sh.tile, sh.tnode);
        fr.land(stage, e.im & 31u, cmp, scale_f, w, nbr_reg);
        fi.land(stage, yi);
        fj.land(stage, yj);
    }
    // the node-by-node order: classes before this one that hold the node must be through with it; classes after it will wait for us
    const uint32_t below = (1u << q) - 1u;
    const uint32_t above = q >= 31u ? 0u : ~((2u << q) - 1u);
    const uint32_t need_i = (cmp && !half) ? (dep_classes(wi) & below) : 0u;
    const uint32_t need_j = (cmp && !inrun) ? (dep_classes(wj) & below) : 0u;   // (a chain's head stands for the chain)
    const bool succ_i = cmp && !half && (dep_classes(wi) & above) != 0u;
    const bool succ_j = act0 && (dep_classes(wj) & above) != 0u;
    const uint32_t chunk = (s_begin >> 6) + ((p - s_begin) >> 6);
    SplitSample<DIM, F64, TILE> sm;
    uint32_t neg[5], got = 0;
    if (cmp && !half) {   // (the negatives do not wait for anybody: drawn and requested before the first poll)
        got = draw_negatives<DIM, KREG, TILE>(c, hub, sh.tnode, pcg_hash(nkey + (p + rep - 1u)), i, nbr_reg, neg);
        sm.fetch(c, sh.tile, neg, through);
    }
    // LANE BY LANE: a lane runs as soon as ITS OWN predecessors are through -- the classes before it on its two nodes, the lane
    // before it in a chain -- and announces its rows at once.  (The first form waited for all 64 lanes' predecessors before any
    // lane ran: since nearly every wave holds some lane with a predecessor, and that one sat in a wave that waited likewise, the
    // slice ran as a chain of waves k + 5 deep -- 140 us for 344 k events -- although an event's own chain is 1-2 events long.)
    //   phase 0: waiting / ready for the attraction; 1: attracted, the target's row handed on; 2: through.
    // The repulsions (y_i only) and the stores wait while a chain is taking its turns in this wave: a turn stays one attraction.
    const bool follows = inrun && lane > 0;            // the target's row comes from the lane before (wave shuffle)
    const bool from_prev_chunk = inrun && lane == 0;   // ... from the chunk before (memory, the chunk's flag)
    const bool store_j = act0 && last_in_seg && !hand_over;
    const bool store_i = cmp && !half;
    uint32_t phase = act0 ? 0u : 2u;
    bool ok_dep = !(need_i | need_j), ok_flag = !from_prev_chunk;
    uint32_t idle = 0;
    for (;;) {
        float in[DIM];
#pragma unroll
        for (int z = 0; z < DIM; z++) in[z] = __shfl_up(yj[z], 1);
        const uint32_t pphase = __shfl_up(phase, 1);
        if (phase == 0u && !ok_dep) {
            const bool ok_i = !need_i || (dep_done(__hip_atomic_load(dep_i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) & need_i) == need_i;
            const bool ok_j = !need_j || (dep_done(__hip_atomic_load(dep_j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) & need_j) == need_j;
            ok_dep = ok_i && ok_j;
        }
        if (phase == 0u && !ok_flag) ok_flag = __hip_atomic_load(&a.chunk_flag[chunk - 1u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == s_seq + 1u;
        const bool go = phase == 0u && ok_dep && ok_flag && (!follows || pphase >= 1u);
        if (go) {
            if (follows) {
#pragma unroll
                for (int z = 0; z < DIM; z++) yj[z] = in[z];
            } else if (from_prev_chunk) {   // the target's row as the previous chunk left it
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                load_row_agent<DIM>(c.y, j, yj, c.ystride);
            } else if (need_j) load_row_agent<DIM>(c.y, j, yj, c.ystride);
            if (need_i) load_row_agent<DIM>(c.y, i, yi, c.ystride);
            if (cmp) {
                for (uint32_t z = 0; z + 1u < rep; z++) {   // earlier repetitions of the edge: whole samples, one after the other
                    uint32_t ng[5] = {0u, 0u, 0u, 0u, 0u};
                    const uint32_t gt = half ? 0u : draw_negatives<DIM, KREG, TILE>(c, hub, sh.tnode, pcg_hash(nkey + (p + z)), i, nbr_reg, ng);
                    run_sample<DIM, F64, TILE>(c, sh.tile, yi, yj, w, scale_f, a.step, ng, gt, through);
                }
                sm.attract(c, yi, yj, w, scale_f, a.step);
            }
            // the target's row is final for this class once its chain segment's last lane has attracted: handed to the next chunk, or
            // to the classes behind, NOW (write-through stores, waited for, then the flag / the bit: no cache write-back in between)
            if (hand_over) {
                store_row_agent<DIM>(c.y, j, yj, c.ystride);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                __hip_atomic_store(&a.chunk_flag[chunk], s_seq + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else if (store_j && succ_j) {
                store_row_agent<DIM>(c.y, j, yj, c.ystride);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                atomicOr(dep_j, 1ull << (32u + q));
            }
            phase = 1u;
        }
        const bool moving = __ballot(go && next_inrun && lane < 63) != 0ull;   // a lane of this wave takes its turn next time round
        if (!moving && __ballot(phase == 1u)) {
            const bool fin = phase == 1u;
            if (fin && store_i) {
                if constexpr (SplitSample<DIM, F64, TILE>::kAhead) {
                    // AE_SL_DBG bit 64 (an experiment on the form's bias, tools/run_blobs_forms.py): the negatives' rows are read NOW, past the
                    // caches (agent scope: what the other workgroups of this launch have written through), instead of before the first poll
                    if (a.dbg & 64) {
#pragma unroll
                        for (int g = 0; g < 5; g++) load_row_agent<DIM>(c.y, neg[g], sm.nrow[g], c.ystride);
                    }
                }
                sm.repulse(c, sh.tile, yi, scale_f, a.step, neg, got, through);
                done += rep;
            }
            // stores: as sl_step_body for the rows no later class of the slice will ask for; the node's last event of the slice wipes its
            // word (everybody who had to read it has: they are earlier classes, or this chain)
            if (lines) {   // (the row's store takes the wiped word along: one request)
                if constexpr (kHasLines<DIM>) {
                    line_store<DIM>(c.y, j, fin && store_j && !succ_j, stage, yj, ra.set, through);   // :1239
                    line_store<DIM>(c.y, i, fin && store_i && !succ_i, stage, yi, ra.set, through);   // :1301
                }
            } else {
                row_store<DIM>(c.y, j, fin && store_j && !succ_j, stage, yj, c.ystride, through ? 2 : 0);   // :1239
                row_store<DIM>(c.y, i, fin && store_i && !succ_i, stage, yi, c.ystride, through ? 2 : 0);   // :1301
                if (fin && store_i && !succ_i && !(a.dbg & 16)) *dep_i = 0ull;
                if (fin && store_j && !succ_j && !(a.dbg & 16)) *dep_j = 0ull;
            }
            if (fin && store_i && succ_i) {
                store_row_agent<DIM>(c.y, i, yi, c.ystride);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                atomicOr(dep_i, 1ull << (32u + q));
            }
            if (fin) phase = 2u;
        }
        if (!__ballot(phase != 2u)) break;
        if (!__ballot(go)) {
            __builtin_amdgcn_s_sleep(1);
            if (++idle > (1u << 22)) {   // a budget instead of a hang: the batch fails with the flag's message
                if (phase != 2u) atomicOr(reinterpret_cast<unsigned int*>(a.done_counter + 1024), (!ok_flag) ? 2u : kErrDepPoll);
                break;
            }
        } else idle = 0;
    }
    if (ra.window) {   // this workgroup's rows are out (the stores waited for): its class counts it
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(&ra.class_done[q], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // The NEXT slice's events enter their words now: two fire-and-forget atomics per event (~24 G/s: 29 us for a slice of 344 k events
    // as a launch of its own) that nothing in this slice waits for -- issued when a wave is through, they run beside the other waves' work.
    if (ra.next_sptr) {
        // (the class pointers are read as uniform scalar loads, not shuffled out of a register: the loop's trip count differs from lane
        // to lane, and a shuffle FROM a lane that has left the loop is undefined -- an empty trailing class would have been marked
        // in a wrong class bit)
        const uint32_t n_begin = ra.next_sptr[0], n_end = ra.next_sptr[ra.classes];
        for (uint32_t t = n_begin + blockIdx.x * 256u + threadIdx.x; t < n_end; t += gridDim.x * 256u) {
            uint32_t qn = 0;
            for (uint32_t l = 1; l < ra.classes; l++) qn += t >= ra.next_sptr[l] ? 1u : 0u;
            dep_mark_event(ra.dep_next, ra.dep_stride, a.ev[t], qn);
        }
    }
    for (int off = 32; off > 0; off >>= 1) done += __shfl_xor(done, off);
    if ((threadIdx.x & 63) == 0 && done) atomicAdd(&a.done_counter[(blockIdx.x * 4u + (threadIdx.x >> 6)) & 1023u], (unsigned long long)done);
}
template <int DIM, bool F64, bool TILE>
void launch_slice3(const SliceRunArgs& a, unsigned grid, uint32_t srec) {
    if (srec == 16) hipLaunchKernelGGL((sl_slice_kernel<DIM, 16, F64, TILE>), dim3(grid), dim3(256), 0, stream(), a);
    else if (srec == 32) hipLaunchKernelGGL((sl_slice_kernel<DIM, 32, F64, TILE>), dim3(grid), dim3(256), 0, stream(), a);
    else if (srec == 64) hipLaunchKernelGGL((sl_slice_kernel<DIM, 64, F64, TILE>), dim3(grid), dim3(256), 0, stream(), a);
    else hipLaunchKernelGGL((sl_slice_kernel<DIM, 128, F64, TILE>), dim3(grid), dim3(256), 0, stream(), a);
}
template <int DIM>
void launch_slice(const SliceRunArgs& a, unsigned grid, uint32_t srec, bool f64, bool tile_wanted) {
    const bool tile = tile_wanted && a.d.c.n > (uint64_t)TileShape<DIM>::kRows * 4ull;
    if (f64) { if (tile) launch_slice3<DIM, true, true>(a, grid, srec); else launch_slice3<DIM, true, false>(a, grid, srec); }
    else { if (tile) launch_slice3<DIM, false, true>(a, grid, srec); else launch_slice3<DIM, false, false>(a, grid, srec); }
}

// (A persistent form of this kernel -- the class steps of a run of slices in ONE launch, a grid barrier between steps, the next step's
// events and static records requested before the wait -- was built and measured in round 5 and is not kept: on a rank's share of a
// configs[3] batch, 31 k events per step, it ran 24.3 us per step against 23.6 us for one launch per step; the counter barrier with its
// release / acquire fences costs what the prefetch saves, and even a free barrier would have bought 15 ms of 71.  profiles/r05/
// r5_rank_share_notes.md; the code: commit 9deaf5d, `git show 9deaf5d:annembed_amd/csrc/ce_slice_kernels.h`.)

// ------------------------------------------------------------------------------------------------------------------
// the overflow class of a slice: optimistic passes (sl_mark_kernel: ce_slice.hip)
// ------------------------------------------------------------------------------------------------------------------
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
        const uint32_t pj = ev_node(p.j);
        const bool half = ev_half(p.j);
        const uint32_t o1 = own_chk[p.im >> 5], o2 = own_chk[pj];
        TileFetch<DIM> ft;
        if constexpr (TILE && FIRST) ft.issue(c, wkey, hub, a.hub_pool, a.hub_pool_n);
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
        fi.issue(c.y, i, win, yi, c.ystride);
        fj.issue(c.y, pj, win, yj, c.ystride);
        if constexpr (TILE && FIRST) ft.land(s_tile, s_tnode);
        fr.land(stage, p.im & 31u, win, scale_f, w, nbr_reg);
        fi.land(stage, yi);
        fj.land(stage, yj);
        if (win) {
            uint32_t neg[5] = {0u, 0u, 0u, 0u, 0u};
            const uint32_t got = half ? 0u : draw_negatives<DIM, KREG, TILE>(c, hub, s_tnode, pcg_hash(nkey + idx), i, nbr_reg, neg);
            run_sample<DIM, F64, TILE>(c, s_tile, yi, yj, w, scale_f, a.step, neg, got);
            done += half ? 0u : 1u;
        }
        row_store<DIM>(c.y, pj, win, stage, yj, c.ystride);           // :1239
        row_store<DIM>(c.y, i, win && !half, stage, yi, c.ystride);   // :1301
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
                    own_mark[pj] = idx;
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

constexpr uint32_t kNil = 0xFFFFFFFFu;
// chain rounds (link / unlink kernels and the description: ce_slice.hip)
template <int DIM, int SREC, bool F64>
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
        const uint32_t tj = ev_node(p0.j);
        if (head[tj] != pos) continue;  // the head of its target's list walks it
        float yj[DIM];
        load_row<DIM>(c.y + (uint64_t)tj * c.ystride, 0u, yj);
        uint32_t cur = pos, nxt = next[pos];
        Pending e = p0;
        for (;;) {
            Pending en{0u, 0u, 0u, 0u};
            uint32_t nn = kNil;
            if (nxt != kNil) { en = src[nxt]; nn = next[nxt]; }  // the next link travels while this event runs
            const uint32_t i = e.im >> 5;
            if (a.owner[i] == e.idx && head[i] == kNil) {
                float yi[DIM];
                load_row<DIM>(c.y + (uint64_t)i * c.ystride, 0u, yi);
                const float* r = a.srec + (uint64_t)i * SREC;
                const float scale_f = r[0], w = r[1 + KP + (e.im & 31u)];
                uint32_t nbr_reg[KREG];
#pragma unroll
                for (int q = 0; q < KREG; q++) nbr_reg[q] = __float_as_uint(r[1 + q]);
                uint32_t neg[5] = {0u, 0u, 0u, 0u, 0u};
                const bool half = ev_half(e.j);
                const uint32_t got = half ? 0u : draw_negatives<DIM, KREG, false>(c, hub, nullptr, pcg_hash(nkey + e.idx), i, nbr_reg, neg);
                run_sample<DIM, F64, false>(c, nullptr, yi, yj, w, scale_f, a.step, neg, got);
                if (!half) { store_row<DIM>(c.y + (uint64_t)i * c.ystride, 0u, yi); done++; }  // :1301
            } else {  // the source is claimed by another event or is a target of this round: next round
                const uint32_t dsub = (cur + blockIdx.x) % (uint32_t)kSub;
                const uint32_t at = atomicAdd(&a.counts[a.dst_list * kSub + dsub], 1u);
                if (at < a.cap) a.lists[((uint64_t)a.dst_list * kSub + dsub) * a.cap + at] = e;
                else atomicOr(reinterpret_cast<unsigned int*>(a.done_counter + 1024), 1u);
            }
            if (nxt == kNil) break;
            cur = nxt; e = en; nxt = nn;
        }
        store_row<DIM>(c.y + (uint64_t)tj * c.ystride, 0u, yj);  // :1239
    }
    for (int off = 32; off > 0; off >>= 1) done += __shfl_xor(done, off);
    if ((threadIdx.x & 63) == 0 && done) atomicAdd(&a.done_counter[(blockIdx.x * 4u + (threadIdx.x >> 6) + blockIdx.y * 64u) & 1023u], done);
    if (blockIdx.x == 0 && threadIdx.x == 0) a.counts[a.zero_list * kSub + sub] = 0;
}
template <int DIM, bool F64>
void launch_chain_run2(const SliceArgs& a, unsigned grid, uint32_t srec, const uint32_t* head, const uint32_t* next) {
    if (srec == 16) hipLaunchKernelGGL((sl_chain_run_kernel<DIM, 16, F64>), dim3(grid, kSub), dim3(256), 0, stream(), a, head, next);
    else if (srec == 32) hipLaunchKernelGGL((sl_chain_run_kernel<DIM, 32, F64>), dim3(grid, kSub), dim3(256), 0, stream(), a, head, next);
    else if (srec == 64) hipLaunchKernelGGL((sl_chain_run_kernel<DIM, 64, F64>), dim3(grid, kSub), dim3(256), 0, stream(), a, head, next);
    else hipLaunchKernelGGL((sl_chain_run_kernel<DIM, 128, F64>), dim3(grid, kSub), dim3(256), 0, stream(), a, head, next);
}
template <int DIM>
void launch_chain_run(const SliceArgs& a, unsigned grid, uint32_t srec, bool f64, const uint32_t* head, const uint32_t* next) {
    if (f64) launch_chain_run2<DIM, true>(a, grid, srec, head, next); else launch_chain_run2<DIM, false>(a, grid, srec, head, next);
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
// workgroups of the step kernel that one CU holds at a time (registers / LDS of the instantiation that would run)
template <int DIM, bool F64, bool TILE>
int direct_blocks_per_cu3(uint32_t srec) {
    int nb = 0;
    hipError_t e;
    if (srec == 16) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, sl_direct_kernel<DIM, 16, F64, TILE>, 256, 0);
    else if (srec == 32) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, sl_direct_kernel<DIM, 32, F64, TILE>, 256, 0);
    else if (srec == 64) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, sl_direct_kernel<DIM, 64, F64, TILE>, 256, 0);
    else e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, sl_direct_kernel<DIM, 128, F64, TILE>, 256, 0);
    if (e != hipSuccess || nb < 1) nb = 1;
    return nb;
}
template <int DIM>
void direct_blocks_per_cu(uint32_t srec, bool f64, bool tile, int* out) {
    if (f64) *out = tile ? direct_blocks_per_cu3<DIM, true, true>(srec) : direct_blocks_per_cu3<DIM, true, false>(srec);
    else *out = tile ? direct_blocks_per_cu3<DIM, false, true>(srec) : direct_blocks_per_cu3<DIM, false, false>(srec);
}
template <int DIM>
void launch_direct(const DirectArgs& a, uint32_t srec, bool f64) {
    const bool tile = a.tile && a.c.n > (uint64_t)TileShape<DIM>::kRows * 4ull;
    if (f64) { if (tile) launch_direct3<DIM, true, true>(a, srec); else launch_direct3<DIM, true, false>(a, srec); }
    else { if (tile) launch_direct3<DIM, false, true>(a, srec); else launch_direct3<DIM, false, false>(a, srec); }
}

// the step kernel on node lines (LineRec): LINE in {8, 16, 32} floats, longer than the row and its scale
template <int DIM, int LINE, bool F64, bool TILE>
void launch_direct_line4(const DirectArgs& a, int* blocks_per_cu) {
    if constexpr (DIM <= 16 && LINE > DIM + 1) {
        if (blocks_per_cu) {   // occupancy query (sl_resident_events), no launch
            int nb = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, sl_direct_kernel<DIM, LINE, F64, TILE, LINE>, 256, 0) != hipSuccess || nb < 1) nb = 1;
            *blocks_per_cu = nb;
            return;
        }
        const unsigned grid = (a.end - a.begin + 256u * a.ept - 1u) / (256u * a.ept);
        hipLaunchKernelGGL((sl_direct_kernel<DIM, LINE, F64, TILE, LINE>), dim3(grid), dim3(256), 0, stream(), a);
    } else {
        fail(AE_ERR_INVALID_ARG, "internal: no node-line step kernel for rows of %d columns in lines of %d floats", DIM, LINE);
    }
}
template <int DIM, bool F64, bool TILE>
void launch_direct_line3(const DirectArgs& a, uint32_t line, int* blocks_per_cu) {
    if (line == 8) launch_direct_line4<DIM, 8, F64, TILE>(a, blocks_per_cu);
    else if (line == 16) launch_direct_line4<DIM, 16, F64, TILE>(a, blocks_per_cu);
    else launch_direct_line4<DIM, 32, F64, TILE>(a, blocks_per_cu);
}
// blocks_per_cu non-null: only the occupancy of the instantiation that `tile` / `f64` / `line` select is reported
template <int DIM>
void launch_direct_line(const DirectArgs& a, uint32_t line, bool f64, bool tile_wanted, int* blocks_per_cu) {
    const bool tile = tile_wanted && a.c.n > (uint64_t)TileShape<DIM>::kRows * 4ull;
    if (f64) { if (tile) launch_direct_line3<DIM, true, true>(a, line, blocks_per_cu); else launch_direct_line3<DIM, true, false>(a, line, blocks_per_cu); }
    else { if (tile) launch_direct_line3<DIM, false, true>(a, line, blocks_per_cu); else launch_direct_line3<DIM, false, false>(a, line, blocks_per_cu); }
}
#define AE_SL_LINE_LAUNCHERS(PREFIX, D) PREFIX template void launch_direct_line<D>(const DirectArgs&, uint32_t, bool, bool, int*);
#ifdef AE_SL_INSTANTIATE_LINE_DIM
AE_SL_LINE_LAUNCHERS(, AE_SL_INSTANTIATE_LINE_DIM)
#else
AE_SL_LINE_LAUNCHERS(extern, 2)
AE_SL_LINE_LAUNCHERS(extern, 3)
AE_SL_LINE_LAUNCHERS(extern, 4)
AE_SL_LINE_LAUNCHERS(extern, 8)
AE_SL_LINE_LAUNCHERS(extern, 16)
#endif

// The launchers are instantiated one row stride per translation unit (ce_slice_dim*.hip define AE_SL_INSTANTIATE_DIM; the node-line
// step kernels in units of their own, ce_slice_line_dim*.hip: AE_SL_INSTANTIATE_LINE_DIM); everybody else only sees the declarations.
#define AE_SL_LAUNCHERS(PREFIX, D)                                                                                                        \
    PREFIX template void launch_direct<D>(const DirectArgs&, uint32_t, bool);                                                             \
    PREFIX template void direct_blocks_per_cu<D>(uint32_t, bool, bool, int*);                                                             \
    PREFIX template void launch_slice<D>(const SliceRunArgs&, unsigned, uint32_t, bool, bool);                                            \
    PREFIX template void launch_exec<D>(const SliceArgs&, unsigned, uint32_t, bool);                                                      \
    PREFIX template void launch_chain_run<D>(const SliceArgs&, unsigned, uint32_t, bool, const uint32_t*, const uint32_t*);
#ifdef AE_SL_INSTANTIATE_DIM
AE_SL_LAUNCHERS(, AE_SL_INSTANTIATE_DIM)
#else
AE_SL_LAUNCHERS(extern, 2)
AE_SL_LAUNCHERS(extern, 3)
AE_SL_LAUNCHERS(extern, 4)
AE_SL_LAUNCHERS(extern, 8)
AE_SL_LAUNCHERS(extern, 16)
AE_SL_LAUNCHERS(extern, 32)
AE_SL_LAUNCHERS(extern, 64)
#endif

}  // namespace sl
}  // namespace ae
