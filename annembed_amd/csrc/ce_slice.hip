// ce_slice.hip -- AE_CE_SLICED: the CE gradient batch (gradient_iteration_threaded, src/embedder.rs:1311-1315) as a
// TIME-SLICED execution on CONFLICT-FREE MATCHINGS -- the faithful mode for graphs of any size (throughput-bound).
//
// What it keeps of the reference: every sample is applied to the CURRENT rows of both its end points with one
// gradient (embedder.rs:1228-1239), the samples come in an i.i.d. order, the reference's f64 scalars (f32 on request); the five
// negatives are read as the memory system has them (at most one launch old).  What it gives up: reproducibility sample by sample.
//
// How.  The i.i.d. edge draws of a batch are a Poisson process per edge (as in ce_event.hip): edge e fires c_e ~ Poisson(mu_e)
// times at i.i.d. uniform times.  The events of a batch are generated edge by edge (count -> scan -> fill), their times cut into
// thin SLICES (about half an event per node and slice).  Which events of a slice may run side by side is a property of the
// GRAPH, not of the draws: two samples conflict when their edges share a node.  So the edges are coloured ONCE per graph
// (slice_color_edges) so that every class is a forest of IN-STARS: no node is the source of two edges of a class, none is source of one
// and target of another -- but any number of edges of a class may share their target (k + 5 classes whatever the in-degrees; a proper
// colouring needs as many as the largest degree).  The events are bucketed by (slice, class) with one radix sort, classes in an order
// drawn afresh for every slice, and a STEP = the events of one class in one slice is one launch of sl_direct_kernel
// (ce_slice_kernels.h): every lane reads y_i and y_j, applies the attraction to both and the five repulsions to y_i exactly as
// embedder.rs:1207-1301 and writes both -- no ownership marks, no retries, no pending lists; the events of a step that share their
// target sit side by side in the array and run as a chain through the target's row, handed from lane to lane (the reference: the
// row's lock, one holder at a time).  Within a slice the order of two events that share a node is the order of their classes, i.e.
// uniformly random.
//
// Edges that find no colour (a fraction of a per cent) form the OVERFLOW class of every slice, executed optimistically after the
// slice's steps (the first form of this mode):
//   * every pending event marks its two rows in an owner array with its own id (plain stores: the last writer wins);
//   * the events that find their id on BOTH rows run, the others are deferred to the next pass (later passes: with probability
//     1/2 per pass, which breaks repeating stand-offs), what is left after the slice's passes joins the next slice, the batch
//     ends with passes until nothing is pending.
#include "ce_slice_kernels.h"

#include <rocprim/rocprim.hpp>

#include <chrono>
#include <random>

using namespace ae;
using namespace ae::sl;

namespace ae {
void sort_pairs_u32_u32(uint32_t* d_keys_in, uint32_t* d_keys_out, uint32_t* d_vals_in, uint32_t* d_vals_out, uint64_t count, unsigned end_bit);
}

namespace {

// ------------------------------------------------------------------------------------------------------------------
// edge colouring (once per graph): every class is a forest of IN-STARS
// ------------------------------------------------------------------------------------------------------------------
// Which events of a time slice may run side by side is a property of the graph.  A proper edge colouring (every class a matching:
// round 3) needs as many classes as the largest degree -- and in-degrees of kNN graphs reach the hundreds in 28-D, the thousands in
// 128-D --, and a class is a launch per slice.  The step kernel can do more than a matching: the events of a step that share their
// TARGET sit side by side in the event array and run as a chain through the target's row (sl_step_body), exactly as the row's lock
// serialises them in the reference.  So a class only has to be a forest of in-stars:
//     (1) the out-edges of a node have pairwise different colours        (a node is the SOURCE of at most one event of a step);
//     (2) no in-edge of a node has the colour of one of its out-edges     (a node is never source and target in the same step).
// Any number of in-edges of a node may share a colour.  A node of out-degree k then needs k colours for its out-edges and a few more
// for its in-edges, WHATEVER its in-degree: k + 5 classes colour the kNN graph of 11 M Higgs-shaped points (k = 6, in-degrees up
// to 132) with 0.5 % of the edge mass left over, where the proper colouring took 19 classes + 11 % (hub-hub edges).
//
// Parallel greedy, node-centric: in a round every node proposes colours for ALL its uncoloured out-edges at once (pairwise different,
// none of its own in- or out-colours, none of the target's out-colours; a uniformly drawn colour among those left, so that a hub's
// in-edges spread evenly over the classes), then commits the proposals that do not collide with what the TARGET proposes for its own
// out-edges in the same round (rule (2) between simultaneous commits: the in-edge yields).  A target whose free colours are down to
// what its own uncoloured out-edges need accepts only colours its in-edges already use.  Nodes whose in-degree would fill the
// in-palette go first (phase A: their out-edges pick freely, their in-edges then take what is left) -- measured on kNN graphs of
// blob data: 0.5 % left over at k + 6 classes against 2 % in one phase.  What is left after the rounds is the OVERFLOW class,
// executed optimistically after the slice's steps.
__device__ __forceinline__ uint32_t nth_set_bit(unsigned long long m, uint32_t nth) {
    for (uint32_t q = 0; q < nth; q++) m &= m - 1ull;
    return (uint32_t)__builtin_ctzll(m);
}
__device__ __forceinline__ void sl_row(const CeDev& c, uint64_t i, uint64_t& b, uint32_t& len) {
    if (c.uniform_k) { b = i * c.uniform_k; len = c.uniform_k; }
    else { b = c.indptr[i]; len = (uint32_t)(c.indptr[i + 1] - b); }
}
__global__ void __launch_bounds__(256) sl_in_degree_kernel(uint64_t nnz, const uint32_t* __restrict__ nbr, uint32_t* __restrict__ indeg) {
    const uint64_t e = blockIdx.x * 256ull + threadIdx.x;
    if (e < nnz) atomicAdd(&indeg[nbr[e]], 1u);
}
__global__ void __launch_bounds__(256) sl_max_u32_kernel(uint64_t n, const uint32_t* __restrict__ x, uint32_t* __restrict__ out) {
    uint32_t m = 0;
    for (uint64_t v = blockIdx.x * 256ull + threadIdx.x; v < n; v += (uint64_t)gridDim.x * 256ull) m = max(m, x[v]);
    for (int off = 32; off > 0; off >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, off));
    if ((threadIdx.x & 63) == 0) atomicMax(out, m);
}
// one thread per node: proposals for its uncoloured out-edges
__global__ void __launch_bounds__(256) sl_star_propose_kernel(CeDev c, const uint8_t* __restrict__ color, const unsigned long long* __restrict__ out_used,
                                                              const unsigned long long* __restrict__ in_used, const uint32_t* __restrict__ indeg,
                                                              uint32_t first_indeg, unsigned long long* __restrict__ prop_mask, uint8_t* __restrict__ prop,
                                                              uint32_t classes, uint32_t hkey) {
    const uint64_t i = blockIdx.x * 256ull + threadIdx.x;
    if (i >= c.n) return;
    uint64_t b;
    uint32_t len;
    sl_row(c, i, b, len);
    const unsigned long long full = classes >= 64 ? ~0ull : ((1ull << classes) - 1ull);
    const unsigned long long mine = out_used[i] | in_used[i];
    unsigned long long chosen = 0ull;
    if (indeg[i] >= first_indeg) {
        for (uint32_t m = 0; m < len; m++) {
            if (color[b + m] != kNoColor) continue;
            prop[b + m] = kNoColor;
            const uint32_t j = c.nbr[b + m];
            const unsigned long long ou_j = out_used[j], iu_j = in_used[j];
            unsigned long long av = full & ~(mine | chosen | ou_j);
            uint64_t bj;
            uint32_t lenj;
            sl_row(c, j, bj, lenj);
            const int need_j = (int)lenj - __popcll(ou_j), free_j = (int)classes - __popcll(ou_j | iu_j);
            if (free_j <= need_j) av &= iu_j;   // j's last free colours are kept for its own out-edges
            // (Preferring a colour none of j's in-edges has yet -- fewer chains -- was measured: the in-palettes fill up in the first round
            // and 10 % of a lattice's edge mass is left without a colour, against 2.5 %.)
            if (!av) continue;
            const uint32_t pick = nth_set_bit(av, __umulhi(pcg_hash((uint32_t)(b + m) ^ hkey), (uint32_t)__popcll(av)));
            prop[b + m] = (uint8_t)pick;
            chosen |= 1ull << pick;
        }
    }
    prop_mask[i] = chosen;
}
// commits the proposals that the target does not propose for itself this round
__global__ void __launch_bounds__(256) sl_star_commit_kernel(CeDev c, uint8_t* __restrict__ color, unsigned long long* __restrict__ out_used,
                                                             unsigned long long* __restrict__ in_used, const unsigned long long* __restrict__ prop_mask,
                                                             const uint8_t* __restrict__ prop, uint32_t classes, unsigned long long* __restrict__ remaining) {
    const uint64_t i = blockIdx.x * 256ull + threadIdx.x;
    uint32_t left = 0;
    if (i < c.n) {
        uint64_t b;
        uint32_t len;
        sl_row(c, i, b, len);
        unsigned long long add = 0ull;
        const unsigned long long pm = prop_mask[i];
        for (uint32_t m = 0; m < len; m++) {
            if (color[b + m] != kNoColor) continue;
            const uint8_t p = pm ? prop[b + m] : kNoColor;
            if (p == kNoColor) { left++; continue; }
            const uint32_t j = c.nbr[b + m];
            if ((prop_mask[j] >> p) & 1ull) { left++; continue; }
            {   // the target's palette as it is NOW (the proposal saw it a kernel ago; its in-edges commit side by side): a NEW in-colour
                // only while the target keeps enough free colours for its own uncoloured out-edges
                const unsigned long long iu_j = __hip_atomic_load(&in_used[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), ou_j = out_used[j];
                if (!((iu_j >> p) & 1ull)) {
                    uint64_t bj;
                    uint32_t lenj;
                    sl_row(c, j, bj, lenj);
                    if ((int)classes - __popcll(iu_j | ou_j) <= (int)lenj - __popcll(ou_j)) { left++; continue; }
                }
            }
            color[b + m] = p;
            add |= 1ull << p;
            atomicOr(&in_used[j], 1ull << p);
        }
        if (add) out_used[i] |= add;   // (only this thread writes out_used[i])
    }
    for (int off = 32; off > 0; off >>= 1) left += __shfl_xor(left, off);
    if ((threadIdx.x & 63) == 0 && left) atomicAdd(&remaining[(blockIdx.x * 4u + (threadIdx.x >> 6)) & 1023u], (unsigned long long)left);
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
// per node: the probability mass of its overflow edges (sizes the pending lists); per edge: the key that puts the edges of a class
// that share a target side by side in the event-generation order (below 2^31: overflow edges, shuffled; see `finish`).  A sharded node range
// [lo, hi) (multi-GPU): the edges this shard generates events for are those whose SOURCE it owns and -- as half events, flagged -- those
// whose target it owns while the source is another shard's; every other edge gets the key kDropKey (sorted to the end and cut off).
// mass: [0] the edges this shard generates, [1] its cross-shard edges (one end here, one elsewhere).
constexpr uint32_t kDropKey = 0xFFFFFFFFu;
constexpr uint32_t kClsBits = 7;   // step keys of a batch with one launch per class: (slice << kClsBits) | (0: the overflow class, 1 + c: class c <= 63)
// probability mass that arrives at every node (its out-edges carry 1): with it, the rate of the events that touch a node
__global__ void __launch_bounds__(256) sl_in_mass_kernel(uint64_t nnz, const EdgeRec* __restrict__ erec, float* __restrict__ in_mass) {
    const uint64_t e = blockIdx.x * 256ull + threadIdx.x;
    if (e < nnz) atomicAdd(&in_mass[erec[e].j], erec[e].w);
}
constexpr uint32_t kEndMassMask = 0xFFFFu;   // EdgeRec.flags, low bits: mass of both end points (in + out) in 1/256, saturating
// a bijection of the node ids (28 bits: kNodeMask has 27): targets in an order that has nothing to do with their labels
__device__ __forceinline__ uint32_t mix_node(uint32_t v) {
    constexpr uint32_t kM = (1u << 28) - 1u;
    v = (v * 0x9E3779B1u) & kM;
    v ^= v >> 15;
    v = (v * 0x85EBCA6Bu) & kM;
    v ^= v >> 13;
    return v;
}
__global__ void __launch_bounds__(256) sl_color_finish_kernel(uint64_t nnz, EdgeRec* __restrict__ erec, const uint8_t* __restrict__ color,
                                                              float* __restrict__ node_ov, const float* __restrict__ in_mass,
                                                              uint32_t* __restrict__ group_key, uint32_t* __restrict__ ident,
                                                              int by_source, int label_order, uint64_t lo, uint64_t hi, double* __restrict__ mass) {
    const uint64_t e = blockIdx.x * 256ull + threadIdx.x;
    double m_gen = 0., m_cross = 0.;
    if (e < nnz) {
        EdgeRec r = erec[e];
        const uint32_t src = r.im >> 5;
        const bool s_in = src >= lo && src < hi, t_in = r.j >= lo && r.j < hi;
        uint32_t key = kDropKey;
        if (s_in || t_in) {
            const float ends = 2.f + in_mass[src] + in_mass[r.j];
            r.flags = (s_in ? 0u : kHalfEvent) | min((uint32_t)(ends * 256.f + 0.5f), kEndMassMask);
            erec[e].flags = r.flags;
            m_gen = (double)r.w;
            if (!(s_in && t_in)) m_cross = (double)r.w;
            if (color[e] == kOverflowColor) {
                atomicAdd(&node_ov[src], r.w);
                atomicAdd(&node_ov[r.j], r.w);
                key = label_order ? 0u : pcg_hash((uint32_t)e) >> 1;
            } else {
                const uint32_t g = by_source ? src : r.j;   // (by_source: a timing experiment only -- chains would be torn apart)
                key = label_order ? g + 1u : 0x80000000u | mix_node(g);
            }
        }
        group_key[e] = key;
        ident[e] = (uint32_t)e;
    }
    for (int off = 32; off > 0; off >>= 1) { m_gen += __shfl_xor(m_gen, off); m_cross += __shfl_xor(m_cross, off); }
    if ((threadIdx.x & 63) == 0) {
        if (m_gen != 0.) atomicAdd(&mass[0], m_gen);
        if (m_cross != 0.) atomicAdd(&mass[1], m_cross);
    }
}
// the class of the edge at every position of the target-grouped order: 0 overflow, 1 + class, 255 an edge this shard does not generate
__global__ void __launch_bounds__(256) sl_class_key_kernel(uint64_t nnz, const uint32_t* __restrict__ perm, const uint32_t* __restrict__ sorted_group_key,
                                                           const uint8_t* __restrict__ color, uint32_t* __restrict__ ckey) {
    const uint64_t x = blockIdx.x * 256ull + threadIdx.x;
    if (x >= nnz) return;
    const uint8_t c = color[perm[x]];
    ckey[x] = sorted_group_key[x] == kDropKey ? 255u : (c == kOverflowColor || c == kNoColor ? 0u : 1u + (uint32_t)c);
}
__global__ void __launch_bounds__(256) sl_count_below_kernel(uint64_t nnz, const uint32_t* __restrict__ keys, uint32_t bound, unsigned long long* __restrict__ out) {
    unsigned long long c = 0;
    for (uint64_t e = blockIdx.x * 256ull + threadIdx.x; e < nnz; e += (uint64_t)gridDim.x * 256ull) c += keys[e] < bound ? 1ull : 0ull;
    for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, c);
}
// the edges in event-generation order: out[x] = in[perm[x]]
// (relabel: the internal numbering of the nodes, or null)
__global__ void __launch_bounds__(256) sl_permute_edges_kernel(uint64_t nnz, const uint32_t* __restrict__ perm, const EdgeRec* __restrict__ erec,
                                                               const uint8_t* __restrict__ color, const uint32_t* __restrict__ relabel,
                                                               EdgeRec* __restrict__ erec_out, uint8_t* __restrict__ color_out) {
    const uint64_t x = blockIdx.x * 256ull + threadIdx.x;
    if (x >= nnz) return;
    const uint32_t e = perm[x];
    EdgeRec r = erec[e];
    if (relabel) {
        r.j = relabel[r.j];
        r.im = (relabel[r.im >> 5] << 5) | (r.im & 31u);
    }
    erec_out[x] = r;
    color_out[x] = color[e];
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
// (e: position in the event-generation order of the edges, slice_color_edges)
__global__ void __launch_bounds__(256) sl_count_kernel(CeDev c, uint64_t n_gen, const EdgeRec* __restrict__ erec, float unit, uint32_t key, uint32_t* __restrict__ cnt) {
    const uint64_t e = blockIdx.x * 256ull + threadIdx.x;
    if (e >= n_gen) return;
    const uint32_t ck = round_hash_key(key, c.seed) ^ kTagSlCount;
    const float mu = unit * erec[e].w;
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
// slice's class order (class_pos[slice][class]; overflow last).  The slices of an edge's events are i.i.d. uniform.  REPEATS of an
// edge inside a slice are where a slice differs from the stretch of the i.i.d. sequence it stands for: they land in the same step and
// run back to back, with no other event of their end points in between -- in the sequence that happens with probability p_stay (below)
// only.  Rounds 3-4 moved every repeat to the next slice (`spread` 1: never back to back); leaving them (`spread` 0) errs the other
// way, and both show: 1 M Higgs-shaped points, 8 columns, against the exact mode: lambda 1/8: CE 0.992 / 0.990 (moved / left),
// 1/4: 0.995 / 0.991, 1/2: 1.008 / 0.967, 1: 1.051 / 0.949 -- clipped attractions are violent (a pair moves to 2 % of its distance),
// so how two of them on one edge are spaced matters although < 3 % of the events are repeats.  `spread` 2 (the default): a repeat
// stays with probability p_stay and moves on otherwise.
// (Round 6: the same kernel with its writes through a wave-private LDS stage -- every thread drops its keys at its run's place, the wave
// then stores the stretch 64 places at a time, the records fetched from their lanes by shuffles: fully coalesced stores -- was built and
// measured: 134.2 against 134.2 ms per configs[3] batch.  The kernel is not bound by how its stores coalesce; commit d82df92 holds the code.)
// (Round 6, too: the events BUCKETED AT GENERATION -- the edges come in class order, so a virtual block of 512 edges of one class counts its
// events per slice, a scan over [slice][block] gives every block's place in every slice and the step pointers, and a second pass draws the
// events again, ranks them by slice inside the block (stably, in LDS: eight ballots per 64 events) and stores each slice's run in one
// piece: no keys, no radix pass, 12 bytes an event written once.  Word for word the events and step pointers of this path (checked on the
// device), and SLOWER: count pass 3.7 ms + fill pass 14.3 ms against 16.5 ms for count + scan + fill + the radix pass; 135.9 against 133.3 ms
// per configs[3] batch.  Drawing an edge's slices (the sort and the repeat rule below) is a third of this kernel's time and is paid twice
// there, and the ranking runs at two workgroups per CU under 75 KB of LDS.  Commit f2c9fab holds the code; profiles/r06/r6_c4_buckets_ab.jsonl,
// r6_evgen_buckets_kernel_stats_top.txt.)
// (And GENERATION AHEAD: a second set of event buffers, the next segment's events -- after a batch's last segment the next batch's first,
// speculatively: the same call with iter + 1 -- counted, filled and sorted on a side stream while the current segment's steps run.  The
// generator then takes 20 ms beside the steps instead of 17 alone, and the steps take that much longer: 128.7 -> 127.2 ms per configs[3]
// batch, 32.6 -> 31.9 on configs[2]'s large graph, at any stream priority -- for twice the event buffers (21 GB more at configs[3]).  Not
// kept; commit d266210 holds the code, profiles/r06/r6_c4_ahead_ab.jsonl the measurement.)
__global__ void __launch_bounds__(256) sl_fill_kernel(CeDev c, uint64_t n_gen, uint32_t key, const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ offs,
                                                      uint32_t n_slices, const EdgeRec* __restrict__ erec, const uint8_t* __restrict__ color,
                                                      const uint8_t* __restrict__ class_pos, uint32_t classes, int spread, float ev_per_mass,
                                                      uint32_t ov_every, uint32_t* __restrict__ keys, Event* __restrict__ vals) {
    const uint64_t e = blockIdx.x * 256ull + threadIdx.x;
    if (e >= n_gen) return;
    const uint32_t k = cnt[e], o = offs[e];
    if (!k) return;
    const uint32_t tk = pcg_hash(round_hash_key(key, c.seed) ^ kTagSlTime);
    const uint8_t cl = color[e];
    const EdgeRec er = erec[e];
    const Event evv{er.im, er.j | (er.flags & kHalfEvent), er.w};
    constexpr uint32_t kSortMax = 24;
    __shared__ uint32_t s_sl[kSortMax * 256];  // [r][thread]: the slices of this thread's edge (dynamic indexing: LDS, not scratch)
    uint32_t* sl = s_sl + threadIdx.x;
#define SL(r) sl[(r) * 256u]
    bool sorted = (spread & 3) && k > 1 && k <= kSortMax && k <= n_slices;
    if (sorted && n_slices <= 256u && !(spread & 4)) {
        // Only an edge with two events in ONE slice needs its slices in order (the repeat rule): a 256-bit set of the slices drawn says
        // whether there is one (k = 10 events in 240 slices: one edge in five).  Without a repeat the slices go out as drawn -- the events
        // of an edge are equal and the sort by slice is stable: the sorted array is the same.  (spread bit 4: always in order, for the A/B)
        unsigned long long m0 = 0ull, m1 = 0ull, m2 = 0ull, m3 = 0ull;
        bool dup = false;
        for (uint32_t r = 0; r < k; r++) {
            const uint32_t sd = __umulhi(pcg_hash((pcg_hash((uint32_t)e) + r * 0x9E3779B9u) ^ tk), n_slices), ws = sd >> 6;
            const unsigned long long bit = 1ull << (sd & 63u);
            const unsigned long long cur = ws == 0u ? m0 : (ws == 1u ? m1 : (ws == 2u ? m2 : m3));
            dup = dup || (cur & bit) != 0ull;
            m0 |= ws == 0u ? bit : 0ull; m1 |= ws == 1u ? bit : 0ull; m2 |= ws == 2u ? bit : 0ull; m3 |= ws == 3u ? bit : 0ull;
        }
        sorted = dup;
    }
    spread &= 3;
    if (sorted) {  // the k slices in ascending order (insertion sort); then the repeats inside a slice are dealt with
        for (uint32_t r = 0; r < k; r++) {
            const uint32_t s = __umulhi(pcg_hash((pcg_hash((uint32_t)e) + r * 0x9E3779B9u) ^ tk), n_slices);
            uint32_t q = r;
            while (q > 0 && SL(q - 1) > s) { SL(q) = SL(q - 1); q--; }
            SL(q) = s;
        }
        // p_stay: the probability that NO other event of the edge's end points falls between two events of the edge that share a slice,
        // in the i.i.d. sequence the slices stand for: the two at uniform places of the slice (distance x: density 2 (1 - x)), the
        // others a Poisson stream of rate rho per slice: 2 (rho - 1 + exp(-rho)) / rho^2
        const float rho = (float)(er.flags & kEndMassMask) * (1.0f / 256.0f) * ev_per_mass;
        const float p_stay = spread == 1 ? 0.f : (rho < 0.05f ? 1.f - rho * (1.0f / 3.0f) : 2.f * (rho - 1.f + __expf(-rho)) / (rho * rho));
        for (uint32_t r = 1; r < k; r++) {
            if (SL(r) > SL(r - 1)) continue;
            const float u = (float)(pcg_hash((pcg_hash((uint32_t)e) ^ 0x5851F42Du) + r * 0x9E3779B9u + tk) >> 8) * (1.0f / 16777216.0f);
            SL(r) = SL(r - 1) + (u < p_stay ? 0u : 1u);
        }
        // what ran past the end of the segment is pulled back from the top
        if (SL(k - 1) >= n_slices) {
            SL(k - 1) = n_slices - 1u;
            for (uint32_t r = k - 1; r > 0 && SL(r - 1) > SL(r); r--) SL(r - 1) = SL(r);
        }
    }
    for (uint32_t r = 0; r < k; r++) {
        uint32_t s = sorted ? SL(r) : __umulhi(pcg_hash((pcg_hash((uint32_t)e) + r * 0x9E3779B9u) ^ tk), n_slices);
        if (cl >= classes && ov_every > 1u) s = min(n_slices - 1u, s - s % ov_every + ov_every / 2u);   // (a thin overflow class runs in every ov_every-th slice: ce_slice_gradient_iteration)
        // class_pos null: one launch per class -- the key is (slice << kClsBits) | class, sorted on its slice bits alone (the edges come in
        // class order); else merged slices: slice * (classes + 1) + the class's POSITION in the slice's order
        // (the overflow class is 0 there, class c is 1 + c: the ORDER the edges are generated in -- the sort on the slice bits is stable, the
        // class bits must already ascend inside a slice)
        const uint32_t cls = cl < classes ? 1u + (uint32_t)cl : 0u;
        keys[o + r] = class_pos ? s * (classes + 1u) + (cl < classes ? (uint32_t)class_pos[s * classes + cl] : classes) : (s << kClsBits) | cls;
        vals[o + r] = evv;
    }
#undef SL
}
// (hubness weighting) the batch's pool of i.i.d. draws of the NodeSampler (embedder.rs:927-930), ce_slice_kernels.h: TileFetch
__global__ void __launch_bounds__(256) sl_hub_pool_kernel(CeDev c, uint32_t key, uint32_t count, uint32_t* __restrict__ pool) {
    const uint32_t x = blockIdx.x * 256u + threadIdx.x;
    if (x >= count) return;
    const uint32_t w0 = pcg_hash(pcg_hash(round_hash_key(key, c.seed) ^ kTagSlPool) + x * 0x9E3779B9u);
    const uint32_t xs = __umulhi(w0, (uint32_t)c.n);
    const float uu = (float)(pcg_hash(w0 ^ 0x9E3779B9u) >> 8) * (1.0f / 16777216.0f);
    const uint2 he = c.hub_tab[xs];
    pool[x] = (uu < __uint_as_float(he.x)) ? xs : he.y;
}
// (AE_SL_CHECK_FILL) words that differ between two arrays
__global__ void __launch_bounds__(256) sl_diff_words_kernel(const uint32_t* __restrict__ x, const uint32_t* __restrict__ y, uint64_t words, unsigned long long* __restrict__ out) {
    unsigned long long c = 0;
    for (uint64_t q = blockIdx.x * 256ull + threadIdx.x; q < words; q += (uint64_t)gridDim.x * 256ull) c += x[q] != y[q] ? 1ull : 0ull;
    for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, c);
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
    for (uint64_t e = b; e < e1; e++) out[e] = EdgeRec{c.nbr[e], c.proba[e], 0u, ((uint32_t)i << 5) | (uint32_t)(e - b)};
}
// static record of a node: SREC floats = {embedded scale, KP neighbour ids (padded with ~0), KP edge probabilities}, KP = (SREC - 1) / 2
// (perm: the internal numbering, ae_entropy_optim::sl_perm, or null: the record of node i sits in row perm[i] and names its neighbours
// by their internal numbers)
__global__ void sl_static_rec_kernel(CeDev c, uint32_t srec, const uint32_t* __restrict__ perm, float* __restrict__ out) {
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i >= c.n) return;
    uint64_t b;
    uint32_t k;
    if (c.uniform_k) { b = i * c.uniform_k; k = c.uniform_k; }
    else { b = c.indptr[i]; k = (uint32_t)(c.indptr[i + 1] - b); }
    const uint32_t kp = (srec - 1u) / 2u;
    float* r = out + (uint64_t)(perm ? perm[i] : (uint32_t)i) * srec;
    r[0] = c.emb_scale[i];
    for (uint32_t m = 0; m < kp; m++) {
        r[1 + m] = __uint_as_float(m < k ? (perm ? perm[c.nbr[b + m]] : c.nbr[b + m]) : 0xFFFFFFFFu);
        r[1 + kp + m] = m < k ? c.proba[b + m] : 0.f;
    }
    for (uint32_t m = 1 + 2 * kp; m < srec; m++) r[m] = 0.f;
}

// node lines (ce_slice_kernels.h: LineRec): what is static about node i as a source -- embedded scale, neighbour ids in internal numbers,
// padded with ~0 -- behind its row in the batch's internal copy of the coordinates (row perm[i], `line` floats per node)
__global__ void sl_line_static_kernel(CeDev c, uint32_t line, const uint32_t* __restrict__ perm, float* __restrict__ lines) {
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i >= c.n) return;
    uint64_t b;
    uint32_t k;
    if (c.uniform_k) { b = i * c.uniform_k; k = c.uniform_k; }
    else { b = c.indptr[i]; k = (uint32_t)(c.indptr[i + 1] - b); }
    float* r = lines + (uint64_t)(perm ? perm[i] : (uint32_t)i) * line;
    r[c.dim] = c.emb_scale[i];
    for (uint32_t m = 0; c.dim + 1u + m < line; m++) r[c.dim + 1u + m] = __uint_as_float(m < k ? (perm ? perm[c.nbr[b + m]] : c.nbr[b + m]) : 0xFFFFFFFFu);
}

// internal numbering: keys for the random order, its inverse, and the row moves of a batch's start and end
__global__ void __launch_bounds__(256) sl_perm_keys_kernel(uint64_t n, uint32_t seed, uint32_t* __restrict__ keys, uint32_t* __restrict__ ident) {
    const uint64_t v = blockIdx.x * 256ull + threadIdx.x;
    if (v < n) { keys[v] = pcg_hash((uint32_t)v ^ seed); ident[v] = (uint32_t)v; }
}
// the rank whose range holds node order[x] (ranges: lo_0, hi_0, lo_1, ... tiling [0, n) in order)
__global__ void __launch_bounds__(256) sl_perm_block_kernel(uint64_t n, const uint32_t* __restrict__ order, const uint64_t* __restrict__ ranges, uint32_t world,
                                                            uint32_t* __restrict__ block) {
    const uint64_t x = blockIdx.x * 256ull + threadIdx.x;
    if (x >= n) return;
    const uint64_t v = order[x];
    uint32_t q = 0;
    while (q + 1u < world && v >= ranges[2u * q + 1u]) q++;
    block[x] = q;
}
__global__ void __launch_bounds__(256) sl_perm_invert_kernel(uint64_t n, const uint32_t* __restrict__ order, uint32_t* __restrict__ perm) {
    const uint64_t x = blockIdx.x * 256ull + threadIdx.x;
    if (x < n) perm[order[x]] = (uint32_t)x;
}
// rows of `dim` floats: to_internal: dst[perm[v]] = src[v]; else dst[v] = src[perm[v]] (perm null: the identity).  The internal copy's
// rows are `stride` floats apart (dim, or more: a node's dependency words sit around its row, ce_slice_kernels.h -- the copy is zeroed first);
// [v_lo, v_hi) \ [skip_lo, skip_hi): the rows moved (internal numbers: an exchange moves a rank's own rows out and the others' in)
__global__ void __launch_bounds__(256) sl_move_rows_kernel(uint64_t v_lo, uint64_t v_hi, uint64_t skip_lo, uint64_t skip_hi, uint32_t dim, uint32_t stride,
                                                           const uint32_t* __restrict__ perm, const float* __restrict__ src, float* __restrict__ dst, int to_internal) {
    for (uint64_t t = blockIdx.x * 256ull + threadIdx.x; t < (v_hi - v_lo) * dim; t += (uint64_t)gridDim.x * 256ull) {   // (n x dim may pass 2^32)
        const uint64_t v = v_lo + t / dim, q = t % dim;
        if (v >= skip_lo && v < skip_hi) continue;
        const uint64_t p = perm ? perm[v] : v;
        if (to_internal) dst[p * stride + q] = src[v * dim + q];
        else dst[v * dim + q] = src[p * stride + q];
    }
}
__global__ void __launch_bounds__(256) sl_hub_tab_kernel(uint64_t n, const uint32_t* __restrict__ perm, const uint2* __restrict__ tab, uint2* __restrict__ out) {
    const uint64_t v = blockIdx.x * 256ull + threadIdx.x;
    if (v < n) out[perm[v]] = make_uint2(tab[v].x, perm[tab[v].y]);
}

// ------------------------------------------------------------------------------------------------------------------
// merged slices (sl_slice_kernel, ce_slice_kernels.h)
// ------------------------------------------------------------------------------------------------------------------
// before the launch: every event of the slice enters its class in the words of its two nodes (fire-and-forget atomics).  Only for a slice
// whose predecessor could not do it (the first of a segment, one after an empty slice): sl_slice_kernel prepares the next slice itself.
__global__ void __launch_bounds__(256) sl_dep_mark_kernel(SliceRunArgs a) {
    __shared__ uint32_t s_ptr[kDepBits + 1];
    if (threadIdx.x <= a.classes) s_ptr[threadIdx.x] = a.sptr[threadIdx.x];
    __syncthreads();
    const uint32_t p = s_ptr[0] + blockIdx.x * 256u + threadIdx.x;
    if (p >= s_ptr[a.classes]) return;
    uint32_t q = 0;
    while (q + 1u < a.classes && p >= s_ptr[q + 1u]) q++;
    dep_mark_event(a.dep, a.dep_stride, a.d.ev[p], q);
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
            a.owner[(uint64_t)a.owner_mark * a.c.n + ev_node(p.j)] = p.idx;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        a.counts[a.dst_list * kSub + sub] = total < a.cap ? total : (uint32_t)a.cap;  // (cap is sized so that this never truncates; flagged otherwise)
        if (total > a.cap) atomicOr(reinterpret_cast<unsigned int*>(a.done_counter + 1024), 1u);
        a.counts[a.zero_list * kSub + sub] = 0;
    }
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
        if (pcg_hash(ev_node(p.j) ^ coin) & 1u) {
            next[pos] = atomicExch(&head[ev_node(p.j)], pos);
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
    for (uint64_t t = blockIdx.x * 256ull + threadIdx.x; t < total; t += (uint64_t)gridDim.x * 256ull) head[ev_node(a.lists[so + t].j)] = kNil;
}

// stable (LSD radix) sort of (step key, event) pairs on the key bits [0, end_bit)
// Sorts the events by their (slice, class) key between the two buffer pairs; returns true if the result sits in the second pair.
// rocPRIM's double-buffer form: the caller's two pairs ARE the ping-pong buffers, the temporary storage is histograms only.  (The
// in/out form asks for a full copy of keys and values as temporary storage -- 7.6 GB at the C4 shape -- and its size follows the
// batch's event count: whenever a batch set a new record the stream-ordered pool had to get a fresh block from the driver, 1.5-2 s,
// a few times per run.  Found in round 3 as C4-shape batches of 300-900 ms among batches of 121 ms.)
bool sort_events(ae_entropy_optim* o, uint32_t* keys_a, uint32_t* keys_b, Event* vals_a, Event* vals_b, uint64_t count, unsigned begin_bit, unsigned end_bit) {
    static_assert(sizeof(Event) == 12, "events are sorted as 12-byte values");
    rocprim::double_buffer<uint32_t> dk(keys_a, keys_b);
    rocprim::double_buffer<Event> dv(vals_a, vals_b);
    size_t tmp_bytes = 0;
    if (rocprim::radix_sort_pairs(nullptr, tmp_bytes, dk, dv, count, begin_bit, end_bit, stream()) != hipSuccess)
        fail(AE_ERR_NO_DEVICE, "rocprim radix_sort_pairs (size query) failed");
    if (o->sl_sort_tmp.n < tmp_bytes + 1) o->sl_sort_tmp.alloc(2 * tmp_bytes + 4096);  // (kept with the handle: no allocation in the batch)
    if (rocprim::radix_sort_pairs(o->sl_sort_tmp.p, tmp_bytes, dk, dv, count, begin_bit, end_bit, stream()) != hipSuccess)
        fail(AE_ERR_NO_DEVICE, "rocprim radix_sort_pairs failed");
    return dk.current() == keys_b;
}

}  // namespace

namespace ae {

constexpr double kMaxCrossShardMass = 0.10;
double ce_slice_max_cross_mass() { return kMaxCrossShardMass; }

const char* ce_slice_unsupported(const ae_entropy_optim* o) {
    if (o->dev.nnz >= 0xFFFFFFFFull) return "more than 2^32 edges";
    if (o->dev.n > (1ull << 27)) return "more than 2^27 nodes";
    if (o->g->max_nbng > 32) return "rows of more than 32 neighbours";
    return nullptr;
}

// the edge colouring of the graph (see the kernels above): sl_color[e] = class of edge e or kOverflowColor, and the event-generation
// order of the edges (the edges of a class that share a target side by side)
static double sl_resident_events(ae_entropy_optim* o);
// Slices run merged (sl_slice_kernel) while a class step holds less than this share of what the device holds at once.  Measured on a
// rank's share of an 11 M-node graph (resident step 190 k events): 31 k events per step 59.8 -> 33.2 ms per batch, 62 k 68.5 -> 57.2,
// 125 k 85.8 -> 112 (full steps are bound by requests, and the dependency words add four per event to the step's six).
constexpr double kMergeBelow = 0.45;
static void slice_color_edges(ae_entropy_optim* o) {
    const uint64_t n = o->dev.n, nnz = o->dev.nnz;
    EdgeRec* erec = reinterpret_cast<EdgeRec*>(o->sl_erec.p);
    o->sl_color.alloc(nnz);
    AE_HIP(hipMemsetAsync(o->sl_color.p, kNoColor, nnz, stream()));
    o->sl_node_ov.alloc(n);
    o->sl_node_ov.zero();
    o->sl_classes = 0;
    o->sl_ov_frac = 1.0;
    o->sl_max_in_degree = 0;
    o->sl_gen_edges = nnz;
    o->sl_gen_mass = (double)n;
    o->sl_cross_frac = 0.;
    o->sl_erec_gen.release();
    o->sl_color_gen.release();
    const unsigned grid = blocks_for(nnz, 256), ngrid = blocks_for(n, 256);
    DevBuf<uint32_t> group_key, ident;
    group_key.alloc_pooled(nnz); ident.alloc_pooled(nnz);
    // Tail of both paths: overflow masses per node, the shard's edges (multi-GPU: node range [node_lo, node_hi)) with their half-event
    // flags, and the event-generation order -- the overflow edges in a random order, then the class edges grouped by target (stable
    // sort: the events of a step that share a target end up side by side), the targets in the order of a HASH of their labels
    // (mix_node); the edges of other shards are cut off.  Why not the order of the labels: a workgroup runs 256 consecutive events and
    // shares a tile of negatives among them (ce_slice_kernels.h: TileShape).  With labels that carry locality (a graph stored component
    // by component) those 256 samples then sit in one or two clusters and all meet the same few tile windows in a launch: measured CE
    // 1.06 of the exact mode's, lower quartile of the edge lengths 0.70 on 1 M Higgs-shaped points in component order, against
    // 1.015 / 0.92 with gathered negatives -- the result depended on how the caller had numbered the nodes.  Hashed, a workgroup's
    // samples are unrelated whatever the labels say (same graph: 1.016 / 0.88); the batch time did not move (the targets of a step
    // were ~44 rows apart in label order too: no line was ever shared).
    const bool label_order = debug_knob("AE_SL_LABEL_ORDER") != nullptr;   // A/B: targets in the order of their labels (rounds 3-4)
    auto finish = [&] {
        DevBuf<double> gm;
        gm.alloc_pooled(2);
        gm.zero();
        DevBuf<float> in_mass;
        in_mass.alloc_pooled(n);
        in_mass.zero();
        hipLaunchKernelGGL(sl_in_mass_kernel, dim3(grid), dim3(256), 0, stream(), nnz, (const EdgeRec*)erec, in_mass.p);
        hipLaunchKernelGGL(sl_color_finish_kernel, dim3(grid), dim3(256), 0, stream(), nnz, erec, (const uint8_t*)o->sl_color.p, o->sl_node_ov.p,
                           (const float*)in_mass.p, group_key.p, ident.p, debug_knob("AE_SL_SORT_SRC") ? 1 : 0, label_order ? 1 : 0, o->dev.node_lo, o->dev.node_hi, gm.p);
        check_launch("sl_color_finish");
        const std::vector<double> hg = gm.to_host();
        {   // the busiest row of the overflow class (its events run one per pass)
            DevBuf<uint32_t> mx;
            mx.alloc_pooled(1);
            mx.zero();
            hipLaunchKernelGGL(sl_max_u32_kernel, dim3(grid_cap(n, 256, 1024)), dim3(256), 0, stream(), n, reinterpret_cast<const uint32_t*>(o->sl_node_ov.p), mx.p);
            uint32_t bits = 0;   // (non-negative floats order like their bit patterns)
            mx.download(&bits, 1);
            memcpy(&o->sl_node_ov_max, &bits, 4);
        }
        o->sl_gen_mass = hg[0];
        o->sl_cross_frac = hg[0] > 0. ? hg[1] / hg[0] : 0.;
        const bool sharded = o->dev.node_lo != 0 || o->dev.node_hi != n;
        if (!o->sl_classes && !sharded && label_order) { sync(); return; }   // (everything optimistic on one device: graph order as it is)
        DevBuf<uint32_t> key_out, perm;
        key_out.alloc_pooled(nnz); perm.alloc_pooled(nnz);
        sort_pairs_u32_u32(group_key.p, key_out.p, ident.p, perm.p, nnz, 32);
        // ... then CLASS-MAJOR (round 6; a stable pass on 8 bits: inside a class the targets stay in hashed order): the events of a time slice
        // then leave the generator already in class order, and a batch with one launch per class sorts its events by SLICE alone -- one
        // radix pass instead of two (ce_slice_gradient_iteration)
        {
            DevBuf<uint32_t> ckey, perm_in;
            ckey.alloc_pooled(nnz); perm_in.alloc_pooled(nnz);
            hipLaunchKernelGGL(sl_class_key_kernel, dim3(grid), dim3(256), 0, stream(), nnz, (const uint32_t*)perm.p, (const uint32_t*)key_out.p,
                               (const uint8_t*)o->sl_color.p, ckey.p);
            AE_HIP(hipMemcpyAsync(perm_in.p, perm.p, sizeof(uint32_t) * nnz, hipMemcpyDeviceToDevice, stream()));
            sort_pairs_u32_u32(ckey.p, key_out.p, perm_in.p, perm.p, nnz, 8);   // (key_out: the sorted class keys from here on)
            sync();
        }
        // edges of this shard = class keys below 255 (sorted: a binary search on the device would do; the count comes with the masses)
        DevBuf<unsigned long long> cnt;
        cnt.alloc_pooled(1);
        cnt.zero();
        hipLaunchKernelGGL(sl_count_below_kernel, dim3(grid_cap(nnz, 256, 2048)), dim3(256), 0, stream(), nnz, (const uint32_t*)key_out.p, 255u, cnt.p);
        unsigned long long n_gen = 0;
        cnt.download(&n_gen, 1);
        o->sl_gen_edges = n_gen;
        o->sl_erec_gen.alloc(std::max<uint64_t>(1, n_gen) * 4);
        o->sl_color_gen.alloc(std::max<uint64_t>(1, n_gen));
        if (n_gen)
            hipLaunchKernelGGL(sl_permute_edges_kernel, dim3(blocks_for(n_gen, 256)), dim3(256), 0, stream(), (uint64_t)n_gen, (const uint32_t*)perm.p,
                               (const EdgeRec*)erec, (const uint8_t*)o->sl_color.p, (const uint32_t*)(o->sl_perm.n ? o->sl_perm.p : nullptr),
                               reinterpret_cast<EdgeRec*>(o->sl_erec_gen.p), o->sl_color_gen.p);
        check_launch("sl_permute_edges");
        sync();
        o->sl_erec.release();   // (the generation order is what the batches read)
        o->sl_color.release();
    };
    auto all_optimistic = [&] {
        hipLaunchKernelGGL(sl_color_giveup_kernel, dim3(grid), dim3(256), 0, stream(), nnz, o->sl_color.p);
        check_launch("sl_color");
        finish();
    };
    // in-degrees (the reference's hubness counts, hubness.rs:39-76)
    DevBuf<uint32_t> indeg, dmax;
    indeg.alloc_pooled(n); dmax.alloc_pooled(1);
    indeg.zero(); dmax.zero();
    hipLaunchKernelGGL(sl_in_degree_kernel, dim3(grid), dim3(256), 0, stream(), nnz, o->dev.nbr, indeg.p);
    hipLaunchKernelGGL(sl_max_u32_kernel, dim3(grid_cap(n, 256, 1024)), dim3(256), 0, stream(), n, (const uint32_t*)indeg.p, dmax.p);
    check_launch("sl_in_degree");
    uint32_t indeg_max = 0;
    dmax.download(&indeg_max, 1);
    o->sl_max_in_degree = indeg_max;
    if (debug_knob("AE_SL_NO_MATCH")) {  // A/B: everything through the optimistic passes (the first form of the mode)
        all_optimistic();
        return;
    }
    // Classes or none.  A step is a launch: ~9 us of latency whatever it holds; an event in a step costs 0.14 ns (rows of <= 8 columns:
    // ~5.6 random requests at the ~55 G requests/s the memory system serves) to 0.24 ns (wider rows: one wave per SIMD); an event of the
    // overflow class 0.30 / 0.28 ns (two owner marks, two checks, a pending-list trip, 1.4 - 2 attempts) and the class two to four
    // launches per slice -- and ONE event per row and pass: the busiest row's events of a slice are so many passes.  Graphs of a few
    // million edges (steps of a few thousand events) run everything optimistically; large graphs and graphs with hubs in classes.
    // Constants measured on MI355X.
    const uint32_t kmax = o->g->max_nbng;
    uint32_t classes = std::min<uint32_t>(kMaxClasses, kmax + std::max<uint32_t>(4u, (4u * kmax + 4u) / 5u));
    const double events = (double)o->params.nb_sampling_by_edge * (double)nnz;
    // More classes where they cost nothing: a batch whose steps are cut to what the device holds at once (ce_slice_gradient_iteration:
    // "slices as thin as a resident step") runs events / resident steps however many classes share a slice, so the palette may grow
    // until the slices are back at their lambda thickness -- and a wider palette leaves less to the overflow class (11 M-node kNN
    // graph, k = 6: 11 classes 1.9 %, 13: 0.4 %, 15: 0.07 %; 166 -> 160 ms per batch).  One device only: the ranks of a sharded run
    // must cut their batches alike, and the slice count follows the classes.
    const bool whole = o->dev.node_lo == 0 && o->dev.node_hi == n;
    if (whole && !debug_knob("AE_SL_NO_FIT") && !debug_knob("AE_SL_BASE_CLASSES")) {
        const double slices_lambda = std::max(1.0, std::ceil(2.0 * events / (double)n / 0.5));
        const double fit = events / (sl_resident_events(o) * slices_lambda);
        if (fit > (double)classes && fit < 4.0 * (double)classes)   // (the regime in which the slices are thinned: a step of 1 ... 4 device loads)
            classes = std::min<uint32_t>(std::min<uint32_t>(kMaxClasses, classes + 4u), (uint32_t)fit);
    }
    bool merged_regime = false;
    // ... and where the slices run MERGED (under-filled steps: a rank's share of a sharded batch, a graph of ~10^6 nodes; the rule of
    // ce_slice_gradient_iteration): one launch holds every class of a slice, so four more classes cost nothing and leave next to nothing
    // to the overflow class and its passes (configs[3]'s graph, a rank of 8: 11 classes 1.9 % -- 39.3 ms per batch; 15: 0.07 % --
    // 35.6; 19: 35.2).  The ranks of a sharded run must agree on the palette: the share that enters is the LARGEST rank's.
    {
        double share = (double)(o->dev.node_hi - o->dev.node_lo) / (double)n;
        if (o->comm)
            for (size_t q = 0; 2 * q + 1 < o->comm_ranges.size(); q++) share = std::max(share, (double)(o->comm_ranges[2 * q + 1] - o->comm_ranges[2 * q]) / (double)n);
        const double slices_lambda = std::max(1.0, std::ceil(2.0 * events / (double)n / 0.5));
        const double per_step = events * share / (slices_lambda * (double)classes);
        merged_regime = per_step < kMergeBelow * sl_resident_events(o) && !debug_knob("AE_SL_NO_MERGE");
        if (merged_regime && !debug_knob("AE_SL_BASE_CLASSES")) classes = std::min<uint32_t>(std::min<uint32_t>(kMaxClasses, kDepBits), classes + 4u);
    }
    if (debug_knob("AE_SL_CLASS_CAP")) classes = std::min<uint32_t>(kMaxClasses, std::max<int>((int)kmax + 1, atoi(debug_knob("AE_SL_CLASS_CAP"))));
    const double slices = std::max(1.0, 4.0 * events / (double)n);
    const double busiest = 0.5 * (double)(indeg_max + kmax) * (double)n / (double)(2 * nnz);   // events per slice on the busiest row
    const double c_match = o->dev.dim <= 8 ? 0.14e-9 : 0.24e-9;
    const double c_opt = debug_knob("AE_SL_COPT") ? atof(debug_knob("AE_SL_COPT")) * 1e-9 : (o->dev.dim <= 8 ? 0.30e-9 : 0.28e-9);
    double cost_none = slices * std::max(4.0, busiest) * 9e-6 + events * c_opt;
    double cost_classes = slices * (classes + 2.0) * (9e-6 + busiest / classes * 0.4e-6) + events * (0.99 * c_match + 0.01 * c_opt);
    if (merged_regime && !debug_knob("AE_SL_COPT")) {
        // Merged slices: ONE launch per slice whatever the palette.  Measured on exact kNN graphs of Higgs-shaped points (k = 6, 2
        // columns, in-degrees to ~140; tools/run_auto_crossover.py), ms per batch at 24 / 48 / 72 / 99 M events: class path merged 18.5 /
        // 21.4 / 27.1 / 33.8 (57 us per slice + 0.20 ns per event), everything optimistic 29.7 / 34.5 / 44.1 / 51.1 (the busiest row's
        // ~6 events per slice at 16 us each + 0.285 ns per event: hubs lose passes to conflicts); on a lattice (uniform in-degree)
        // the optimistic passes take 28.4 ms at 99 M events (4 launches of 9 us per slice + 0.20 ns per event) and win.
        const bool skewed = indeg_max > 4u * kmax;
        cost_classes = slices * 57e-6 + events * 0.20e-9;
        cost_none = slices * std::max(4.0, busiest) * (skewed ? 16e-6 : 9e-6) + events * (skewed ? 0.285e-9 : 0.20e-9);
    }
    if (cost_none <= cost_classes && !debug_knob("AE_SL_CLASS_CAP") && !debug_knob("AE_SL_FORCE_CLASSES")) {
        all_optimistic();
        return;
    }
    DevBuf<unsigned long long> out_used, in_used, prop_mask, remaining;
    DevBuf<uint8_t> prop;
    out_used.alloc_pooled(n); in_used.alloc_pooled(n); prop_mask.alloc_pooled(n); remaining.alloc_pooled(1024); prop.alloc_pooled(nnz);
    out_used.zero(); in_used.zero();
    // phase A: the nodes whose in-edges alone would fill the in-palette (classes - out-degree colours) colour their out-edges first
    const uint32_t first_indeg = classes > kmax + 2u ? classes - kmax - 2u : 1u;
    uint32_t round = 0;
    unsigned long long prev_left = ~0ull;
    int stalled = 0;
    for (; round < 40; round++) {
        const bool phase_a = round < 4;
        remaining.zero();
        hipLaunchKernelGGL(sl_star_propose_kernel, dim3(ngrid), dim3(256), 0, stream(), o->dev, (const uint8_t*)o->sl_color.p,
                           (const unsigned long long*)out_used.p, (const unsigned long long*)in_used.p, (const uint32_t*)indeg.p,
                           phase_a ? first_indeg : 0u, prop_mask.p, prop.p, classes, pcg_hash(round ^ kTagSlColor ^ (uint32_t)o->dev.seed));
        hipLaunchKernelGGL(sl_star_commit_kernel, dim3(ngrid), dim3(256), 0, stream(), o->dev, o->sl_color.p, out_used.p, in_used.p,
                           (const unsigned long long*)prop_mask.p, (const uint8_t*)prop.p, classes, remaining.p);
        if (phase_a) continue;
        unsigned long long left = 0;
        for (unsigned long long v : remaining.to_host()) left += v;
        if (!left) { round++; break; }
        stalled = left >= prev_left ? stalled + 1 : 0;   // (what is left has no colour to take: it will not get one)
        prev_left = left;
        if (stalled >= 2) { round++; break; }
    }
    hipLaunchKernelGGL(sl_color_giveup_kernel, dim3(grid), dim3(256), 0, stream(), nnz, o->sl_color.p);
    check_launch("sl_color");
    DevBuf<double> mass;
    mass.alloc_pooled(kMaxClasses + 1);
    mass.zero();
    hipLaunchKernelGGL(sl_class_mass_kernel, dim3(grid_cap(nnz, 256, 2048)), dim3(256), 0, stream(), nnz, (const EdgeRec*)erec, (const uint8_t*)o->sl_color.p, mass.p);
    std::vector<double> hm = mass.to_host();
    double total = 0.;
    for (double v : hm) total += v;
    o->sl_classes = classes;
    o->sl_ov_frac = total > 0. ? hm[kMaxClasses] / total : 0.;
    o->sl_color_rounds = round;
    finish();
    if (debug_knob("AE_CE_PROF"))
        fprintf(stderr, "CESLICE colouring: largest in-degree %u, %u classes (in-stars), %u rounds, overflow mass %.4f\n", indeg_max, classes, round, o->sl_ov_frac);
}

void ce_slice_prepare(ae_entropy_optim* o) {
    const ae_kgraph* g = o->g;
    if (ce_slice_unsupported(o)) return;  // reported by the first batch
    o->sl_erec.alloc(g->nnz * 4);  // EdgeRec as four words
    hipLaunchKernelGGL(sl_edge_rec_kernel, dim3(blocks_for(g->n, 256)), dim3(256), 0, stream(), o->dev, reinterpret_cast<EdgeRec*>(o->sl_erec.p));
    // static records: 16 floats serve rows of <= 7 neighbours, 32: <= 15, 64: <= 31, 128: 32
    o->sl_srec_floats = g->max_nbng <= 7 ? 16u : (g->max_nbng <= 15 ? 32u : (g->max_nbng <= 31 ? 64u : 128u));
    // Internal numbering (one device): a uniform random relabelling of the nodes.  The mode reads RUNS of consecutive rows (the windows
    // of the tile of negatives); with the caller's labels a run is whatever the caller put side by side -- a graph stored component by
    // component gave every window to one cluster, and the result moved by 1.5-2.5 % in CE (lower quartile of the edge lengths -9 ...
    // +12 %, eight seeds a side, 1 M Higgs-shaped points) where the same graph with shuffled labels matched the exact mode.  With the
    // relabelling the two are the same run.  The coordinates move to the internal order at the start of a batch and back at its
    // end (two row moves: ~1 % of a batch); everything between the two -- records, events, pending lists -- speaks internal numbers.
    // A sharded range keeps the caller's labels: its ranks own label ranges and exchange rows in place.
    o->sl_perm.release();
    o->sl_y_lines = 0;   // (the node lines' static part names the neighbours by their internal numbers: a new numbering, a new fill)
    const bool whole_range = o->dev.node_lo == 0 && o->dev.node_hi == g->n;
    // (a sharded range: once the communicator is attached and every rank's range is known -- entropy_optim_attach_comm prepares again --
    // the same relabelling on every rank, inside every rank's range: a rank's rows stay one contiguous run, the exchanges stay in place)
    // (measurement only, AE_SL_ASSUME_WORLD = W under AE_DEBUG_KNOBS: a range WITHOUT a communicator prepares as rank r of W equal ranges would
    // with one -- relabelled inside the ranges, hence tile and node lines -- so that a rank's share can be timed alone on one GPU:
    // tools/run_shard_time.py)
    std::vector<uint64_t> assumed_ranges;
    if (!whole_range && !o->comm && debug_knob("AE_SL_ASSUME_WORLD")) {
        const uint64_t w = (uint64_t)std::max(2, atoi(debug_knob("AE_SL_ASSUME_WORLD"))), base = g->n / w, rem = g->n % w;
        for (uint64_t r = 0; r < w; r++) {
            const uint64_t lo = r * base + std::min(r, rem);
            assumed_ranges.push_back(lo);
            assumed_ranges.push_back(lo + base + (r < rem ? 1 : 0));
        }
    }
    const std::vector<uint64_t>& rank_ranges = assumed_ranges.empty() ? o->comm_ranges : assumed_ranges;
    const bool ranges_known = !whole_range && (o->comm || !assumed_ranges.empty()) && rank_ranges.size() >= 4;
    if ((whole_range || ranges_known) && !debug_knob("AE_SL_LABEL_ORDER")) {
        DevBuf<uint32_t> keys, keys_out, ident, order;
        keys.alloc_pooled(g->n); keys_out.alloc_pooled(g->n); ident.alloc_pooled(g->n); order.alloc_pooled(g->n);
        hipLaunchKernelGGL(sl_perm_keys_kernel, dim3(blocks_for(g->n, 256)), dim3(256), 0, stream(), (uint64_t)g->n,
                           pcg_hash((uint32_t)o->dev.seed ^ 0x51ED270Bu), keys.p, ident.p);
        sort_pairs_u32_u32(keys.p, keys_out.p, ident.p, order.p, g->n, 32);
        if (ranges_known) {   // stable sort by owner: the random order survives inside every range
            const uint32_t world = (uint32_t)(rank_ranges.size() / 2);
            DevBuf<uint64_t> d_ranges;
            d_ranges.alloc_pooled(rank_ranges.size());
            d_ranges.upload(rank_ranges.data(), rank_ranges.size());
            hipLaunchKernelGGL(sl_perm_block_kernel, dim3(blocks_for(g->n, 256)), dim3(256), 0, stream(), (uint64_t)g->n, (const uint32_t*)order.p,
                               (const uint64_t*)d_ranges.p, world, keys.p);
            unsigned bits = 1;
            while ((1u << bits) < world) bits++;
            sort_pairs_u32_u32(keys.p, keys_out.p, order.p, ident.p, g->n, bits);
            sync();   // (d_ranges is read by the kernel above)
            std::swap(order.p, ident.p);
        }
        o->sl_perm.alloc(g->n);
        hipLaunchKernelGGL(sl_perm_invert_kernel, dim3(blocks_for(g->n, 256)), dim3(256), 0, stream(), (uint64_t)g->n, (const uint32_t*)order.p, o->sl_perm.p);
        if (o->dev.hub_odds) {
            o->sl_hub_tab.alloc(g->n);
            hipLaunchKernelGGL(sl_hub_tab_kernel, dim3(blocks_for(g->n, 256)), dim3(256), 0, stream(), (uint64_t)g->n, (const uint32_t*)o->sl_perm.p,
                               o->dev.hub_tab, o->sl_hub_tab.p);
        }
        check_launch("sl_perm");
        sync();
    }
    o->sl_srec.alloc(g->n * o->sl_srec_floats);
    hipLaunchKernelGGL(sl_static_rec_kernel, dim3(blocks_for(g->n, 256)), dim3(256), 0, stream(), o->dev, o->sl_srec_floats,
                       (const uint32_t*)(o->sl_perm.n ? o->sl_perm.p : nullptr), o->sl_srec.p);
    check_launch("sl_prepare");
    // largest edge probability (segments keep the per-edge Poisson mean below 64)
    std::vector<float> hp = o->np->proba.to_host();
    float pmax = 0.f;
    for (float v : hp) pmax = std::max(pmax, v);
    o->sl_pmax = pmax;
    slice_color_edges(o);
    // (A sharded range runs a cross-shard edge as two half events, each against a replica of the far end that is as old as the last
    // exchange: fine for a few per cent of the edges, not for a graph in arbitrary order.  The limit is enforced where every rank takes
    // the same decision: entropy_optim_attach_comm, comm.hip; a range without a communicator: the first batch, below.)
    o->sl_owner.alloc(2 * g->n);
    AE_HIP(hipMemsetAsync(o->sl_owner.p, 0xFF, sizeof(uint32_t) * 2 * g->n, stream()));
    o->sl_counts.alloc(3 * kSub);
    o->sl_counts.zero();
    o->sl_done.alloc(1025);
    o->sl_done.zero();
    o->sl_chunk_flag.release();
    if (!o->sample_counter.n) { o->sample_counter.alloc(1024); o->sample_counter.zero(); }
    o->sl_prepared = true;
}

// Floats of a node's line where a batch with one launch per class keeps rows and static records together (ce_slice_kernels.h: LineRec);
// 0: this handle's steps read dense rows and the static records (rows of more than 16 columns, lines of more than 128 bytes, no
// internal copy to keep them in).  Lines live in the relabelled internal copy of the coordinates: one device, or a sharded range whose
// ranks' ranges are known.
static uint32_t sl_node_line(const ae_entropy_optim* o) {
    if (!o->sl_perm.n || debug_knob("AE_SL_NO_LINES")) return 0u;
    return (uint32_t)node_line_floats((int)o->dev.dim, (int)o->g->max_nbng);
}
static void launch_step_line(uint32_t dim, const DirectArgs& da, uint32_t line, bool f64, bool tile, int* blocks_per_cu) {
    switch (dim) {
        case 2: launch_direct_line<2>(da, line, f64, tile, blocks_per_cu); break;
        case 3: launch_direct_line<3>(da, line, f64, tile, blocks_per_cu); break;
        case 4: launch_direct_line<4>(da, line, f64, tile, blocks_per_cu); break;
        case 8: launch_direct_line<8>(da, line, f64, tile, blocks_per_cu); break;
        case 16: launch_direct_line<16>(da, line, f64, tile, blocks_per_cu); break;
        default: fail(AE_ERR_INVALID_ARG, "internal: no node lines for rows of %u columns", dim);
    }
}

// events a step can hold with ALL its workgroups resident at once (the step kernel's occupancy x CUs x 256, less 3 %: the classes'
// sizes and the Poisson totals scatter)
static double sl_resident_events(ae_entropy_optim* o) {
    int dev = 0, bpc = 1;
    hipDeviceProp_t prop;
    AE_HIP(hipGetDevice(&dev));
    AE_HIP(hipGetDeviceProperties(&prop, dev));
    const bool sharded_range = o->dev.node_lo != 0 || o->dev.node_hi != o->dev.n;
    const bool tile_fit = !debug_knob("AE_SL_NO_TILE") && (!sharded_range || o->sl_perm.n != 0 || debug_knob("AE_SL_SHARD_TILE")) &&
                          (uint64_t)o->dev.n * o->dev.dim * 4ull > (4ull << 20);
    if (const uint32_t line = sl_node_line(o)) {   // (the instantiation a full step runs)
        DirectArgs none{};
        none.c = o->dev;
        launch_step_line((uint32_t)o->dev.dim, none, line, o->params.ce_precision != AE_PRECISION_F32, tile_fit, &bpc);
    } else {
        AE_DISPATCH_DIM(o->dev.dim, direct_blocks_per_cu, o->sl_srec_floats, o->params.ce_precision != AE_PRECISION_F32, tile_fit, &bpc);
    }
    return 0.97 * 256.0 * (double)bpc * (double)prop.multiProcessorCount;
}

void ce_slice_gradient_iteration(ae_entropy_optim* o, uint64_t nb_sample, double grad_step, uint32_t iter) {
    if (const char* why = ce_slice_unsupported(o)) fail(AE_ERR_INVALID_ARG, "AE_CE_SLICED: %s", why);
    if (!o->sl_prepared) ce_slice_prepare(o);   // (a sharded range nobody attached a communicator to)
    const uint64_t n = o->dev.n, nnz = o->dev.nnz;
    if ((o->dev.node_lo != 0 || o->dev.node_hi != n) && !o->comm && o->sl_cross_frac > kMaxCrossShardMass && !debug_knob("AE_SL_ANY_PARTITION"))
        fail(AE_ERR_INVALID_ARG, "AE_CE_SLICED on nodes [%llu, %llu): %.1f %% of the shard's edge probability mass lies on cross-shard edges (limit %.0f %%): "
                                 "order the nodes by locality / connected component before sharding (ae_kgraph_partition), or ask for the approximate rounds mode (AE_CE_HOGWILD)",
             (unsigned long long)o->dev.node_lo, (unsigned long long)o->dev.node_hi, 100. * o->sl_cross_frac, 100. * kMaxCrossShardMass);
    // A sharded node range (multi-GPU): nb_sample counts this shard's samples (nb_sampling_by_edge x its edges); everything that shapes
    // the batch -- segments, slices, exchange points -- follows the WHOLE graph's total so that every rank cuts the batch alike, and the
    // per-edge Poisson means are the whole graph's law (edges are drawn in proportion to p_e over the whole graph, embedder.rs:987).
    const bool sharded = o->dev.node_lo != 0 || o->dev.node_hi != n;
    double total_samples = (double)nb_sample;
    if (sharded) {
        if (nb_sample % o->dev.shard_edges) fail(AE_ERR_INVALID_ARG, "AE_CE_SLICED on a sharded range: nb_sample must be a multiple of the shard's edges (nb_sampling_by_edge x edges)");
        total_samples = (double)(nb_sample / o->dev.shard_edges) * (double)nnz;
    }
    const uint64_t n_gen = o->sl_gen_edges;   // edges this handle generates events for
    const int world = o->comm ? comm_world(o->comm) : 1;
    const double per_node = total_samples / (double)n;
    // segments of the batch: per-edge Poisson mean <= 64 (f32 inversion: exp(-64) is a normal number, the count is capped at 255),
    // at most 2^30 events per segment
    uint32_t segments = (uint32_t)std::max(1.0, std::ceil(per_node * (double)o->sl_pmax / 64.0));
    segments = std::max(segments, (uint32_t)(total_samples / (double)(1ull << 30)) + 1u);
    if (iter >= (1u << 20) || segments >= 4096) fail(AE_ERR_INVALID_ARG, "AE_CE_SLICED: batch / segment index too large for the RNG key");
    const double seg_samples = total_samples / segments;                                        // the whole graph's events of a segment
    const double seg_local = seg_samples * o->sl_gen_mass / (double)n;                          // this handle's (half events included)
    const double seg_rank = o->comm ? seg_samples / (double)world : seg_local;                  // what every rank sizes its steps by
    // slices of a segment: about half an event per node and slice (a sample is an event at two nodes)
    const double lambda_s = debug_knob("AE_SL_LAMBDA") ? atof(debug_knob("AE_SL_LAMBDA")) : 0.5;
    uint32_t n_slices = (uint32_t)std::max(1.0, std::ceil(2.0 * seg_samples / (double)n / lambda_s));
    // A step is bound by latency, not by its size, as long as ALL its workgroups are resident at once: its time is one chain of
    // memory round trips (~28 us at the configs[3] shape for 98 k or 196 k events alike); what does not fit starts when the first
    // workgroups end -- a second chain (measured: 250 k events on 768 resident workgroups of 256: 59 us).  So the slices are made as
    // thin as it takes for a step to fit the device (never thicker than lambda: thinner slices are the more faithful ones).
    // (a sharded run: the overflow share the ranks agreed on when the communicator was attached -- the parallel colouring's own share
    // depends on the scheduling, and every rank must cut its batch alike)
    const double ov_sched = (o->comm && o->sl_ov_frac_sched >= 0.) ? o->sl_ov_frac_sched : o->sl_ov_frac;
    if (o->sl_classes && !debug_knob("AE_SL_NO_FIT")) {
        const double resident = sl_resident_events(o);
        const double per_step = seg_rank * (1.0 - ov_sched) / ((double)n_slices * (double)o->sl_classes);
        if (per_step > resident && per_step < 4.0 * resident)
            n_slices = (uint32_t)std::ceil(seg_rank * (1.0 - ov_sched) / (resident * (double)o->sl_classes));
    }
    // A rank of a sharded run holds a fraction of every step, and a step costs ~25-30 us however few events it has (one chain of memory
    // round trips): configs[3] over 8 ranks runs 2 640 steps of 31 k events -- 81 ms per batch and rank where one device takes 160 for
    // the whole graph.  Thicker slices on a rank's under-filled steps (up to lambda = 1: half the steps, 81 -> 52 ms) were built and
    // measured and are NOT the default: on one device lambda = 1 stays within ~1 % of lambda = 1/2, but an 11 M-node
    // 64-component graph in 4 and 8 shards came out with its edges 10-15 % short (CE +2 %), where lambda = 1/2 in 2 shards matched
    // the exact mode to 0.5 % (those runs had ONE exchange per batch, which alone costs that much: DESIGN 5) -- and with 4 exchanges
    // per batch a 600 k-node graph of 64 tight components in 2 shards still came out 13-18 % short at lambda = 1 where lambda = 1/2
    // sits at +4 ... +9 % (the 11 M-node graph: no difference).  How much a thick slice shows depends on the graph.  Kept behind a knob.
    if (sharded && o->sl_classes && !debug_knob("AE_SL_LAMBDA") && debug_knob("AE_SL_THICK")) {   // (A/B only: see above)
        const double per_step = seg_rank * (1.0 - o->sl_ov_frac) / ((double)n_slices * (double)o->sl_classes);
        const double resident = sl_resident_events(o);   // (a step up to what the device holds at once: 125 k events cost what 190 k do)
        if (per_step < resident) {
            const double lam = std::min(1.0, lambda_s * resident / per_step);
            n_slices = (uint32_t)std::max(1.0, std::ceil(2.0 * seg_samples / (double)n / lam));
        }
    }
    // An overflow class that is next to empty (a wide palette: < 0.2 % of the events) still costs a mark and two passes per slice --
    // three launches of ~5 us for a few hundred events, 4 ms of a 33 ms rank share at N = 8.  Its events are dealt to every 8th slice
    // instead (the middle one of their group of eight: sl_fill_kernel; an event of the class moves by at most four slices of ~240, its
    // place in the batch stays uniform and independent of every other event's), and only those slices run the class.
    const uint32_t ov_every = (o->sl_classes && o->sl_ov_frac > 0. && o->sl_ov_frac < 0.002 && n_slices >= 64u && !debug_knob("AE_SL_OV_EVERY_SLICE")) ? 8u : 1u;
    const uint32_t ov_slices = (n_slices + ov_every - 1u) / ov_every;   // slices that run the overflow class
    // passes per slice of the overflow class: a thin one (a few per cent of the events: conflicts among them are rare) runs once and
    // carries its losers into the next slice
    int passes = debug_knob("AE_SL_PASSES") ? atoi(debug_knob("AE_SL_PASSES")) : (o->sl_ov_frac < 0.05 ? 1 : 3);
    // ... and as many as the BUSIEST row of the class needs: a pass runs one event per row, and a row that receives more overflow
    // events per slice than the slice has passes runs its events late -- at the end of the batch, all of them after everybody else's.
    // Measured on a graph of 16 tight components (64 k nodes, k = 6, in-degrees up to ~100, everything optimistic): 3 passes for ~4
    // events per slice on the busiest rows ended at CE 0.90-0.96x the sequential mode's (edge lengths +5 ... +20 %) although the late
    // events were < 1 % of all (the round-3 criterion for draining inside the slice); 6-8 passes: 1.000.
    if (!debug_knob("AE_SL_PASSES") && o->sl_ov_frac > 0.) {
        const double busiest = (double)o->sl_node_ov_max * (seg_samples / (double)n) / (double)ov_slices;
        passes = std::max(passes, (int)std::min(16.0, std::ceil(1.5 * busiest + 1.0)));
    }
    const int spread = debug_knob("AE_SL_NO_SPREAD") ? 0 : (debug_knob("AE_SL_SPREAD_ALL") ? 1 : 2);
    // scalar arithmetic: the reference's f64 (embedder.rs:1207-1229) unless the caller opted into f32 (ae_embedder_params.ce_precision)
    const bool f64 = o->params.ce_precision != AE_PRECISION_F32;
    const uint32_t classes = o->sl_classes;
    const bool has_overflow = o->sl_ov_frac > 0.;
    // Under-filled steps (a rank's share of a sharded batch, graphs of ~10^6 nodes): the classes of a slice in ONE launch, ordered node
    // by node (sl_slice_kernel) -- a step of few events is a chain of latencies whatever it holds, and a slice is k + 5 of them.
    // Full steps stay one launch per class: they are bound by requests, and the dependency words would only add to them.
    // (a sharded run: decided from what every rank shares -- the rank-agreed share of a segment and overflow fraction, as the slice
    // count above -- because the form decides which label space the in-batch exchanges move: ranks on different sides of the
    // threshold would read each other's rows in the wrong numbering)
    const double step_events = classes ? seg_rank * (1.0 - ov_sched) / ((double)n_slices * (double)classes) : 0.;
    const bool merged = classes && classes <= kDepBits && !debug_knob("AE_SL_NO_MERGE") &&
                        (debug_knob("AE_SL_MERGE") || step_events < kMergeBelow * sl_resident_events(o));
    // The events' sort keys.  Merged slices: slice * (classes + 1) + the POSITION of the event's class in the slice's order (the launch
    // walks the classes in that order).  One launch per class (round 6): (slice << kClsBits) | class -- the edges are generated in
    // class order (slice_color_edges), so a stable sort on the SLICE bits alone leaves the events sorted by (slice, class): one radix
    // pass of 8 bits for up to 256 slices where the composite key took two; the host walks a slice's classes in the drawn order.
    const bool slice_keys = !merged && !debug_knob("AE_SL_COMPOSITE_KEYS");
    const uint32_t kstride = slice_keys ? (1u << kClsBits) : classes + 1u;   // step pointers per slice
    const uint64_t n_keys = (uint64_t)n_slices * kstride;
    if (n_keys >= (1ull << 31)) fail(AE_ERR_INVALID_ARG, "AE_CE_SLICED: too many steps in a batch");
    o->rounds = segments * n_slices;
    const uint64_t ev_cap = (uint64_t)(seg_local + 8.0 * std::sqrt(seg_local) + 1024.0);
    // pending lists of the overflow class: a slice's overflow events (+ 16 sigma) four times over, plus what the rows that receive
    // more overflow events than a slice's passes can run (hubs) accumulate until the drain
    const double per_slice_ov = seg_local / ov_slices * o->sl_ov_frac;
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
                           (float)(seg_samples / (double)n), 0.25f * (float)ov_slices, d_busy.p);
        d_busy.download(&busy, 1);
        if (busy > 0.) passes = 3;
    }
    if (has_overflow) {
        DevBuf<double> d_backlog;
        d_backlog.alloc_pooled(1);
        d_backlog.zero();
        hipLaunchKernelGGL(sl_backlog_kernel, dim3(grid_cap(n, 256, 1024)), dim3(256), 0, stream(), n, (const float*)o->sl_node_ov.p,
                           (float)(seg_samples / (double)n), (float)(passes * ov_slices), d_backlog.p);
        d_backlog.download(&backlog, 1);
    }
    const uint64_t cap = (uint64_t)((4.0 * per_slice_ov + 16.0 * std::sqrt(per_slice_ov) + 2.0 * backlog) / kSub + 8192.0);  // per sub-list
    if (o->sl_cnt.n < n_gen + 1) { o->sl_cnt.alloc(n_gen + 1); o->sl_offs.alloc(n_gen + 1); }
    if (o->sl_keys0.n < ev_cap) { o->sl_keys0.alloc(ev_cap); o->sl_keys1.alloc(ev_cap); o->sl_vals0.alloc(3 * ev_cap); o->sl_vals1.alloc(3 * ev_cap); }   // (Event: three words)
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
    const unsigned kbegin = slice_keys ? kClsBits : 0u;   // (the class bits are in order already)
    // Rows of <= 8 columns keep the two sets of dependency words BEHIND the node's row in the batch's internal copy (rows 32 / 64
    // bytes apart instead of 8 ... 32): a word comes with the line of the row the event reads anyway and is wiped into the line it
    // writes anyway -- four of a merged event's six extra requests.
    const bool words_in_rows = merged && o->dev.dim <= 8 && !debug_knob("AE_SL_DEP_ARRAY");
    // One launch per class (full steps, bound by the number of requests that miss the L2): NODE LINES -- the internal copy keeps a node's
    // embedded scale and neighbour ids behind its row, the step kernel fetches a source's line as ONE request (ce_slice_kernels.h: LineRec)
    const uint32_t line = (classes && !merged) ? sl_node_line(o) : 0u;
    const bool strided = words_in_rows || line != 0u;   // the internal copy's rows are further apart than the caller's
    o->sl_last_form = !classes ? AE_SLICE_OPTIMISTIC : (merged ? AE_SLICE_MERGED : (line ? AE_SLICE_PER_CLASS_LINES : AE_SLICE_PER_CLASS));
    const uint32_t ystride = words_in_rows ? (o->dev.dim <= 4 ? 8u : 16u) : (line ? line : (uint32_t)o->dev.dim);
    // (floats from the start of a node's line: rows of 8 columns sit in the middle of their line, a set of words on either side --
    // ce_slice_kernels.h: LineFetch --; shorter rows at its start, the two sets behind them)
    const bool lines = words_in_rows && (o->dev.dim == 8 || o->dev.dim == 2);
    const uint32_t row_at = !lines ? 0u : (o->dev.dim == 8 ? (uint32_t)LineShape<8>::kRowAt : (uint32_t)LineShape<2>::kRowAt);
    const uint32_t odd_at = !lines ? 0u : (o->dev.dim == 8 ? (uint32_t)LineShape<8>::kOddWordAt : (uint32_t)LineShape<2>::kOddWordAt);
    const uint32_t word_at[2] = {lines ? 0u : (((uint32_t)o->dev.dim + 1u) & ~1u), lines ? odd_at : (((uint32_t)o->dev.dim + 1u) & ~1u) + 2u};
    // internal numbering (ce_slice_prepare): the batch runs on a relabelled copy of the coordinates
    CeDev cdev = o->dev;
    cdev.ystride = ystride;
    const bool relabelled = o->sl_perm.n != 0;
    const bool own_copy = relabelled || strided;
    const uint32_t* perm = relabelled ? (const uint32_t*)o->sl_perm.p : nullptr;
    auto move_rows = [&](uint64_t v_lo, uint64_t v_hi, uint64_t skip_lo, uint64_t skip_hi, const float* src, float* dst, int to_internal) {
        if (v_hi <= v_lo) return;
        hipLaunchKernelGGL(sl_move_rows_kernel, dim3(grid_cap((v_hi - v_lo) * o->dev.dim, 256, 1u << 20)), dim3(256), 0, stream(), v_lo, v_hi, skip_lo, skip_hi,
                           (uint32_t)o->dev.dim, ystride, perm, src, dst, to_internal);
    };
    if (own_copy) {
        if (o->sl_y.n < n * ystride) { o->sl_y.alloc(n * ystride); o->sl_y_lines = 0; }
        if (words_in_rows) o->sl_y.zero();
        if (line && o->sl_y_lines != line) {   // the static part of the lines: once per handle (unless a batch of another layout came between)
            hipLaunchKernelGGL(sl_line_static_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, stream(), o->dev, line, perm, o->sl_y.p);
            check_launch("sl_line_static");
        }
        o->sl_y_lines = line;
        move_rows(0, n, 0, 0, (const float*)o->dev.y, o->sl_y.p + row_at, 1);
        cdev.y = o->sl_y.p + row_at;
        if (relabelled && o->dev.hub_odds) cdev.hub_tab = o->sl_hub_tab.p;
    }
    // (a sharded range: the in-batch exchanges act on the internal copy -- a rank's rows are the same contiguous run in both numberings --;
    // with the words behind the rows, on the caller's array as a dense staging area: the rank's rows are moved out before, the others' in after)
    struct CommY {
        ae_entropy_optim* o;
        CommY(ae_entropy_optim* oo, float* y) : o(oo) { o->comm_y = y; }
        ~CommY() { o->comm_y = nullptr; }
    } comm_y_scope(o, (own_copy && !strided) ? o->sl_y.p : nullptr);
    SliceArgs a;
    a.c = cdev;
    a.srec = o->sl_srec.p;
    a.owner = o->sl_owner.p;
    a.lists = reinterpret_cast<Pending*>(o->sl_lists.p);
    // The tile's windows are runs of consecutive rows: harmless under the internal numbering of one device, but a sharded range works
    // in the caller's labels, and with labels that carry locality (a graph stored component by component) the tile moved the result
    // of an 11 M-node run in 2 shards by -3 % in CE and +18 % in the edge lengths (gathered negatives: -0.3 % / +1.5 %).  No tile there.
    // (... unless the range is relabelled: a communicator is attached and every rank's range known, ce_slice_prepare)
    const bool use_tile = !debug_knob("AE_SL_NO_TILE") && (!sharded || o->sl_perm.n != 0 || debug_knob("AE_SL_SHARD_TILE"));
    // the LDS tile pays when a negative's row would come from beyond the L2s and the step has enough events to fill the chip anyway
    // (16 384: a quarter of the chip's workgroup slots; measured on configs[3]'s shards: 62 k events per step 107 -> 85 ms per batch
    // with the tile, 31 k: 81 -> 73)
    const uint64_t tile_min_events = debug_knob("AE_SL_TILE_MIN") ? (uint64_t)atoll(debug_knob("AE_SL_TILE_MIN")) : 16384ull;
    const bool y_in_cache = (uint64_t)n * o->dev.dim * 4ull <= (4ull << 20) && !debug_knob("AE_SL_TILE_ALWAYS");
    a.counts = o->sl_counts.p;
    a.cap = cap;
    a.step = grad_step;
    a.done_counter = o->sl_done.p;
    DirectArgs da;
    da.c = cdev;
    da.srec = o->sl_srec.p;
    da.step = grad_step;
    da.dbg = debug_knob("AE_SL_DBG") ? atoi(debug_knob("AE_SL_DBG")) : 0;
    da.done_counter = o->sl_done.p;
    // hand-over flags of the hub chains (one per 64-event chunk of the sorted event array; a step's token is its running number + 1)
    if (o->sl_chunk_flag.n < (ev_cap >> 6) + 4) o->sl_chunk_flag.alloc((ev_cap >> 6) + 4);
    o->sl_chunk_flag.zero();
    da.chunk_flag = o->sl_chunk_flag.p;
    a.hub_pool = da.hub_pool = nullptr;
    a.hub_pool_n = da.hub_pool_n = 0;
    if (o->dev.hub_odds) {   // fresh every batch
        const uint32_t pool_n = n >= (1ull << 20) ? (1u << 24) : (1u << 20);
        if (o->sl_hub_pool.n < pool_n) o->sl_hub_pool.alloc(pool_n);
        hipLaunchKernelGGL(sl_hub_pool_kernel, dim3(blocks_for(pool_n, 256)), dim3(256), 0, stream(), cdev, (uint32_t)(iter << 12), pool_n, o->sl_hub_pool.p);
        a.hub_pool = da.hub_pool = o->sl_hub_pool.p;
        a.hub_pool_n = da.hub_pool_n = pool_n;
    }
    // the edges in event-generation order (the edges of a class sorted by target: slice_color_edges) and their classes
    const EdgeRec* gen_erec = reinterpret_cast<const EdgeRec*>(o->sl_erec_gen.n ? o->sl_erec_gen.p : o->sl_erec.p);
    const uint8_t* gen_color = o->sl_color_gen.n ? o->sl_color_gen.p : o->sl_color.p;
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
    const bool use_chains = !debug_knob("AE_SL_NO_CHAIN");
    auto chain_round = [&](unsigned grid) {
        a.src_list = cur; a.dst_list = (cur + 1) % 3; a.zero_list = (cur + 2) % 3;
        a.pass_seq = pass_seq++;
        hipLaunchKernelGGL(sl_chain_link_kernel, dim3(grid, kSub), dim3(256), 0, stream(), a, o->sl_chain_head.p, o->sl_chain_next.p);
        AE_DISPATCH_DIM(o->dev.dim, launch_chain_run, a, grid, o->sl_srec_floats, f64, (const uint32_t*)o->sl_chain_head.p, (const uint32_t*)o->sl_chain_next.p);
        hipLaunchKernelGGL(sl_chain_unlink_kernel, dim3(grid, kSub), dim3(256), 0, stream(), a, o->sl_chain_head.p);
        cur = (cur + 1) % 3;
    };
    // Multi-GPU: the owned rows are all-gathered `comm_exchanges` times per segment at equal runs of slices (the last one after the
    // segment's drain): between two exchanges a shard reads the other shards' rows -- negatives, the far ends of its cross-shard
    // edges -- as of the last one.  Every rank has the same slices (they follow the whole graph's totals), hence the same exchange points.
    const uint32_t exchanges = o->comm ? std::max(1u, std::min(o->comm_exchanges, n_slices)) : 0u;
    uint64_t exchanges_done = 0;
    auto exchange_now = [&] {
        if (strided && o->comm) move_rows(o->dev.node_lo, o->dev.node_hi, 0, 0, (const float*)(o->sl_y.p + row_at), o->dev.y, 0);   // (perm null or a relabelling inside the range: the rows land in the rank's run)
        ce_comm_exchange(o);
        if (strided && o->comm) move_rows(0, n, o->dev.node_lo, o->dev.node_hi, (const float*)o->dev.y, o->sl_y.p + row_at, 1);
        exchanges_done++;
    };
    auto exchange_after = [&](uint32_t s) {
        if (exchanges < 2u) return;
        const uint32_t q = (uint32_t)(((uint64_t)(s + 1u) * exchanges) / n_slices), q0 = (uint32_t)(((uint64_t)s * exchanges) / n_slices);
        if (q != q0 && q < exchanges) exchange_now();
    };
    // A batch that fails on THIS rank (event capacity, a pending list, a poll budget) still owes the other ranks its part in the batch's
    // remaining exchanges: they are collectives the others are about to enter.  The error is reported after them.
    struct OweExchanges {
        ae_entropy_optim* o; const uint64_t& done; uint64_t owed; bool armed = true;
        ~OweExchanges() {
            if (!armed || !o->comm) return;
            try { for (uint64_t x = done; x < owed; x++) ce_comm_exchange(o); } catch (...) {}
        }
    } owe{o, exchanges_done, (uint64_t)segments * exchanges};
    if (merged && !words_in_rows) {   // two sets of words: a slice runs on one while the next slice's events enter the other
        if (o->sl_dep.n < 2 * n) o->sl_dep.alloc(2 * n);
        o->sl_dep.zero();
    }
    const int chains_dbg = (!merged && debug_knob("AE_SL_CHAINS_DBG")) ? std::min(8, atoi(debug_knob("AE_SL_CHAINS_DBG"))) : 0;
    static hipStream_t dbg_streams[8] = {};
    static hipEvent_t dbg_ev = nullptr;
    if (chains_dbg > 1 && !dbg_ev) {
        AE_HIP(hipEventCreateWithFlags(&dbg_ev, hipEventDisableTiming));
        for (int pc = 0; pc < 8; pc++) AE_HIP(hipStreamCreateWithFlags(&dbg_streams[pc], hipStreamNonBlocking));
    }
    // EXPERIMENT (AE_SL_NEG_SNAPSHOT = k, one launch per class only): the negatives' rows come from a copy of the coordinates taken every k
    // slices -- what a form whose negatives are a slice old (merged slices: drawn and fetched at the launch's start) does to the statistics
    const int neg_snapshot = (!merged && debug_knob("AE_SL_NEG_SNAPSHOT")) ? std::max(1, atoi(debug_knob("AE_SL_NEG_SNAPSHOT"))) : 0;
    if (neg_snapshot && o->sl_neg_snap.n < n * ystride) o->sl_neg_snap.alloc(n * ystride);
    bool premarked = false;   // the slice about to run had its words filled by the slice before it
    uint32_t step_seq_base = 0;
    // THE CLASS WINDOW of the merged launches (SliceRunArgs::window): half the palette.  A merged launch without it runs every class of a
    // slice at once, so an event reads its negatives' rows as up to a whole slice of its predecessors has NOT yet moved them -- the form's
    // bias of round 5 (stiff 2-D blobs, 256 seeds a side against one launch per class: CE +0.34 +- 0.23 %, median edge -0.69 +- 0.43 %).
    // Window 8 of 15 classes: +0.12 +- 0.22 % / -0.18 +- 0.43 %; 6: +0.03 / -0.17; 4 and 2: nothing at 64 seeds.  What it costs is the
    // concurrency it takes away: configs[2]'s large graph 29.4 ms per batch without, 32.2 with 8, 40.8 with 6, 54.6 with 4 (one launch
    // per class: 50.8).  profiles/r06/r6_blobs_window*.txt, r6_c3_window_ab2.jsonl.  AE_SL_WINDOW (debug knob): another width, 0 = none.
    const uint32_t window = !merged ? 0u : (debug_knob("AE_SL_WINDOW") ? (uint32_t)std::max(0, atoi(debug_knob("AE_SL_WINDOW"))) : (classes + 1u) / 2u);
    if (window && o->sl_class_done.n < (uint64_t)n_slices * classes) o->sl_class_done.alloc((uint64_t)n_slices * classes);
    if (merged && window) o->sl_last_form = AE_SLICE_MERGED_WINDOW;
    auto slice_args = [&](uint32_t s) {
        SliceRunArgs ra;
        ra.window = window;
        ra.class_done = window ? o->sl_class_done.p + (size_t)s * classes : nullptr;
        ra.d = da;
        ra.d.ept = 1;
        ra.sptr = o->sl_sptr.p + (size_t)s * (classes + 1u);
        ra.classes = classes;
        ra.step_seq0 = step_seq_base + s * classes;
        ra.next_sptr = nullptr;
        ra.set = s & 1u;
        ra.lines = lines ? 1u : 0u;
        if (words_in_rows) {
            ra.dep = reinterpret_cast<unsigned long long*>(o->sl_y.p + word_at[s & 1u]);
            ra.dep_next = reinterpret_cast<unsigned long long*>(o->sl_y.p + word_at[(s + 1u) & 1u]);
            ra.dep_stride = ystride / 2u;
        } else {
            ra.dep = o->sl_dep.p + (size_t)(s & 1u) * n;
            ra.dep_next = o->sl_dep.p + (size_t)((s + 1u) & 1u) * n;
            ra.dep_stride = 1u;
        }
        return ra;
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
        hipLaunchKernelGGL(sl_count_kernel, dim3(blocks_for(n_gen, 256)), dim3(256), 0, stream(), o->dev, n_gen, gen_erec, (float)(seg_samples / (double)n), key, o->sl_cnt.p);
        {
            size_t tmp_bytes = 0;
            if (rocprim::exclusive_scan(nullptr, tmp_bytes, o->sl_cnt.p, o->sl_offs.p, 0u, n_gen, rocprim::plus<uint32_t>(), stream()) != hipSuccess)
                fail(AE_ERR_NO_DEVICE, "rocprim exclusive_scan (size query) failed");
            DevBuf<char> tmp;
            tmp.alloc_pooled(tmp_bytes ? tmp_bytes : 1);
            if (rocprim::exclusive_scan(tmp.p, tmp_bytes, o->sl_cnt.p, o->sl_offs.p, 0u, n_gen, rocprim::plus<uint32_t>(), stream()) != hipSuccess)
                fail(AE_ERR_NO_DEVICE, "rocprim exclusive_scan failed");
        }
        uint32_t last[2];
        AE_HIP(hipMemcpyAsync(&last[0], o->sl_offs.p + (n_gen - 1), 4, hipMemcpyDeviceToHost, stream()));
        AE_HIP(hipMemcpyAsync(&last[1], o->sl_cnt.p + (n_gen - 1), 4, hipMemcpyDeviceToHost, stream()));
        sync();
        const double t_cnt = wall();
        const uint32_t total = last[0] + last[1];
        if (total > ev_cap) fail(AE_ERR_STATE, "AE_CE_SLICED: more events than the 8-sigma capacity");
        hipLaunchKernelGGL(sl_fill_kernel, dim3(blocks_for(n_gen, 256)), dim3(256), 0, stream(), o->dev, n_gen, key, (const uint32_t*)o->sl_cnt.p,
                           (const uint32_t*)o->sl_offs.p, n_slices, gen_erec, gen_color,
                           slice_keys ? (const uint8_t*)nullptr : (const uint8_t*)o->sl_class_pos.p, classes, spread, (float)(seg_samples / (double)n / (double)n_slices),
                           ov_every, o->sl_keys0.p, ev0);
        if (debug_knob("AE_SL_CHECK_FILL")) {   // (test) the same fill with every edge's slices in order, both sorted: word for word the same events and keys
            DevBuf<uint32_t> k2, k3, v2, v3;
            k2.alloc_pooled(total + 1); k3.alloc_pooled(total + 1); v2.alloc_pooled(3ull * total + 3); v3.alloc_pooled(3ull * total + 3);
            hipLaunchKernelGGL(sl_fill_kernel, dim3(blocks_for(n_gen, 256)), dim3(256), 0, stream(), o->dev, n_gen, key, (const uint32_t*)o->sl_cnt.p,
                               (const uint32_t*)o->sl_offs.p, n_slices, gen_erec, gen_color,
                               slice_keys ? (const uint8_t*)nullptr : (const uint8_t*)o->sl_class_pos.p, classes, spread | 4, (float)(seg_samples / (double)n / (double)n_slices),
                               ov_every, k2.p, reinterpret_cast<Event*>(v2.p));
            const bool ref_second = sort_events(o, k2.p, k3.p, reinterpret_cast<Event*>(v2.p), reinterpret_cast<Event*>(v3.p), total, kbegin, kbits);
            DevBuf<uint32_t> k0c, v0c, k1c, v1c;   // (copies: the batch's own sort follows on the originals)
            k0c.alloc_pooled(total + 1); k1c.alloc_pooled(total + 1); v0c.alloc_pooled(3ull * total + 3); v1c.alloc_pooled(3ull * total + 3);
            AE_HIP(hipMemcpyAsync(k0c.p, o->sl_keys0.p, 4ull * total, hipMemcpyDeviceToDevice, stream()));
            AE_HIP(hipMemcpyAsync(v0c.p, ev0, 12ull * total, hipMemcpyDeviceToDevice, stream()));
            const bool got_second = sort_events(o, k0c.p, k1c.p, reinterpret_cast<Event*>(v0c.p), reinterpret_cast<Event*>(v1c.p), total, kbegin, kbits);
            DevBuf<unsigned long long> bad;
            bad.alloc_pooled(2);
            bad.zero();
            hipLaunchKernelGGL(sl_diff_words_kernel, dim3(1024), dim3(256), 0, stream(), (const uint32_t*)(got_second ? v1c.p : v0c.p), (const uint32_t*)(ref_second ? v3.p : v2.p), 3ull * total, bad.p);
            hipLaunchKernelGGL(sl_diff_words_kernel, dim3(1024), dim3(256), 0, stream(), (const uint32_t*)(got_second ? k1c.p : k0c.p), (const uint32_t*)(ref_second ? k3.p : k2.p), (uint64_t)total, bad.p + 1);
            const std::vector<unsigned long long> hb = bad.to_host();
            if (hb[0] || hb[1]) {
                char msg[200];
                snprintf(msg, sizeof msg, "AE_SL_CHECK_FILL: %llu event words and %llu keys differ from the fill with every edge's slices in order (%u events)", hb[0], hb[1], total);
                fail(AE_ERR_STATE, msg);
            }
        }
        if (prof) sync();
        const double t_fill = wall();
        const bool in_second = sort_events(o, o->sl_keys0.p, o->sl_keys1.p, ev0, ev1, total, kbegin, kbits);
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
        if (merged) {
            step_seq_base = step_seq;
            step_seq += n_slices * classes;
            if (window) o->sl_class_done.zero();
        }
        if (chains_dbg > 1) {   // fork: the side streams start behind the event generation
            AE_HIP(hipEventRecord(dbg_ev, stream()));
            for (int pc = 0; pc < chains_dbg; pc++) AE_HIP(hipStreamWaitEvent(dbg_streams[pc], dbg_ev, 0));
        }
        for (uint32_t s = 0; s < n_slices; s++) {
            // step pointers of the slice.  Merged: sp[q] = the class at POSITION q, sp[classes] the overflow class.  One launch per class:
            // the overflow class FIRST (key 0: the order its edges are generated in), class c at sp[1 + c]
            const uint32_t* sp = hptr.data() + (size_t)s * kstride;
            const uint32_t* spc = slice_keys ? sp + 1 : sp;                       // spc[c] .. spc[c + 1]: class c (one launch per class) / position c (merged)
            const uint32_t ov0 = slice_keys ? sp[0] : sp[classes], ov1 = slice_keys ? sp[1] : sp[classes + 1u];   // the overflow class's events
            const uint8_t* order = class_pos.data() + (size_t)s * std::max(1u, classes);   // (one launch per class) the slice's class order: a fresh uniform permutation
            if (neg_snapshot && s % (uint32_t)neg_snapshot == 0u) {   // (experiment: the negatives of the next `neg_snapshot` slices are read from the rows as they are NOW)
                AE_HIP(hipMemcpyAsync(o->sl_neg_snap.p, cdev.y, sizeof(float) * n * ystride, hipMemcpyDeviceToDevice, stream()));
                da.c.yneg = a.c.yneg = o->sl_neg_snap.p;
            }
            if (merged) {
                // every class of the slice in ONE launch, the order between events kept node by node (sl_slice_kernel).  Its dependency words
                // were filled by the slice before it, from inside its kernel (a wave that is through enters the next slice's events: the
                // atomics run beside the other waves' work); only a slice without such a predecessor takes the pass of its own (29 us of
                // random atomics at 344 k events).  (The same pass on a side stream bought nothing: 50.8 against 50.0 ms -- its workgroups
                // wait for slots the slice's own hold.)
                if (sp[classes] > sp[0]) {
                    SliceRunArgs ra = slice_args(s);
                    unsigned grid = 0;
                    for (uint32_t q = 0; q < classes; q++) grid += (sp[q + 1] - sp[q] + 255u) / 256u;
                    if (!premarked) hipLaunchKernelGGL(sl_dep_mark_kernel, dim3(blocks_for(sp[classes] - sp[0], 256)), dim3(256), 0, stream(), ra);
                    premarked = s + 1u < n_slices && sp[2u * classes + 1u] > sp[classes + 1u] && !debug_knob("AE_SL_NO_PREMARK");
                    if (premarked) ra.next_sptr = ra.sptr + (classes + 1u);
                    const bool tile_run = use_tile && !y_in_cache && (uint64_t)(sp[classes] - sp[0]) >= tile_min_events;   // (a workgroup's tile serves its 256 events whatever their class)
                    AE_DISPATCH_DIM(o->dev.dim, launch_slice, ra, grid, o->sl_srec_floats, f64, tile_run);
                }
            } else if (chains_dbg > 1) {
                // TIMING EXPERIMENT (AE_SL_CHAINS_DBG = P, wrong results): every step cut into P launches on P streams that are never joined
                // inside the batch -- what P desynchronised chains of steps would cost if the graph fell into P independent parts
                for (uint32_t qq = 0; qq < classes; qq++) {
                    const uint32_t q = slice_keys ? (uint32_t)order[qq] : qq;
                    if (spc[q + 1] == spc[q]) continue;
                    const uint32_t b0 = spc[q], cnt = spc[q + 1] - spc[q];
                    da.step_seq = step_seq++;
                    for (int pc = 0; pc < chains_dbg; pc++) {
                        da.begin = b0 + (uint32_t)(((uint64_t)cnt * pc / chains_dbg) & ~63ull);
                        da.end = pc + 1 == chains_dbg ? b0 + cnt : b0 + (uint32_t)(((uint64_t)cnt * (pc + 1) / chains_dbg) & ~63ull);
                        if (da.end <= da.begin) continue;
                        da.ept = 1;
                        da.tile = (use_tile && !y_in_cache && da.end - da.begin >= tile_min_events) ? 1 : 0;
                        StreamScope sc(dbg_streams[pc]);
                        if (line) launch_step_line((uint32_t)o->dev.dim, da, line, f64, da.tile != 0, nullptr);
                        else AE_DISPATCH_DIM(o->dev.dim, launch_direct, da, o->sl_srec_floats, f64);
                    }
                }
            } else
            for (uint32_t qq = 0; qq < classes; qq++) {  // the slice's matchings, in this slice's order
                const uint32_t q = slice_keys ? (uint32_t)order[qq] : qq;
                if (spc[q + 1] == spc[q]) continue;
                da.begin = spc[q];
                da.end = spc[q + 1];
                const uint32_t cnt = da.end - da.begin;
                // events per thread: one, unless the step is several times what the device holds at once (then the tile is amortised)
                da.ept = ept_force ? ept_force : std::min(4u, std::max(1u, cnt / (256u * 3072u)));
                da.tile = (use_tile && !y_in_cache && cnt >= tile_min_events) ? 1 : 0;
                da.step_seq = step_seq++;
                if (line) launch_step_line((uint32_t)o->dev.dim, da, line, f64, da.tile != 0, nullptr);
                else AE_DISPATCH_DIM(o->dev.dim, launch_direct, da, o->sl_srec_floats, f64);
            }
            if (!has_overflow || (ov_every > 1u && s != std::min(n_slices - 1u, s - s % ov_every + ov_every / 2u))) { exchange_after(s); continue; }
            a.f0 = ov0;
            a.f1 = ov1;
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
            exchange_after(s);
        }
        if (chains_dbg > 1) {   // join
            for (int pc = 0; pc < chains_dbg; pc++) {
                AE_HIP(hipEventRecord(dbg_ev, dbg_streams[pc]));
                AE_HIP(hipStreamWaitEvent(stream(), dbg_ev, 0));
            }
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
        if (exchanges) exchange_now();   // (after the drain: the segment's last exchange)
        t_drain += wall() - t_enq;
    }
    owe.armed = false;
    if (own_copy) move_rows(0, n, 0, 0, (const float*)(o->sl_y.p + row_at), o->dev.y, 0);   // back to the caller's labels and row stride
    if (prof) fprintf(stderr, "CESLICE batch %u: event generation %.1f ms, slices enqueued in %.1f ms, first look + drain %.1f ms (%d looks), total %.1f ms\n", iter,
                      t_evgen * 1e3, t_enqueue * 1e3, t_drain * 1e3, drain_iterations, (wall() - t_begin) * 1e3);
    check_launch("ce_slice");
    std::vector<unsigned long long> h = o->sl_done.to_host();
    o->sl_done.zero();
    if (h[1024]) {
        o->sl_counts.zero();
        sync();
        if (h[1024] & 2ull) fail(AE_ERR_STATE, "AE_CE_SLICED: a hub chain waited for the previous chunk beyond the poll budget (is another process using this GPU?)");
        if (h[1024] & (unsigned long long)kErrDepPoll)
            fail(AE_ERR_STATE, "AE_CE_SLICED: an event of a merged slice waited for an earlier class on one of its nodes beyond the poll budget (is another process using this GPU?)");
        if (h[1024] & (unsigned long long)kErrWindowPoll)
            fail(AE_ERR_STATE, "AE_CE_SLICED: a workgroup of a merged slice waited for the classes before its window beyond the poll budget (is another process using this GPU?)");
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
