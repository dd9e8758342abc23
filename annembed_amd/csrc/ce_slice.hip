// ce_slice.hip -- AE_CE_SLICED: the CE gradient batch (gradient_iteration_threaded, src/embedder.rs:1311-1315) as an
// OPTIMISTIC, TIME-SLICED execution -- the faithful mode for graphs of any size (throughput-bound, no resident-lane limit).
//
// What it keeps of the reference: every sample is applied to the CURRENT rows of both its end points with one
// gradient (embedder.rs:1228-1239), the samples come in an i.i.d. order, the reference's f64 scalars; the five negatives are
// read as the memory system has them (at most one pass old).  What it gives up: reproducibility sample by sample.
//
// How.  The i.i.d. edge draws of a batch are a Poisson process per edge (as in ce_event.hip): edge e fires c_e ~ Poisson(mu_e)
// times at i.i.d. uniform times.  The events of a batch are generated edge by edge (count -> scan -> fill), their times cut into
// thin SLICES (about half an event per node and slice) and the events bucketed by slice with one radix sort.  Inside a slice the
// order of the events is exchangeable, so a slice is executed optimistically in a few PASSES (one launch each):
//   * every pending event marks its two rows in an owner array with its own id (plain stores: the last writer wins);
//   * the events that find their id on BOTH rows run -- no other running event touches those rows, so one lane reads y_i and
//     y_j, applies the attraction to both and the five repulsions to y_i exactly as embedder.rs:1207-1301, and writes both;
//   * the others are deferred to the next pass (later passes: with probability 1/2 per pass, which breaks repeating
//     stand-offs between events that each hold one row of the other), what is left after the slice's passes joins the next
//     slice, the batch ends with passes until nothing is pending.
// Conflicts are rare by construction (a node has >= 2 events in a slice with probability ~0.09), hubs serialise as they do in
// the reference (one event per pass).
#include "ce_node_common.h"
#include "ce_sample_math.h"

#include <rocprim/rocprim.hpp>

using namespace ae;

namespace ae {
void sort_pairs_u32_u32(uint32_t* d_keys_in, uint32_t* d_keys_out, uint32_t* d_vals_in, uint32_t* d_vals_out, uint64_t count, unsigned end_bit);
}

namespace {

constexpr uint32_t kTagSlCount = 0xFFFF0031u, kTagSlTime = 0xFFFF0032u, kTagSlNeg = 0xFFFF0033u, kTagSlCoin = 0xFFFF0034u;
constexpr uint32_t kNoOwner = 0xFFFFFFFFu;
// the pending list is kept as kSub sub-lists with a counter each: appends (one atomic per WORKGROUP) spread over kSub addresses --
// one counter serialises at ~12 ns per atomic, which with one atomic per wave was 2/3 of a pass at the C3 shape
constexpr int kSub = 16;

struct EdgeRec {       // per edge, 16 bytes: everything a sample needs to know about its edge
    uint32_t j;        // target
    float w;           // probability
    float s_src;       // embedded scale of the source
    uint32_t src;      // source
};
struct Pending {       // a pending event: 16 bytes, read and written coalesced
    uint32_t idx, i, j;
    float w;
};

struct SliceArgs {
    CeDev c;
    const EdgeRec* erec;        // per edge: target, probability
    const uint32_t* ev_edge;    // the batch segment's events sorted by slice: edge ids
    const uint32_t* sptr;       // slice s = events [sptr[s], sptr[s + 1])
    uint32_t* owner;            // [2][n]
    Pending* lists;             // [3][kSub][cap]: pending events, in kSub independent sub-lists (cap entries each)
    float* list_scale;          // [3][kSub][cap]: the source's embedded scale of every pending event
    int tile;                   // 1: the negatives of this pass are drawn from a tile of rows staged in LDS (see sl_exec_kernel)
    uint32_t* counts;           // [3][kSub]
    uint64_t cap;
    uint32_t slice;
    uint32_t key;               // (batch << 12) | segment
    uint32_t pass_seq;          // running pass number of the batch (RNG key of the back-off coin)
    int src_list, dst_list, zero_list, owner_chk, owner_mark;
    int backoff;                // 1: a deferred event marks only with probability 1/2
    double step;
    unsigned long long* done_counter;   // [1024] spread counters of executed samples; [1024] = overflow flag
};


// events per edge of this segment
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
// (slice id, edge id) of every event, at the edge's offset
__global__ void __launch_bounds__(256) sl_fill_kernel(CeDev c, uint32_t key, const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ offs,
                                                      uint32_t n_slices, uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    const uint64_t e = blockIdx.x * 256ull + threadIdx.x;
    if (e >= c.nnz) return;
    const uint32_t tk = pcg_hash(round_hash_key(key, c.seed) ^ kTagSlTime);
    const uint32_t k = cnt[e], o = offs[e];
    for (uint32_t r = 0; r < k; r++) {
        keys[o + r] = __umulhi(pcg_hash((pcg_hash((uint32_t)e) + r * 0x9E3779B9u) ^ tk), n_slices);
        vals[o + r] = (uint32_t)e;
    }
}
__global__ void sl_sptr_kernel(const uint32_t* __restrict__ keys, uint32_t total, uint32_t n_slices, uint32_t* __restrict__ sptr) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s > n_slices + 1) return;  // two entries past the last slice: an empty range for the drain passes
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

// start of a slice: pending list = what the previous slice left + the slice's own events; every one marks its two rows.
// Sub-list s (blockIdx.y) takes the leftover sub-list s and every kSub-th event of the slice.
__global__ void __launch_bounds__(256) sl_mark_kernel(SliceArgs a) {
    const uint32_t sub = blockIdx.y;
    const uint32_t left = a.counts[a.src_list * kSub + sub];
    const uint32_t f0 = a.sptr[a.slice], f1 = a.sptr[a.slice + 1];
    const uint32_t fresh = f1 > f0 + sub ? (f1 - f0 - sub + (uint32_t)kSub - 1u) / (uint32_t)kSub : 0u;
    const uint32_t total = left + fresh;
    const uint64_t so = ((uint64_t)a.src_list * kSub + sub) * a.cap, dof = ((uint64_t)a.dst_list * kSub + sub) * a.cap;
    for (uint64_t t = blockIdx.x * 256ull + threadIdx.x; t < total; t += (uint64_t)gridDim.x * 256ull) {
        Pending p;
        float sc;
        if (t < left) {
            p = a.lists[so + t];
            sc = a.list_scale[so + t];
        } else {
            p.idx = f0 + sub + (uint32_t)(t - left) * (uint32_t)kSub;
            const EdgeRec er = a.erec[a.ev_edge[p.idx]];
            p.i = er.src; p.j = er.j; p.w = er.w;
            sc = er.s_src;
        }
        if (t < a.cap) { a.lists[dof + t] = p; a.list_scale[dof + t] = sc; }
        a.owner[(uint64_t)a.owner_mark * a.c.n + p.i] = p.idx;
        a.owner[(uint64_t)a.owner_mark * a.c.n + p.j] = p.idx;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        a.counts[a.dst_list * kSub + sub] = total < a.cap ? total : (uint32_t)a.cap;  // (cap is sized so that this never truncates; flagged otherwise)
        if (total > a.cap) atomicOr(reinterpret_cast<unsigned int*>(a.done_counter + 1024), 1u);
        a.counts[a.zero_list * kSub + sub] = 0;
    }
}

// one pass: the pending events that own both their rows run, the others go to the next list and mark for the next pass
template <int DIM, int KMAX>
__global__ void __launch_bounds__(256) sl_exec_kernel(SliceArgs a) {
    // tile of coordinate rows for the negatives of a crowded pass (a.tile): kW windows of kL consecutive rows each, window starts
    // uniform over the nodes (wrapping) and fresh per workgroup and pass, staged in LDS with coalesced loads.  A negative is then
    // "window uniform, row uniform": every node has the same probability 1/n, as in embedder.rs:1121; the rows are as fresh as
    // the pass (the launch started after every earlier pass's writes).  What differs from the reference: the negatives of the
    // ~150 samples a workgroup runs in a pass come from the same kW windows (the marginals are exact, the joint law is not).
    constexpr int kW = 8, kL = DIM <= 8 ? 128 : 64, kTile = kW * kL;
    __shared__ __attribute__((aligned(16))) float s_tile[kTile * DIM];
    __shared__ uint32_t s_wstart[kW];
    __shared__ uint32_t s_wave_cnt[4], s_base;
    const CeDev c = a.c;
    const uint32_t sub = blockIdx.y, dsub = (blockIdx.x + blockIdx.y) % (uint32_t)kSub;
    const uint32_t total = a.counts[a.src_list * kSub + sub];
    const uint64_t so = ((uint64_t)a.src_list * kSub + sub) * a.cap, dof = ((uint64_t)a.dst_list * kSub + dsub) * a.cap;
    const uint32_t* own_chk = a.owner + (uint64_t)a.owner_chk * c.n;
    uint32_t* own_mark = a.owner + (uint64_t)a.owner_mark * c.n;
    const bool hub = c.hub_odds != nullptr;
    const bool tile = a.tile && !hub && (uint64_t)blockIdx.x * 256ull < total && c.n > (uint64_t)kTile * 4ull;
    if (tile) {
        if (threadIdx.x < kW)
            s_wstart[threadIdx.x] = __umulhi(pcg_hash(pcg_hash(a.key ^ kTagSlNeg) + a.pass_seq * 0x9E3779B9u + (blockIdx.x * (uint32_t)kSub + sub) * 8u + threadIdx.x), (uint32_t)c.n);
        __syncthreads();
        for (int r = threadIdx.x; r < kTile; r += 256) {
            uint64_t node = (uint64_t)s_wstart[r / kL] + (uint64_t)(r % kL);
            node -= node >= c.n ? c.n : 0ull;
            float row[DIM];
            load_row<DIM>(c.y, (uint32_t)node, row);
#pragma unroll
            for (int t = 0; t < DIM; t++) s_tile[r * DIM + t] = row[t];
        }
        __syncthreads();
    }
    unsigned long long done = 0;
    for (uint64_t t0 = blockIdx.x * 256ull; t0 < total; t0 += (uint64_t)gridDim.x * 256ull) {
        const uint64_t t = t0 + threadIdx.x;
        const bool have = t < total;
        Pending p{0, 0, 0, 0.f};
        float sc = 1.f;
        bool win = false;
        if (have) {
            p = a.lists[so + t];
            sc = a.list_scale[so + t];
            win = own_chk[p.i] == p.idx && own_chk[p.j] == p.idx;
        }
        const uint32_t i = p.i, idx = p.idx;
        if (win) {
            // everything that depends only on (i, j): both rows, the neighbour row of i -- in flight together.  Plain (cached)
            // vector loads: a pass is a launch of its own, everything earlier passes wrote is visible, and the rows this event
            // owns are touched by nobody else during the pass
            float yi[DIM], yj[DIM], grad[DIM];
            load_row<DIM>(c.y, i, yi);
            load_row<DIM>(c.y, p.j, yj);
            const double scale = (double)sc;
            uint64_t ib;
            uint32_t k;
            if (c.uniform_k) { ib = (uint64_t)i * c.uniform_k; k = c.uniform_k; }
            else { ib = c.indptr[i]; k = (uint32_t)(c.indptr[i + 1] - ib); }
            uint32_t nbr_reg[KMAX];
#pragma unroll
            for (int m = 0; m < KMAX; m++) nbr_reg[m] = c.nbr[ib + ((uint32_t)m < k ? (uint32_t)m : k - 1u)];
#pragma unroll
            for (int m = 0; m < KMAX; m++) nbr_reg[m] = (uint32_t)m < k ? nbr_reg[m] : 0xFFFFFFFFu;  // the pad never equals a candidate
            // the five negatives (embedder.rs:1241-1253): uniform / NodeSampler (:927-930) draws, rejected when k = i, k = j or
            // k in N(i) (nodeparam.rs:83-85; j is in N(i)); eight candidates at a time so that the alias look-ups overlap
            uint32_t kk[5] = {i, i, i, i, i}, kr[5] = {0, 0, 0, 0, 0};
            uint32_t got = 0;
            const uint32_t nb = pcg_hash(pcg_hash((uint32_t)c.seed ^ a.key ^ kTagSlNeg) + idx);
            for (uint32_t round = 0; round < 8u && got < 5u; round++) {
                uint32_t cand[8], crow[8];
                if (tile) {
#pragma unroll
                    for (int z = 0; z < 8; z++) {
                        const uint32_t w0 = pcg_hash(nb + (round * 8u + (uint32_t)z) * 0x9E3779B9u);
                        const uint32_t wi = w0 >> 29, off = __umulhi(pcg_hash(w0 ^ 0x85EBCA6Bu), (uint32_t)kL);
                        uint64_t node = (uint64_t)s_wstart[wi] + off;
                        node -= node >= c.n ? c.n : 0ull;
                        cand[z] = (uint32_t)node;
                        crow[z] = wi * (uint32_t)kL + off;
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
                    for (int z = 0; z < 8; z++) { cand[z] = (uu[z] < od[z]) ? xs[z] : al[z]; crow[z] = 0; }
                } else {
#pragma unroll
                    for (int z = 0; z < 8; z++) { cand[z] = __umulhi(pcg_hash(nb + (round * 8u + (uint32_t)z) * 0x9E3779B9u), (uint32_t)c.n); crow[z] = 0; }  // :1121
                }
#pragma unroll
                for (int z = 0; z < 8; z++) {
                    uint32_t acc = cand[z] ^ i;
#pragma unroll
                    for (int m = 0; m < KMAX; m++) { const uint32_t x = nbr_reg[m] ^ cand[z]; acc = x < acc ? x : acc; }
                    const bool ok = acc != 0u && got < 5u;
#pragma unroll
                    for (int g = 0; g < 5; g++) {  // (static indexing keeps kk / kr in registers)
                        kk[g] = (ok && got == (uint32_t)g) ? cand[z] : kk[g];
                        kr[g] = (ok && got == (uint32_t)g) ? crow[z] : kr[g];
                    }
                    got += ok ? 1u : 0u;
                }
            }
            float nrow[5][DIM];
            if (tile) {
#pragma unroll
                for (int g = 0; g < 5; g++)
#pragma unroll
                    for (int t2 = 0; t2 < DIM; t2++) nrow[g][t2] = s_tile[kr[g] * DIM + t2];
            } else {
#pragma unroll
                for (int g = 0; g < 5; g++) load_row<DIM>(c.y, kk[g], nrow[g]);  // (may be rewritten during this pass by its owner: at most one pass old)
            }
            sample_attract<DIM>(yi, yj, grad, p.w, scale, c.b, a.step);  // :1207-1238: one gradient, both ends
            store_row<DIM>(c.y, p.j, yj);                                 // :1239
#pragma unroll
            for (int g = 0; g < 5; g++)
                if ((uint32_t)g < got) sample_repulse<DIM>(yi, nrow[g], grad, scale, c.b, a.step);  // :1267-1297
            store_row<DIM>(c.y, i, yi);                                   // :1301
            done++;
        }
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
            if (pos < a.cap) { a.lists[dof + pos] = p; a.list_scale[dof + pos] = sc; }
            else atomicOr(reinterpret_cast<unsigned int*>(a.done_counter + 1024), 1u);
            const bool mark = !a.backoff || (pcg_hash(idx ^ pcg_hash(a.pass_seq ^ kTagSlCoin)) & 1u);
            if (mark) {
                own_mark[i] = idx;
                own_mark[p.j] = idx;
            }
        }
        __syncthreads();  // s_wave_cnt / s_base are reused by the next trip
    }
    for (int off = 32; off > 0; off >>= 1) done += __shfl_xor(done, off);
    if ((threadIdx.x & 63) == 0 && done) atomicAdd(&a.done_counter[(blockIdx.x * 4u + (threadIdx.x >> 6) + blockIdx.y * 64u) & 1023u], done);
    if (blockIdx.x == 0 && threadIdx.x == 0) a.counts[a.zero_list * kSub + sub] = 0;
}

template <int DIM>
void launch_exec(const SliceArgs& a, unsigned grid, uint32_t max_nbng) {
    if constexpr (DIM > 0) {
        if (max_nbng <= 8) hipLaunchKernelGGL((sl_exec_kernel<DIM, 8>), dim3(grid, kSub), dim3(256), 0, stream(), a);
        else if (max_nbng <= 16) hipLaunchKernelGGL((sl_exec_kernel<DIM, 16>), dim3(grid, kSub), dim3(256), 0, stream(), a);
        else hipLaunchKernelGGL((sl_exec_kernel<DIM, 32>), dim3(grid, kSub), dim3(256), 0, stream(), a);
    }
}

}  // namespace

namespace ae {

const char* ce_slice_unsupported(const ae_entropy_optim* o) {
    const uint32_t d = o->dev.dim;
    if (!(d == 2 || d == 3 || d == 4 || d == 8 || d == 16)) return "asked_dim must be one of 2, 3, 4, 8, 16";
    if (o->dev.node_lo != 0 || o->dev.node_hi != o->dev.n) return "a sharded node range (both rows of a sample must be on the device)";
    if (o->dev.nnz >= 0xFFFFFFFFull) return "more than 2^32 edges";
    if (o->g->max_nbng > 32) return "rows of more than 32 neighbours";
    return nullptr;
}

void ce_slice_prepare(ae_entropy_optim* o) {
    const ae_kgraph* g = o->g;
    o->sl_erec.alloc(g->nnz * 4);  // EdgeRec as four words
    hipLaunchKernelGGL(sl_edge_rec_kernel, dim3(blocks_for(g->n, 256)), dim3(256), 0, stream(), o->dev, reinterpret_cast<EdgeRec*>(o->sl_erec.p));
    check_launch("sl_prepare");
    // largest edge probability (segments keep the per-edge Poisson mean below 16)
    std::vector<float> hp = o->np->proba.to_host();
    float pmax = 0.f;
    for (float v : hp) pmax = std::max(pmax, v);
    o->sl_pmax = pmax;
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
    // segments of the batch: per-edge Poisson mean <= 16 (f32 inversion), at most 2^30 events per segment
    uint32_t segments = (uint32_t)std::max(1.0, std::ceil(per_node * (double)o->sl_pmax / 16.0));
    segments = std::max(segments, (uint32_t)(nb_sample / (1ull << 30)) + 1u);
    if (iter >= (1u << 20) || segments >= 4096) fail(AE_ERR_INVALID_ARG, "AE_CE_SLICED: batch / segment index too large for the RNG key");
    const double seg_samples = (double)nb_sample / segments;
    // slices of a segment: about half an event per node and slice (a sample is an event at two nodes)
    const double lambda_s = debug_knob("AE_SL_LAMBDA") ? atof(debug_knob("AE_SL_LAMBDA")) : 0.5;
    const uint32_t n_slices = (uint32_t)std::max(1.0, std::ceil(2.0 * seg_samples / (double)n / lambda_s));
    const int passes = debug_knob("AE_SL_PASSES") ? atoi(debug_knob("AE_SL_PASSES")) : 3;
    o->rounds = segments * n_slices;
    const uint64_t ev_cap = (uint64_t)(seg_samples + 8.0 * std::sqrt(seg_samples) + 1024.0);
    // pending-list capacity: a slice's events (+ 8 sigma) plus what hubs may accumulate
    const double per_slice = seg_samples / n_slices;
    const uint64_t cap = (uint64_t)((4.0 * per_slice + 16.0 * std::sqrt(per_slice)) / kSub + 8192.0);  // per sub-list
    if (o->sl_cnt.n < nnz) { o->sl_cnt.alloc(nnz); o->sl_offs.alloc(nnz); }
    if (o->sl_keys0.n < ev_cap) { o->sl_keys0.alloc(ev_cap); o->sl_keys1.alloc(ev_cap); o->sl_vals0.alloc(ev_cap); o->sl_vals1.alloc(ev_cap); }
    if (o->sl_sptr.n < (uint64_t)n_slices + 2) o->sl_sptr.alloc((uint64_t)n_slices + 2);
    if (o->sl_lists.n < 3 * (uint64_t)kSub * cap * 4) { o->sl_lists.alloc(3 * (uint64_t)kSub * cap * 4); o->sl_list_scale.alloc(3 * (uint64_t)kSub * cap); }
    unsigned sbits = 1;
    while (sbits < 32 && (n_slices >> sbits)) sbits++;
    SliceArgs a;
    a.c = o->dev;
    a.erec = reinterpret_cast<const EdgeRec*>(o->sl_erec.p);
    a.owner = o->sl_owner.p;
    a.lists = reinterpret_cast<Pending*>(o->sl_lists.p);
    a.list_scale = o->sl_list_scale.p;
    const bool use_tile = !debug_knob("AE_SL_NO_TILE");
    a.counts = o->sl_counts.p;
    a.cap = cap;
    a.step = grad_step;
    a.done_counter = o->sl_done.p;
    const unsigned grid_full = (unsigned)std::min<uint64_t>(blocks_for((uint64_t)(per_slice * 1.5 / kSub) + 512, 256), 65535u);  // per sub-list
    uint32_t pass_seq = 0;
    int cur = 0;  // list that holds what is pending
    o->sl_counts.zero();
    for (uint32_t sg = 0; sg < segments; sg++) {
        const uint32_t key = (iter << 12) | sg;
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
        const uint32_t total = last[0] + last[1];
        if (total > ev_cap) fail(AE_ERR_STATE, "AE_CE_SLICED: more events than the 8-sigma capacity");
        hipLaunchKernelGGL(sl_fill_kernel, dim3(blocks_for(nnz, 256)), dim3(256), 0, stream(), o->dev, key, (const uint32_t*)o->sl_cnt.p,
                           (const uint32_t*)o->sl_offs.p, n_slices, o->sl_keys0.p, o->sl_vals0.p);
        sort_pairs_u32_u32(o->sl_keys0.p, o->sl_keys1.p, o->sl_vals0.p, o->sl_vals1.p, total, sbits);
        hipLaunchKernelGGL(sl_sptr_kernel, dim3(blocks_for((uint64_t)n_slices + 2, 256)), dim3(256), 0, stream(), (const uint32_t*)o->sl_keys1.p, total, n_slices,
                           o->sl_sptr.p);
        check_launch("sl_events");
        a.ev_edge = o->sl_vals1.p;
        a.sptr = o->sl_sptr.p;
        a.key = key;
        // NOTE: event indices are positions in this segment's sorted array; what is still pending when a segment ends is
        // finished (the drain below) before the next segment reuses the arrays
        for (uint32_t s = 0; s < n_slices; s++) {
            a.slice = s;
            a.src_list = cur; a.dst_list = (cur + 1) % 3; a.zero_list = (cur + 2) % 3;
            a.owner_mark = 0;
            hipLaunchKernelGGL(sl_mark_kernel, dim3(grid_full, kSub), dim3(256), 0, stream(), a);
            cur = (cur + 1) % 3;
            for (int p = 0; p < passes; p++) {
                a.src_list = cur; a.dst_list = (cur + 1) % 3; a.zero_list = (cur + 2) % 3;
                a.owner_chk = p & 1; a.owner_mark = (p + 1) & 1;
                a.backoff = p >= 1;
                a.tile = (p == 0 && use_tile) ? 1 : 0;
                a.pass_seq = pass_seq++;
                const unsigned grid = p == 0 ? grid_full : std::max(4u, grid_full >> (2 * p));
                AE_DISPATCH_DIM(o->dev.dim, launch_exec, a, grid, o->g->max_nbng);
                cur = (cur + 1) % 3;
            }
            // the next slice's mark kernel marks in owner[0]; the last pass above marked in owner[passes & 1]: the mark
            // kernel re-marks everything that is pending anyway
        }
        // drain: passes until nothing is pending (a look at the counter every 8 passes)
        for (int guard = 0; guard < 100000; guard++) {
            uint32_t lefts[kSub];
            AE_HIP(hipMemcpyAsync(lefts, o->sl_counts.p + cur * kSub, 4 * kSub, hipMemcpyDeviceToHost, stream()));
            sync();
            uint64_t left = 0;
            for (int q = 0; q < kSub; q++) left += lefts[q];
            if (!left) break;
            if (guard == 99999) fail(AE_ERR_STATE, "AE_CE_SLICED: pending events did not drain");
            // re-mark all (owner[0]) then 8 passes
            a.slice = n_slices;  // empty range: sptr[n_slices] == sptr[n_slices + 1] is arranged below
            a.src_list = cur; a.dst_list = (cur + 1) % 3; a.zero_list = (cur + 2) % 3;
            a.owner_mark = 0;
            hipLaunchKernelGGL(sl_mark_kernel, dim3(std::max(4u, grid_full >> 2), kSub), dim3(256), 0, stream(), a);
            cur = (cur + 1) % 3;
            for (int p = 0; p < 8; p++) {
                a.src_list = cur; a.dst_list = (cur + 1) % 3; a.zero_list = (cur + 2) % 3;
                a.owner_chk = p & 1; a.owner_mark = (p + 1) & 1;
                a.backoff = 1;
                a.tile = 0;
                a.pass_seq = pass_seq++;
                AE_DISPATCH_DIM(o->dev.dim, launch_exec, a, std::max(4u, grid_full >> 2), o->g->max_nbng);
                cur = (cur + 1) % 3;
            }
        }
    }
    check_launch("ce_slice");
    std::vector<unsigned long long> h = o->sl_done.to_host();
    if (h[1024]) fail(AE_ERR_STATE, "AE_CE_SLICED: pending list overflow");
    // samples executed, into the common counter
    unsigned long long d = 0;
    for (int q = 0; q < 1024; q++) d += h[q];
    o->sl_done.zero();
    std::vector<unsigned long long> hc = o->sample_counter.to_host();
    hc[0] += d;
    o->sample_counter.upload(hc.data(), hc.size());
    sync();
}

}  // namespace ae
