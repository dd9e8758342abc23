// ce_event.hip -- AE_CE_EVENT: the lock-free CE gradient batch (gradient_iteration_threaded, src/embedder.rs:1311-1315)
// as an EVENT-ORDERED execution: every sample is applied to the CURRENT rows of both its end points, one gradient for
// both (embedder.rs:1228-1239), in an i.i.d. random order -- the reference's sequential semantics -- with no global sort
// and no atomics.
//
// Why.  The reference's Hogwild loop runs tens of samples concurrently over N >> threads nodes: it IS the sequential
// loop up to rare races (its own threaded run lands within ~1 % of the sequential one).  A GPU offers more lanes than
// nodes; any schedule that lets a node's partner move between the read and the write of a sample (rounds of stale rows:
// AE_CE_HOGWILD, ce_node.hip) changes what the loop converges to -- the attraction step is stiff (clipped at -0.49: one
// sample closes 98 % of an edge) and the final edge cross entropy is dominated by exactly those collapsed edges.
// Measured with a host emulation of the schedules (tools/sim): stale-row rounds end at 0.4-0.9x the sequential CE
// whatever the round length, and even stale NEGATIVES (rows a few updates old) shift it by 10-20 %.
//
// How.  The i.i.d. edge draws of a batch (alias draw, :987,:1182) are a Poisson process per edge: edge e fires
// c_e ~ Poisson(mu_e) times at i.i.d. uniform times.  Counts and times are pure functions of (seed, batch, window, edge
// id), so BOTH end points of an edge derive the same events with no communication.  A batch is cut into T windows (one
// launch each).  In a window, lane v owns node v: it lists the events of v's out-edges and in-edges, sorts them by time
// and walks the list in order:
//   * v is the SOURCE i of the event: publish y_i in the event's slot, gather the 5 negatives' rows meanwhile, wait for
//     the gradient g, then y_i -= g and the 5 repulsions (:1241-1299) -- the sample's y_i half;
//   * v is the TARGET j: wait for the published y_i, evaluate the attraction ONCE on (y_i, current y_j) in the
//     reference's f64 arithmetic (:1207-1236), y_j += g, hand g to the source through the slot.
// Both owners hold their rows in registers for the whole window; a row has one writer.  Every lane only ever waits for
// an event that precedes its own next event in ONE global order (time, slot id) and all lanes are resident, so the
// earliest unfinished event can always complete: no deadlock (a poll budget turns a violated invariant into an error).
// The result is a sequentially consistent execution of the reference's loop on an i.i.d. sample order; only the rows of
// the negatives are read without synchronisation (they are at most the partner's current event behind).
#include "ce_node_common.h"
#include "ce_sample_math.h"

using namespace ae;

namespace {

constexpr int kSlotCap = 8;        // draws of one edge per window: both ends clamp the Poisson count to it
constexpr int kPool = 2048;        // event entries of one wave (64 nodes) per window, in LDS
constexpr int kPrivSort = 32;      // segments up to this length are insertion-sorted by their lane
constexpr int kCoopMax = 512;      // longer segments are rank-sorted by the whole wave; this is the limit
constexpr uint32_t kTagEvCount = 0xFFFF0021u, kTagEvTime = 0xFFFF0022u, kTagEvNeg = 0xFFFF0023u;
constexpr uint32_t kErrPool = 16u, kErrPoll = 32u, kErrNeg = 64u;

struct EventArgs {
    CeDev c;
    const uint64_t* tptr;
    const InEdge* tin;
    float* slots;               // [nnz * kSlotCap][2 DIM]: y_i published by the source | gradient handed back by the target
    uint32_t window_key;        // (batch << 12) | window
    float unit;                 // mu_e per window = unit * p_e
    double step;
    unsigned long long* sample_counter;
    unsigned int* err;
    uint32_t poll_budget;
    unsigned long long* prof;
};

__device__ __forceinline__ uint32_t ev_count(uint64_t e, float mu, uint32_t ck) {
    // Poisson(mu) by inversion on the edge-keyed uniform, clamped to the slots an edge owns in a window
    const float u = edge_uniform(e, ck);
    float p = __expf(-mu), cdf = p;
    uint32_t c = 0;
    while (u >= cdf && c < (uint32_t)kSlotCap) {
        c++;
        p *= mu * (1.0f / (float)c);
        cdf += p;
    }
    return c;
}
__device__ __forceinline__ uint32_t ev_time(uint64_t e, uint32_t r, uint32_t tk) { return pcg_hash(pcg_hash((uint32_t)e * (uint32_t)kSlotCap + r) ^ tk); }

template <int DIM>
__device__ __forceinline__ void slot_publish(float* p, const float* in) { df_store_version<DIM>(p, 0, in); }
template <int DIM>
__device__ __forceinline__ void slot_clear(float* p) {
    if constexpr (DIM % 2 == 0) {
#pragma unroll
        for (int q = 0; q < DIM / 2; q++) __hip_atomic_store(reinterpret_cast<uint64_t*>(p) + q, kUnpublished64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
#pragma unroll
        for (int t = 0; t < DIM; t++) __hip_atomic_store(reinterpret_cast<uint32_t*>(p) + t, kUnpublished32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

__device__ __forceinline__ void wave_sync_lds() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

template <int DIM, int KMAX, int U>
__global__ void __launch_bounds__(64) ce_event_window_kernel(EventArgs a) {
    __shared__ uint32_t s_time[kPool], s_q[kPool], s_aux[kPool], s_w[kPool];
    const CeDev c = a.c;
    const int lane = threadIdx.x;
    if (__hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & (kErrPoll | kErrPool)) return;  // a failed window: do not spin again
    const uint64_t local = blockIdx.x * 64ull + (uint64_t)lane;
    const bool valid = local < c.n;
    const uint32_t v = (uint32_t)(valid ? local : c.n - 1);
    uint64_t ib;
    uint32_t k;
    if (c.uniform_k) { ib = (uint64_t)v * c.uniform_k; k = c.uniform_k; }
    else { ib = c.indptr[v]; k = (uint32_t)(c.indptr[v + 1] - ib); }
    uint32_t nbr_reg[KMAX];
    float pr[KMAX];
#pragma unroll
    for (int m = 0; m < KMAX; m++) {
        const uint32_t mm = (uint32_t)m < k ? (uint32_t)m : k - 1u;
        nbr_reg[m] = c.nbr[ib + mm];
        pr[m] = c.proba[ib + mm];
    }
    const uint64_t tb = valid ? a.tptr[v] : 0ull, te = valid ? a.tptr[v + 1] : 0ull;
    float yv[DIM];
    load_row_fresh<DIM>(c.y, v, yv);
    const double scale = (double)c.emb_scale[v];
    const uint32_t hk = pcg_hash(pcg_hash((uint32_t)c.seed ^ 0x5bd1e995u) ^ pcg_hash(a.window_key + (uint32_t)(c.seed >> 32)));
    const uint32_t ck = hk ^ kTagEvCount, tk = pcg_hash(hk ^ kTagEvTime);
    unsigned long long t0 = a.prof ? __builtin_amdgcn_s_memtime() : 0ull;
    // ---- pass 1: how many events does the node have in this window
    uint32_t cnt_out[KMAX];
    uint32_t tot = 0;
#pragma unroll
    for (int m = 0; m < KMAX; m++) {
        const bool has = (uint32_t)m < k && valid;
        nbr_reg[m] = (uint32_t)m < k ? nbr_reg[m] : 0xFFFFFFFFu;  // the pad never equals a candidate
        cnt_out[m] = has ? ev_count(ib + m, a.unit * pr[m], ck) : 0u;
        tot += cnt_out[m];
    }
    for (uint64_t x = tb; x < te; x++) {
        const InEdge rec = a.tin[x];
        tot += ev_count(rec.eid, a.unit * rec.w, ck);
    }
    uint32_t incl = tot;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const uint32_t o = __shfl_up(incl, off); incl += lane >= off ? o : 0u; }
    const uint32_t wave_total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    uint32_t tmax = tot;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const uint32_t o = __shfl_xor(tmax, off); tmax = o > tmax ? o : tmax; }
    if (wave_total > (uint32_t)kPool || tmax > (uint32_t)kCoopMax) {  // sized by the host with a wide margin: an error, not a path
        if (lane == 0) atomicOr(a.err, kErrPool);
        return;
    }
    const uint32_t seg = incl - tot;
    // ---- pass 2: the events, unsorted
    {
        uint32_t pos = seg;
#pragma unroll
        for (int m = 0; m < KMAX; m++) {
            for (uint32_t r = 0; r < cnt_out[m]; r++) {
                s_time[pos] = ev_time(ib + m, r, tk);
                s_q[pos] = (uint32_t)(ib + m) * (uint32_t)kSlotCap + r;
                s_aux[pos] = nbr_reg[m];
                s_w[pos] = __float_as_uint(pr[m]);
                pos++;
            }
        }
        for (uint64_t x = tb; x < te; x++) {
            const InEdge rec = a.tin[x];
            const uint32_t cn = ev_count(rec.eid, a.unit * rec.w, ck);
            for (uint32_t r = 0; r < cn; r++) {
                s_time[pos] = ev_time(rec.eid, r, tk);
                s_q[pos] = rec.eid * (uint32_t)kSlotCap + r;
                s_aux[pos] = rec.src | 0x80000000u;
                s_w[pos] = __float_as_uint(rec.w);
                pos++;
            }
        }
    }
    // ---- sort every segment by (time, slot id): one global order that all nodes agree on
    if (tot <= (uint32_t)kPrivSort) {
        for (uint32_t i = 1; i < tot; i++) {
            const uint32_t ti = s_time[seg + i], qi = s_q[seg + i], ai = s_aux[seg + i], wi = s_w[seg + i];
            uint32_t j = i;
            while (j > 0) {
                const uint32_t tj = s_time[seg + j - 1], qj = s_q[seg + j - 1];
                if (tj < ti || (tj == ti && qj < qi)) break;
                s_time[seg + j] = tj; s_q[seg + j] = qj; s_aux[seg + j] = s_aux[seg + j - 1]; s_w[seg + j] = s_w[seg + j - 1];
                j--;
            }
            s_time[seg + j] = ti; s_q[seg + j] = qi; s_aux[seg + j] = ai; s_w[seg + j] = wi;
        }
    }
    wave_sync_lds();
    {   // long segments (hubs): rank sort by the whole wave, entries held in registers meanwhile
        unsigned long long todo = __ballot(tot > (uint32_t)kPrivSort);
        while (todo) {
            const int L = __ffsll((long long)todo) - 1;
            todo &= todo - 1ull;
            const uint32_t sL = (uint32_t)__builtin_amdgcn_readlane((int)seg, L), nL = (uint32_t)__builtin_amdgcn_readlane((int)tot, L);
            constexpr int PER = kCoopMax / 64;
            uint32_t et[PER], eq[PER], ea[PER], ew[PER], rank[PER];
#pragma unroll
            for (int h = 0; h < PER; h++) {
                const uint32_t i = (uint32_t)(h * 64 + lane);
                const bool in = i < nL;
                et[h] = in ? s_time[sL + i] : 0u; eq[h] = in ? s_q[sL + i] : 0u; ea[h] = in ? s_aux[sL + i] : 0u; ew[h] = in ? s_w[sL + i] : 0u;
                rank[h] = 0;
            }
            for (uint32_t j = 0; j < nL; j++) {
                const uint32_t tj = s_time[sL + j], qj = s_q[sL + j];
#pragma unroll
                for (int h = 0; h < PER; h++) rank[h] += (tj < et[h] || (tj == et[h] && qj < eq[h])) ? 1u : 0u;
            }
            wave_sync_lds();
#pragma unroll
            for (int h = 0; h < PER; h++) {
                if ((uint32_t)(h * 64 + lane) < nL) {
                    s_time[sL + rank[h]] = et[h]; s_q[sL + rank[h]] = eq[h]; s_aux[sL + rank[h]] = ea[h]; s_w[sL + rank[h]] = ew[h];
                }
            }
            wave_sync_lds();
        }
    }
    unsigned long long t1 = a.prof ? __builtin_amdgcn_s_memtime() : 0ull;
    // ---- walk the list
    const uint32_t node_base = pcg_hash(pcg_hash((uint32_t)c.seed ^ a.window_key ^ kTagEvNeg) + v);
    const bool hub = c.hub_odds != nullptr;
    uint32_t i_ev = 0;
    bool published = false;
    float nrow[5][DIM], grad[DIM];
#pragma unroll
    for (int g = 0; g < 5; g++)
#pragma unroll
        for (int t = 0; t < DIM; t++) nrow[g][t] = 0.f;
    uint32_t negmask = 0;
    uint32_t idle = 0, iters = 0;
    unsigned long long done_src = 0;
    while (true) {
        const bool act = i_ev < tot;
        if (!__any(act)) break;
        iters++;
        const uint32_t e0 = seg + (act ? i_ev : 0u);
        const uint32_t cur_q = s_q[e0], cur_aux = s_aux[e0];
        const bool is_src = act && !(cur_aux >> 31);
        float* slot = a.slots + (uint64_t)cur_q * (uint64_t)(2 * DIM);
        if (is_src && !published) {
            slot_publish<DIM>(slot, yv);
            published = true;
            // the 5 negatives of the sample (embedder.rs:1241-1253): uniform (or NodeSampler, :927-930) draws, rejected when
            // k = i, k = j or k in N(i) (NodeParam::get_edge, nodeparam.rs:83-85); candidates are drawn 8 at a time so that
            // the alias look-ups of the hubness sampler are in flight together
            uint32_t kk[5] = {v, v, v, v, v};
            uint32_t got = 0;
            for (uint32_t round = 0; round < 8u && got < 5u; round++) {
                uint32_t cand[8];
                if (hub) {
                    uint32_t xs[8], al[8];
                    float od[8], uu[8];
#pragma unroll
                    for (int z = 0; z < 8; z++) {
                        const uint32_t w0 = pcg_hash(node_base + i_ev * 64u + round * 8u + (uint32_t)z);
                        xs[z] = __umulhi(w0, (uint32_t)c.n);
                        uu[z] = (float)(pcg_hash(w0 ^ 0x9E3779B9u) >> 8) * (1.0f / 16777216.0f);
                        od[z] = c.hub_odds[xs[z]];
                        al[z] = c.hub_alias[xs[z]];
                    }
#pragma unroll
                    for (int z = 0; z < 8; z++) cand[z] = (uu[z] < od[z]) ? xs[z] : al[z];
                } else {
#pragma unroll
                    for (int z = 0; z < 8; z++) cand[z] = __umulhi(pcg_hash(node_base + i_ev * 64u + round * 8u + (uint32_t)z), (uint32_t)c.n);  // :1121
                }
#pragma unroll
                for (int z = 0; z < 8; z++) {
                    uint32_t acc = cand[z] ^ v;
#pragma unroll
                    for (int m = 0; m < KMAX; m++) { const uint32_t x = nbr_reg[m] ^ cand[z]; acc = x < acc ? x : acc; }  // j is in N(i)
                    const bool ok = acc != 0u && got < 5u;
#pragma unroll
                    for (int g = 0; g < 5; g++) kk[g] = (ok && got == (uint32_t)g) ? cand[z] : kk[g];
                    got += ok ? 1u : 0u;
                }
            }
            negmask = (1u << got) - 1u;
            if (got < 5u) atomicOr(a.err, kErrNeg);  // 64 rejected draws in a row: the graph is too small for 5 negatives (n >= max_nbng + 8 is checked by the host)
#pragma unroll
            for (int g = 0; g < 5; g++) load_row_fresh<DIM>(c.y, kk[g], nrow[g]);
        }
        // poll: the gradient (source) or the partner's row (target); targets look U events ahead
        float in[U][DIM];
        bool ready[U];
        uint32_t la_q[U], la_aux[U];
        float la_w[U], la_su[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            ready[u] = false;
            la_q[u] = cur_q; la_aux[u] = cur_aux; la_w[u] = 0.f; la_su[u] = 1.f;
        }
        if (act) {
            if (is_src) {
                ready[0] = df_try_load_version<DIM>(slot + DIM, 0, in[0]);
            } else {
                bool chain = true;
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const bool have = chain && i_ev + (uint32_t)u < tot;
                    const uint32_t eu = seg + (have ? i_ev + (uint32_t)u : i_ev);
                    la_q[u] = s_q[eu]; la_aux[u] = s_aux[eu]; la_w[u] = __uint_as_float(s_w[eu]);
                    chain = have && (la_aux[u] >> 31);
                    if (chain) {
                        la_su[u] = c.emb_scale[la_aux[u] & 0x7FFFFFFFu];
                        ready[u] = df_try_load_version<DIM>(a.slots + (uint64_t)la_q[u] * (uint64_t)(2 * DIM), 0, in[u]);
                    }
                }
            }
        }
        bool progressed = false;
        if (act && is_src && ready[0]) {
            slot_clear<DIM>(slot + DIM);
#pragma unroll
            for (int t = 0; t < DIM; t++) { grad[t] = in[0][t]; yv[t] -= grad[t]; }  // :1237, the gradient the target evaluated
#pragma unroll
            for (int g = 0; g < 5; g++)
                if ((negmask >> g) & 1u) sample_repulse<DIM>(yv, nrow[g], grad, scale, c.b, a.step);  // :1267-1297
            store_row_through<DIM>(c.y, v, yv);  // :1301
            done_src++;
            i_ev++;
            published = false;
            progressed = true;
        } else if (act && !is_src) {
            bool go = true;
            bool any = false;
#pragma unroll
            for (int u = 0; u < U; u++) {
                go = go && ready[u];
                if (go) {
                    float* su = a.slots + (uint64_t)la_q[u] * (uint64_t)(2 * DIM);
                    slot_clear<DIM>(su);
                    float g2[DIM];
                    sample_attract<DIM>(in[u], yv, g2, la_w[u], (double)la_su[u], c.b, a.step);  // :1207-1238, y_j += g
                    slot_publish<DIM>(su + DIM, g2);
                    i_ev++;
                    any = true;
                }
            }
            if (any) { store_row_through<DIM>(c.y, v, yv); progressed = true; }  // :1239
        }
        if (__any(progressed)) idle = 0;
        else {
            if (++idle > a.poll_budget) {
                if (lane == 0) atomicOr(a.err, kErrPoll);
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    unsigned long long mine = valid ? done_src : 0ull;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mine += __shfl_xor(mine, off);
    if (lane == 0 && mine) atomicAdd(&a.sample_counter[blockIdx.x & 1023u], mine);
    if (a.prof && lane == 0) {
        const unsigned long long t2 = __builtin_amdgcn_s_memtime();
        atomicAdd(&a.prof[0], t1 - t0);
        atomicAdd(&a.prof[1], t2 - t1);
        atomicAdd(&a.prof[2], (unsigned long long)iters);
        atomicAdd(&a.prof[3], 1ull);
        atomicAdd(&a.prof[4], (unsigned long long)wave_total);
        atomicMax(&a.prof[5], (unsigned long long)iters);
        atomicMax(&a.prof[6], t2 - t0);
    }
}

// per block of 64 nodes: sum of the nodes' event rates (1 + in-weight); per node: its in-degree
__global__ void __launch_bounds__(64) ev_rates_kernel(uint64_t n, const uint64_t* __restrict__ tptr, const InEdge* __restrict__ tin,
                                                      unsigned int* __restrict__ out /* [0] max wave rate, [1] max node rate (float bits), [2] max in-degree */) {
    const uint64_t v = blockIdx.x * 64ull + threadIdx.x;
    float r = 0.f;
    uint32_t deg = 0;
    if (v < n) {
        r = 1.f;
        for (uint64_t x = tptr[v]; x < tptr[v + 1]; x++) r += tin[x].w;
        deg = (uint32_t)(tptr[v + 1] - tptr[v]);
    }
    float rs = r, rm = r;
    for (int off = 32; off > 0; off >>= 1) { rs += __shfl_xor(rs, off); rm = fmaxf(rm, __shfl_xor(rm, off)); const uint32_t o = __shfl_xor(deg, off); deg = o > deg ? o : deg; }
    if (threadIdx.x == 0) {
        atomicMax(&out[0], __float_as_uint(rs));
        atomicMax(&out[1], __float_as_uint(rm));
        atomicMax(&out[2], deg);
    }
}
__global__ void ev_pmax_kernel(uint64_t nnz, const float* __restrict__ proba, unsigned int* __restrict__ out) {
    float m = 0.f;
    for (uint64_t e = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; e < nnz; e += (uint64_t)gridDim.x * blockDim.x) m = fmaxf(m, proba[e]);
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));
}

template <int DIM, int KMAX>
void launch_event_k(const EventArgs& a, unsigned grid, bool lookahead) {
    constexpr int UL = DIM <= 4 ? 4 : 2;
    if (lookahead) hipLaunchKernelGGL((ce_event_window_kernel<DIM, KMAX, UL>), dim3(grid), dim3(64), 0, stream(), a);
    else hipLaunchKernelGGL((ce_event_window_kernel<DIM, KMAX, 1>), dim3(grid), dim3(64), 0, stream(), a);
}
template <int DIM>
void launch_event(ae_entropy_optim* o, const EventArgs& a, unsigned grid, bool lookahead) {
    if constexpr (DIM > 0) {
        const uint32_t k = o->g->max_nbng;
        if (k <= 8) launch_event_k<DIM, 8>(a, grid, lookahead);
        else if (k <= 16) launch_event_k<DIM, 16>(a, grid, lookahead);
        else launch_event_k<DIM, 32>(a, grid, lookahead);
    }
}
template <int DIM>
void event_occupancy(ae_entropy_optim* o, int* blocks_per_cu) {
    if constexpr (DIM > 0) {
        const uint32_t k = o->g->max_nbng;
        constexpr int UL = DIM <= 4 ? 4 : 2;
        if (k <= 8) AE_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, ce_event_window_kernel<DIM, 8, UL>, 64, 0));
        else if (k <= 16) AE_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, ce_event_window_kernel<DIM, 16, UL>, 64, 0));
        else AE_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, ce_event_window_kernel<DIM, 32, UL>, 64, 0));
    } else *blocks_per_cu = 0;
}

}  // namespace

namespace ae {

// Graph statistics that size the windows, once per EntropyOptim (after the transposed graph exists)
void ce_event_prepare(ae_entropy_optim* o) {
    const ae_kgraph* g = o->g;
    DevBuf<unsigned int> st(4);
    st.zero();
    hipLaunchKernelGGL(ev_rates_kernel, dim3(blocks_for(g->n, 64)), dim3(64), 0, stream(), g->n, (const uint64_t*)o->tptr.p, (const InEdge*)o->tin.p, st.p);
    hipLaunchKernelGGL(ev_pmax_kernel, dim3(grid_cap(g->nnz, 256)), dim3(256), 0, stream(), g->nnz, (const float*)o->np->proba.p, st.p + 3);
    check_launch("ev_rates");
    unsigned int h[4];
    st.download(h, 4);
    memcpy(&o->ev_wave_rate_max, &h[0], 4);
    memcpy(&o->ev_node_rate_max, &h[1], 4);
    o->ev_indeg_max = h[2];
    memcpy(&o->ev_pmax, &h[3], 4);
    int bpc = 0;
    AE_DISPATCH_DIM(o->dev.dim, event_occupancy, o, &bpc);
    int dev = 0, cus = 0;
    AE_HIP(hipGetDevice(&dev));
    AE_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    o->ev_resident_blocks = (uint64_t)std::max(0, bpc) * (uint64_t)cus;
}

// why the event-ordered kernel cannot run this problem (nullptr: it can)
const char* ce_event_unsupported(const ae_entropy_optim* o) {
    const uint32_t d = o->dev.dim;
    if (!(d == 2 || d == 3 || d == 4 || d == 8 || d == 16)) return "asked_dim must be one of 2, 3, 4, 8, 16";
    if (o->g->max_nbng > 32) return "rows of more than 32 neighbours";
    if (o->dev.node_lo != 0 || o->dev.node_hi != o->dev.n) return "a sharded node range (the rendezvous of an edge's two owners does not span devices)";
    if (o->dev.nnz * (uint64_t)kSlotCap >= 0xFFFFFFFFull) return "more than 2^32 / 8 edges";
    if (blocks_for(o->dev.n, 64) > o->ev_resident_blocks) return "more nodes than resident lanes (every node's lane must be resident for the whole window)";
    if (o->ev_indeg_max > 4096) return "a node with more than 4096 in-edges (its event list would not fit a wave's LDS at any window count)";
    return nullptr;
}

void ce_event_gradient_iteration(ae_entropy_optim* o, uint64_t nb_sample, double grad_step, uint32_t iter) {
    if (const char* why = ce_event_unsupported(o)) fail(AE_ERR_INVALID_ARG, "AE_CE_EVENT: %s; use AE_CE_SEQUENTIAL or AE_CE_HOGWILD", why);
    const uint64_t n = o->dev.n;
    const double per_node = (double)nb_sample / (double)n;
    // windows per batch: (a) the largest per-edge mean stays <= 2 (so the clamp at kSlotCap draws loses < 3e-5 of an edge's
    // samples), (b) a wave's 64 event lists fit its LDS pool with a 1.5x margin (+ 8 sigma), (c) the busiest node's list
    // stays below the cooperative sort's limit, (d) at least 4
    double T = 4.0;
    T = std::max(T, std::ceil(per_node * (double)o->ev_pmax / 2.0));
    T = std::max(T, std::ceil(per_node * (double)o->ev_wave_rate_max / ((double)kPool / 1.5)));
    T = std::max(T, std::ceil(per_node * (double)o->ev_node_rate_max / ((double)kCoopMax / 1.6)));
    if (getenv("AE_EV_WINDOWS")) T = std::max(1.0, atof(getenv("AE_EV_WINDOWS")));
    if (T >= 4096.0 || iter >= (1u << 20)) fail(AE_ERR_INVALID_ARG, "AE_CE_EVENT: window / batch index too large for the RNG key");
    const uint32_t windows = (uint32_t)T;
    o->rounds = windows;
    const uint64_t slot_floats = o->dev.nnz * (uint64_t)kSlotCap * 2ull * o->dev.dim;
    if (o->ev_slots.n < slot_floats) {
        o->ev_slots.alloc(slot_floats);
        AE_HIP(hipMemsetAsync(o->ev_slots.p, 0xFF, sizeof(float) * slot_floats, stream()));  // every slot "unpublished"
    }
    EventArgs a;
    a.c = o->dev;
    a.tptr = o->tptr.p;
    a.tin = o->tin.p;
    a.slots = o->ev_slots.p;
    a.unit = (float)(per_node / (double)windows);
    a.step = grad_step;
    a.sample_counter = o->sample_counter.p;
    a.err = o->err.p;
    a.poll_budget = 1u << 20;
    static DevBuf<unsigned long long> prof_buf;
    a.prof = nullptr;
    const bool prof = getenv("AE_CE_PROF") != nullptr;
    if (prof) {
        if (!prof_buf.n) { prof_buf.alloc(8); prof_buf.zero(); }
        a.prof = prof_buf.p;
    }
    const bool lookahead = !getenv("AE_EV_NO_LOOKAHEAD");
    const unsigned grid = blocks_for(n, 64);
    for (uint32_t w = 0; w < windows; w++) {
        a.window_key = (iter << 12) | w;
        AE_DISPATCH_DIM(o->dev.dim, launch_event, o, a, grid, lookahead);
    }
    check_launch("ce_event");
    if (prof) {
        unsigned long long h[8];
        prof_buf.download(h, 8);
        if (h[3]) fprintf(stderr, "CEEVPROF windows=%u waves=%llu per wave-window: build %.0f cyc, walk %.0f cyc, iterations %.1f (max %llu), events %.1f, slowest wave %llu cyc\n",
                          windows, h[3], (double)h[0] / h[3], (double)h[1] / h[3], (double)h[2] / h[3], h[5], (double)h[4] / h[3], h[6]);
        prof_buf.zero();
    }
}

}  // namespace ae
