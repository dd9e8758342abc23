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
constexpr int kPoolPerNode = 32;   // LDS event entries per node of a wave (pooled over the wave's nodes)
constexpr int kPrivSort = 32;      // segments up to this length are insertion-sorted by their lane
constexpr int kCoopMax = 512;      // longer segments are rank-sorted by the whole wave; this is the limit
constexpr uint32_t kTagEvCount = 0xFFFF0021u, kTagEvTime = 0xFFFF0022u, kTagEvNeg = 0xFFFF0023u;
constexpr uint32_t kErrPool = 16u, kErrPoll = 32u, kErrNeg = 64u;

struct EventArgs {
    CeDev c;
    const uint64_t* tptr;
    const InEdge* tin;
    float* slots;               // [nnz * kSlotCap][2 DIM]: the source's row y_i | the target's row y_j, as they are when each reaches the event
    uint32_t window_key;        // (batch << 12) | window
    float unit;                 // mu_e per window = unit * p_e
    double step;
    unsigned long long* sample_counter;
    unsigned int* err;
    uint32_t poll_budget;
    uint32_t regather;          // a waiting source lane re-reads its negatives' rows every this many trips
    uint32_t lookahead_min;     // lanes with at least this many events in the window look ahead over runs of target events
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

// One pair step of embedder.rs:1207-1297 on the row y against the row o:  g = (o - y) c ;  y -= g.  That single form is
// the source's attraction (y_i -= (y_j - y_i) c, :1237), the target's (y_j += (y_j - y_i) c, :1238) and a repulsion
// (y_i -= (y_k - y_i) c, :1297).  Scalars in f64 as the reference (:1207-1229); with D = |y - o|^2 and S2 = scale^2 the
// reference's  delta = D / S2, coeff = 2 / (1 + delta) / S2  collapse into ONE division:
//   attraction  c = max( 2 step ((1 - w) S2^2 - w M) / ((S2 + D) M), -0.49 ),  M = max(D^2, 1e4 S2^2)      (:1216-1233)
//   repulsion   c = min( 2 step S2^2 / ((S2 + D) max(D^2, S2^2 / 16)), 2 )                                  (:1275-1293)
// (b == 1; the general exponent goes through sample_attract / sample_repulse of ce_sample_math.h).  A step with D = 0
// leaves `g` as it was (the reference's stale-gradient quirk, :1286-1297, B4).
template <int DIM, bool ATTRACT>
__device__ __forceinline__ void pair_step(float* y, const float* o, float* g, float w, double S2, double two_step, double b, double step) {
    float acc = 0.f;
#pragma unroll
    for (int t = 0; t < DIM; t++) { const float df = y[t] - o[t]; acc += df * df; }
    if (b != 1.) {  // rare: keep the reference's own formulation
        const double sc = sqrt(S2);
        if constexpr (ATTRACT) {
            const double ds = (double)acc / (sc * sc);
            const double coeff = grad_coeff(ds, sc, b);
            if (ds > 0.) {
                const double rep = 1. / fmax(ds * ds, (double)(1.0f / kProbaMin));
                const float cf = (float)fmax(step * coeff * (-(double)w + (1. - (double)w) * rep), -0.49);
#pragma unroll
                for (int t = 0; t < DIM; t++) g[t] = (o[t] - y[t]) * cf;
            } else {
#pragma unroll
                for (int t = 0; t < DIM; t++) g[t] = 0.f;
            }
        } else {
            const double ds = (double)acc / (sc * sc);
            const double coeff = grad_coeff(ds, sc, b);
            if (acc > 0.f) {
                const float cf = (float)fmin(step * coeff * (1. / fmax(ds * ds, 1. / 16.)), 2.);
#pragma unroll
                for (int t = 0; t < DIM; t++) g[t] = (o[t] - y[t]) * cf;
            }
        }
    } else {
        const double D = (double)acc, S4 = S2 * S2;
        double c;
        if constexpr (ATTRACT) {
            const double M = fmax(D * D, (double)(1.0f / kProbaMin) * S4);
            c = fmax(two_step * ((1. - (double)w) * S4 - (double)w * M) / ((S2 + D) * M), -0.49);
        } else {
            c = fmin(two_step * S4 / ((S2 + D) * fmax(D * D, S4 * (1. / 16.))), 2.);
        }
        const float cf = (float)c;
        if (ATTRACT || acc > 0.f) {
#pragma unroll
            for (int t = 0; t < DIM; t++) g[t] = (o[t] - y[t]) * cf;
        }
    }
#pragma unroll
    for (int t = 0; t < DIM; t++) y[t] -= g[t];
}

// NPW nodes per wave (lanes >= NPW idle): fewer nodes per wave = fewer lanes in different states per trip of the walk loop
template <int DIM, int KMAX, int U, int NPW>
__global__ void __launch_bounds__(64) ce_event_window_kernel(EventArgs a) {
    constexpr int P = kPoolPerNode * NPW;
    __shared__ uint32_t s_time[P], s_q[P], s_aux[P];
    const CeDev c = a.c;
    const int lane = threadIdx.x;
    if (__hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & (kErrPoll | kErrPool)) return;  // a failed window: do not spin again
    const uint64_t local = blockIdx.x * (uint64_t)NPW + (uint64_t)lane;
    const bool valid = lane < NPW && local < c.n;
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
    load_row_coherent<DIM>(c.y, v, yv);
    const double scale = (double)c.emb_scale[v];
    const double S2own = scale * scale, two_step = 2. * a.step;
    const uint32_t hk = pcg_hash(pcg_hash((uint32_t)c.seed ^ 0x5bd1e995u) ^ pcg_hash(a.window_key + (uint32_t)(c.seed >> 32)));
    const uint32_t ck = hk ^ kTagEvCount, tk = pcg_hash(hk ^ kTagEvTime);
    unsigned long long t0 = a.prof ? __builtin_amdgcn_s_memtime() : 0ull;
    // ---- pass 1: how many events does the node have in this window.  The in-edge records are read four at a time;
    // the counts of the first 32 in-edges are kept (4 bits each) for pass 2
    uint32_t cnt_out[KMAX];
    uint32_t tot = 0;
#pragma unroll
    for (int m = 0; m < KMAX; m++) {
        const bool has = (uint32_t)m < k && valid;
        nbr_reg[m] = (uint32_t)m < k ? nbr_reg[m] : 0xFFFFFFFFu;  // the pad never equals a candidate
        cnt_out[m] = has ? ev_count(ib + m, a.unit * pr[m], ck) : 0u;
        tot += cnt_out[m];
    }
    unsigned long long cin0 = 0ull, cin1 = 0ull;
    for (uint64_t x = tb; x < te; x += 8) {
        InEdge rec[8];
#pragma unroll
        for (int z = 0; z < 8; z++) rec[z] = a.tin[x + z < te ? x + z : te - 1];
#pragma unroll
        for (int z = 0; z < 8; z++) {
            const uint32_t cn = x + z < te ? ev_count(rec[z].eid, a.unit * rec[z].w, ck) : 0u;
            tot += cn;
            const uint64_t li = x + z - tb;
            if (li < 16) cin0 |= (unsigned long long)cn << (4 * li);
            else if (li < 32) cin1 |= (unsigned long long)cn << (4 * (li - 16));
        }
    }
    uint32_t incl = tot;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const uint32_t o = __shfl_up(incl, off); incl += lane >= off ? o : 0u; }
    const uint32_t wave_total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    uint32_t tmax = tot;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const uint32_t o = __shfl_xor(tmax, off); tmax = o > tmax ? o : tmax; }
    if (wave_total > (uint32_t)P || tmax > (uint32_t)kCoopMax) {  // sized by the host with a wide margin: an error, not a path
        if (lane == 0) atomicOr(a.err, kErrPool);
        return;
    }
    const uint32_t seg = incl - tot;
    // ---- pass 2: the events, unsorted.  aux: source events carry the row position m, target events 2^31 | in-edge position
    {
        uint32_t pos = seg;
#pragma unroll
        for (int m = 0; m < KMAX; m++) {
            for (uint32_t r = 0; r < cnt_out[m]; r++) {
                s_time[pos] = ev_time(ib + m, r, tk);
                s_q[pos] = (uint32_t)(ib + m) * (uint32_t)kSlotCap + r;
                s_aux[pos] = (uint32_t)m;
                pos++;
            }
        }
        for (uint64_t x0 = tb; x0 < te; x0 += 8) {
            InEdge rec[8];
#pragma unroll
            for (int z = 0; z < 8; z++) rec[z] = a.tin[x0 + z < te ? x0 + z : te - 1];
#pragma unroll
            for (int z = 0; z < 8; z++) {
                const uint64_t li = x0 + z - tb;
                uint32_t cn = 0;
                if (x0 + z < te) {
                    if (li < 32) cn = (uint32_t)((li < 16 ? cin0 >> (4 * li) : cin1 >> (4 * (li - 16))) & 15ull);
                    else cn = ev_count(rec[z].eid, a.unit * rec[z].w, ck);
                }
                for (uint32_t r = 0; r < cn; r++) {
                    s_time[pos] = ev_time(rec[z].eid, r, tk);
                    s_q[pos] = rec[z].eid * (uint32_t)kSlotCap + r;
                    s_aux[pos] = 0x80000000u | (uint32_t)li;
                    pos++;
                }
            }
        }
    }
    // ---- sort every segment by (time, slot id): one global order that all nodes agree on
    if (tot <= (uint32_t)kPrivSort) {
        for (uint32_t i = 1; i < tot; i++) {
            const uint32_t ti = s_time[seg + i], qi = s_q[seg + i], ai = s_aux[seg + i];
            uint32_t j = i;
            while (j > 0) {
                const uint32_t tj = s_time[seg + j - 1], qj = s_q[seg + j - 1];
                if (tj < ti || (tj == ti && qj < qi)) break;
                s_time[seg + j] = tj; s_q[seg + j] = qj; s_aux[seg + j] = s_aux[seg + j - 1];
                j--;
            }
            s_time[seg + j] = ti; s_q[seg + j] = qi; s_aux[seg + j] = ai;
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
            uint32_t et[PER], eq[PER], ea[PER], rank[PER];
#pragma unroll
            for (int h = 0; h < PER; h++) {
                const uint32_t i = (uint32_t)(h * 64 + lane);
                const bool in = i < nL;
                et[h] = in ? s_time[sL + i] : 0u; eq[h] = in ? s_q[sL + i] : 0u; ea[h] = in ? s_aux[sL + i] : 0u;
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
                if ((uint32_t)(h * 64 + lane) < nL) { s_time[sL + rank[h]] = et[h]; s_q[sL + rank[h]] = eq[h]; s_aux[sL + rank[h]] = ea[h]; }
            }
            wave_sync_lds();
        }
    }
    // ---- the 5 negatives of every source event (embedder.rs:1241-1253): uniform (or NodeSampler, :927-930) draws, rejected
    // when k = i, k = j or k in N(i) (NodeParam::get_edge, nodeparam.rs:83-85; j is in N(i)).  Done here, all lanes in step, and
    // remembered as the five accepted ATTEMPT numbers (6 bits each) in the event's dead time word: the walk below only
    // re-evaluates five hashes.
    const uint32_t node_base = pcg_hash(pcg_hash((uint32_t)c.seed ^ a.window_key ^ kTagEvNeg) + v);
    const bool hub = c.hub_odds != nullptr;
    auto neg_candidate = [&](uint32_t ev, uint32_t attempt) -> uint32_t { return pcg_hash(node_base + ev * 64u + attempt); };
    {
        uint32_t cur = 0;  // position of the lane's next source event
        while (true) {
            while (cur < tot && (s_aux[seg + cur] >> 31)) cur++;
            const bool mine = cur < tot;
            if (!__any(mine)) break;
            const uint32_t i = cur;
            uint32_t packed = 0, got = 0;
            for (uint32_t round = 0; round < 8u && __any(mine && got < 5u); round++) {
                uint32_t cand[8];
                if (hub) {
                    uint32_t xs[8], al[8];
                    float od[8], uu[8];
#pragma unroll
                    for (int z = 0; z < 8; z++) {
                        const uint32_t w0 = neg_candidate(i, round * 8u + (uint32_t)z);
                        xs[z] = __umulhi(w0, (uint32_t)c.n);
                        uu[z] = (float)(pcg_hash(w0 ^ 0x9E3779B9u) >> 8) * (1.0f / 16777216.0f);
                        const uint2 he = c.hub_tab[xs[z]];
                        od[z] = __uint_as_float(he.x);
                        al[z] = he.y;
                    }
#pragma unroll
                    for (int z = 0; z < 8; z++) cand[z] = (uu[z] < od[z]) ? xs[z] : al[z];
                } else {
#pragma unroll
                    for (int z = 0; z < 8; z++) cand[z] = __umulhi(neg_candidate(i, round * 8u + (uint32_t)z), (uint32_t)c.n);  // :1121
                }
#pragma unroll
                for (int z = 0; z < 8; z++) {
                    uint32_t acc = cand[z] ^ v;
#pragma unroll
                    for (int m = 0; m < KMAX; m++) { const uint32_t x = nbr_reg[m] ^ cand[z]; acc = x < acc ? x : acc; }
                    const bool ok = acc != 0u && got < 5u;
                    packed |= ok ? (round * 8u + (uint32_t)z) << (6u * got) : 0u;
                    got += ok ? 1u : 0u;
                }
            }
            if (mine) {
                s_time[seg + i] = packed | (got == 5u ? 1u << 30 : 0u);  // got < 5 only on a graph too small for five negatives: flagged, no repulsion
                if (got < 5u) atomicOr(a.err, kErrNeg);
                cur++;
            }
        }
    }
    unsigned long long t1 = a.prof ? __builtin_amdgcn_s_memtime() : 0ull;
    // ---- walk the list.  Head event: publish the own row in the event's slot (source: half 0, target: half 1), poll the
    // partner's half; when it is there BOTH owners evaluate the same attraction on the same two rows (identical arithmetic,
    // identical result: one gradient for both ends, embedder.rs:1228-1239) and move on -- one memory hop per event.
    // A run of consecutive TARGET events is served in one trip (their sources published long ago: hubs stream).
    uint32_t i_ev = 0;
    bool published = false;
    uint32_t q_head = 0, aux_head = 0;
    float nrow[5][DIM], grad[DIM];
#pragma unroll
    for (int t = 0; t < DIM; t++) grad[t] = 0.f;
#pragma unroll
    for (int g = 0; g < 5; g++)
#pragma unroll
        for (int t = 0; t < DIM; t++) nrow[g][t] = 0.f;
    uint32_t negmask = 0;
    float w_head = 0.f;
    uint32_t idle = 0, iters = 0;
    unsigned long long done_src = 0;
    const uint64_t SL = 2ull * DIM;
    uint32_t waited = 0;
    // The 5 negatives of the head source event (drawn in the build phase): their rows are gathered when the event is published
    // and AGAIN every `regather` trips while the lane waits for its partner, so that the rows used are at most a few trips
    // old (the reference reads them at the moment of use).  Rows as old as the whole wait were measured to bias the result:
    // final CE +4 %, edge-length quantiles -7 % against the sequential loop (60 k nodes, k = 6, 40 batches).
    auto gather_negatives = [&] {
        const uint32_t packed = s_time[seg + i_ev];
        negmask = (packed >> 30) & 1u ? 31u : 0u;
        uint32_t kk[5];
#pragma unroll
        for (int g = 0; g < 5; g++) {
            const uint32_t w0 = neg_candidate(i_ev, (packed >> (6 * g)) & 63u);
            uint32_t x = __umulhi(w0, (uint32_t)c.n);
            if (hub) {
                const float uu = (float)(pcg_hash(w0 ^ 0x9E3779B9u) >> 8) * (1.0f / 16777216.0f);
                const uint2 he = c.hub_tab[x];
                x = (uu < __uint_as_float(he.x)) ? x : he.y;
            }
            kk[g] = x;
        }
#pragma unroll
        for (int g = 0; g < 5; g++) load_row_coherent<DIM>(c.y, kk[g], nrow[g]);
        waited = 0;
    };
    while (true) {
        const bool act = i_ev < tot;
        if (!__any(act)) break;
        iters++;
        if (act && !published) {  // arrive at the head event
            q_head = s_q[seg + i_ev];
            aux_head = s_aux[seg + i_ev];
            const bool src = !(aux_head >> 31);
            slot_publish<DIM>(a.slots + (uint64_t)q_head * SL + (src ? 0 : DIM), yv);
            published = true;
            if (src) {
                w_head = c.proba[q_head / (uint32_t)kSlotCap];
                gather_negatives();
            }
        }
        const bool is_src = act && !(aux_head >> 31);
        // partner rows: of the head event, and of up to U - 1 following events while they are target events too
        float in[U][DIM], ws[U], ss[U];
        bool rdy[U], tg[U];
        uint32_t qs[U], as[U];
#pragma unroll
        for (int u = 0; u < U; u++) { rdy[u] = false; tg[u] = false; qs[u] = q_head; as[u] = aux_head; ws[u] = 0.f; ss[u] = 1.f; }
        if (act) {
            if (is_src) {
                rdy[0] = df_try_load_version<DIM>(a.slots + (uint64_t)q_head * SL + DIM, 0, in[0]);
            } else {
                bool chain = true;
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const bool have = chain && i_ev + (uint32_t)u < tot && (u == 0 || tot >= a.lookahead_min);
                    if (u > 0 && have) { qs[u] = s_q[seg + i_ev + (uint32_t)u]; as[u] = s_aux[seg + i_ev + (uint32_t)u]; }
                    chain = have && (as[u] >> 31);
                    tg[u] = chain;
                    if (chain) {
                        const InEdge rec = a.tin[tb + (uint64_t)(as[u] & 0x7FFFFFFFu)];
                        ws[u] = rec.w; ss[u] = rec.s_src;
                        rdy[u] = df_try_load_version<DIM>(a.slots + (uint64_t)qs[u] * SL, 0, in[u]);
                    }
                }
            }
        }
        bool progressed = false;
        if (act && is_src) {
            if (rdy[0]) {
                pair_step<DIM, true>(yv, in[0], grad, w_head, S2own, two_step, c.b, a.step);  // :1207-1237, the y_i half
#pragma unroll
                for (int g = 0; g < 5; g++)
                    if ((negmask >> g) & 1u) pair_step<DIM, false>(yv, nrow[g], grad, 0.f, S2own, two_step, c.b, a.step);  // :1267-1297
                store_row_through<DIM>(c.y, v, yv);  // :1301
                done_src++;
                i_ev++;
                published = false;
                progressed = true;
            } else if (++waited >= a.regather) {
                gather_negatives();  // still waiting: refresh the negatives' rows (see gather_negatives)
            }
        } else if (act) {
            bool stop = false;
            uint32_t nproc = 0;
            bool pub = true;  // is the event we stand at published?
#pragma unroll
            for (int u = 0; u < U; u++) {
                if (!stop) {
                    if (!tg[u]) { stop = true; pub = false; }  // the next event is a source event (or the list ends): arrive there next trip
                    else {
                        if (u > 0) slot_publish<DIM>(a.slots + (uint64_t)qs[u] * SL + DIM, yv);  // we reach target event u now
                        if (rdy[u]) {
                            float g2[DIM];
                            const double su = (double)ss[u];
                            pair_step<DIM, true>(yv, in[u], g2, ws[u], su * su, two_step, c.b, a.step);  // :1207-1238, the y_j half
                            nproc++;
                        } else {
                            stop = true;
                            q_head = qs[u]; aux_head = as[u];
                        }
                    }
                }
            }
            if (!stop) pub = false;  // all U served: the next head is not reached yet
            i_ev += nproc;
            published = pub && i_ev < tot;
            if (nproc) { store_row_through<DIM>(c.y, v, yv); progressed = true; }  // :1239
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
    // the partner halves this node has read go back to "unpublished" (each half has one writer and one reader; the next window
    // is another launch): kept out of the walk, where every store in flight lengthens the next poll's wait
    for (uint32_t i = 0; i < i_ev; i++) {
        const uint32_t q = s_q[seg + i];
        slot_clear<DIM>(a.slots + (uint64_t)q * SL + ((s_aux[seg + i] >> 31) ? 0 : DIM));
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

// event rates (1 + in-weight) summed over the nodes of a wave, for waves of 64 / 32 / 16 nodes; the busiest node; the
// largest in-degree
__global__ void __launch_bounds__(64) ev_rates_kernel(uint64_t n, const uint64_t* __restrict__ tptr, const InEdge* __restrict__ tin,
                                                      unsigned int* __restrict__ out /* [0] wave64 [1] node rate [2] in-degree [3] pmax [4] wave32 [5] wave16 (float bits) */) {
    const uint64_t v = blockIdx.x * 64ull + threadIdx.x;
    float r = 0.f;
    uint32_t deg = 0;
    if (v < n) {
        r = 1.f;
        for (uint64_t x = tptr[v]; x < tptr[v + 1]; x++) r += tin[x].w;
        deg = (uint32_t)(tptr[v + 1] - tptr[v]);
    }
    float rs = r, rm = r, r16 = 0.f, r32 = 0.f;
    for (int off = 1; off < 64; off <<= 1) {
        rs += __shfl_xor(rs, off);
        rm = fmaxf(rm, __shfl_xor(rm, off));
        const uint32_t o = __shfl_xor(deg, off);
        deg = o > deg ? o : deg;
        if (off == 8) r16 = rs;
        if (off == 16) r32 = rs;
    }
    for (int off = 32; off > 0; off >>= 1) { r16 = fmaxf(r16, __shfl_xor(r16, off)); r32 = fmaxf(r32, __shfl_xor(r32, off)); }
    if (threadIdx.x == 0) {
        atomicMax(&out[0], __float_as_uint(rs));
        atomicMax(&out[1], __float_as_uint(rm));
        atomicMax(&out[2], deg);
        atomicMax(&out[4], __float_as_uint(r32));
        atomicMax(&out[5], __float_as_uint(r16));
    }
}
__global__ void ev_pmax_kernel(uint64_t nnz, const float* __restrict__ proba, unsigned int* __restrict__ out) {
    float m = 0.f;
    for (uint64_t e = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; e < nnz; e += (uint64_t)gridDim.x * blockDim.x) m = fmaxf(m, proba[e]);
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));
}

template <int DIM>
constexpr int lookahead_of() { return DIM <= 4 ? 4 : 2; }

// op 0: launch; op 1: occupancy query (blocks per CU) of the same instantiation
template <int DIM, int KMAX, int NPW>
void event_kernel_op(int op, const EventArgs& a, unsigned grid, int* bpc) {
    auto kern = ce_event_window_kernel<DIM, KMAX, lookahead_of<DIM>(), NPW>;
    if (op == 0) hipLaunchKernelGGL(kern, dim3(grid), dim3(64), 0, stream(), a);
    else AE_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(bpc, kern, 64, 0));
}
template <int DIM, int KMAX>
void event_kernel_npw(int npw, int op, const EventArgs& a, unsigned grid, int* bpc) {
    if (npw == 64) event_kernel_op<DIM, KMAX, 64>(op, a, grid, bpc);
    else event_kernel_op<DIM, KMAX, 32>(op, a, grid, bpc);
}
template <int DIM>
void event_kernel(ae_entropy_optim* o, int npw, int op, const EventArgs& a, unsigned grid, int* bpc) {
    if constexpr (DIM > 0) {
        const uint32_t k = o->g->max_nbng;
        if (k <= 8) event_kernel_npw<DIM, 8>(npw, op, a, grid, bpc);
        else if (k <= 16) event_kernel_npw<DIM, 16>(npw, op, a, grid, bpc);
        else event_kernel_npw<DIM, 32>(npw, op, a, grid, bpc);
    } else if (bpc) *bpc = 0;
}

}  // namespace

namespace ae {

// tuning / A-B switches are read only when AE_DEBUG_KNOBS is set: a release run cannot be altered from the environment
static const char* knob(const char* name) { return debug_knob("AE_DEBUG_KNOBS") ? getenv(name) : nullptr; }

// Graph statistics that size the windows, once per EntropyOptim (after the transposed graph exists)
void ce_event_prepare(ae_entropy_optim* o) {
    const ae_kgraph* g = o->g;
    DevBuf<unsigned int> st(6);
    st.zero();
    hipLaunchKernelGGL(ev_rates_kernel, dim3(blocks_for(g->n, 64)), dim3(64), 0, stream(), g->n, (const uint64_t*)o->tptr.p, (const InEdge*)o->tin.p, st.p);
    hipLaunchKernelGGL(ev_pmax_kernel, dim3(grid_cap(g->nnz, 256)), dim3(256), 0, stream(), g->nnz, (const float*)o->np->proba.p, st.p + 3);
    check_launch("ev_rates");
    unsigned int h[6];
    st.download(h, 6);
    memcpy(&o->ev_wave_rate_max[0], &h[0], 4);
    memcpy(&o->ev_wave_rate_max[1], &h[4], 4);
    memcpy(&o->ev_wave_rate_max[2], &h[5], 4);
    memcpy(&o->ev_node_rate_max, &h[1], 4);
    o->ev_indeg_max = h[2];
    memcpy(&o->ev_pmax, &h[3], 4);
    // nodes per wave: the fewest that still leaves every node's lane resident (fewer nodes per wave = fewer lanes in
    // different states per trip of the walk loop, more waves to overlap one's memory round trips with another's arithmetic)
    int dev = 0, cus = 0;
    AE_HIP(hipGetDevice(&dev));
    AE_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    EventArgs dummy{};
    o->ev_npw = 0;
    const int forced = knob("AE_EV_NPW") ? atoi(knob("AE_EV_NPW")) : 0;
    for (int npw : {32, 64}) {
        if (forced && npw != forced) continue;
        int bpc = 0;
        AE_DISPATCH_DIM(o->dev.dim, event_kernel, o, npw, 1, dummy, 0u, &bpc);
        const uint64_t resident = (uint64_t)std::max(0, std::min(bpc, 32)) * (uint64_t)cus;
        o->ev_npw = npw;
        o->ev_resident_blocks = resident;
        if ((double)blocks_for(g->n, (unsigned)npw) <= 0.92 * (double)resident) break;  // (margin: the occupancy query can be one block per CU high)
    }
}

// why the event-ordered kernel cannot run this problem (nullptr: it can)
const char* ce_event_unsupported(const ae_entropy_optim* o) {
    const uint32_t d = o->dev.dim;
    if (!(d == 2 || d == 3 || d == 4 || d == 8 || d == 16)) return "asked_dim must be one of 2, 3, 4, 8, 16";
    if (o->g->max_nbng > 32) return "rows of more than 32 neighbours";
    if (o->dev.node_lo != 0 || o->dev.node_hi != o->dev.n) return "a sharded node range (the rendezvous of an edge's two owners does not span devices)";
    if (o->dev.nnz * (uint64_t)kSlotCap >= 0xFFFFFFFFull) return "more than 2^32 / 8 edges";
    if (!o->ev_npw || (double)blocks_for(o->dev.n, (unsigned)o->ev_npw) > 0.92 * (double)o->ev_resident_blocks) return "more nodes than resident lanes (every node's lane must be resident for the whole window)";
    if (o->ev_indeg_max > 4096) return "a node with more than 4096 in-edges (its event list would not fit a wave's LDS at any window count)";
    return nullptr;
}

void ce_event_gradient_iteration(ae_entropy_optim* o, uint64_t nb_sample, double grad_step, uint32_t iter) {
    if (const char* why = ce_event_unsupported(o)) fail(AE_ERR_INVALID_ARG, "AE_CE_EVENT: %s; use AE_CE_SEQUENTIAL or AE_CE_HOGWILD", why);
    const uint64_t n = o->dev.n;
    const double per_node = (double)nb_sample / (double)n;
    // windows per batch: (a) the largest per-edge mean stays <= 2 (so the clamp at kSlotCap draws loses < 3e-5 of an edge's
    // samples), (b) a wave's 64 event lists fit its LDS pool with a 1.5x margin (+ 8 sigma), (c) the busiest node's list
    // stays below the cooperative sort's limit, (d) at least 4.  LONG windows are deliberate: lanes are coupled only through
    // their own events; in a long window the nodes settle into advancing at a common pace (every node waits for its partners),
    // while a short window is over before that happens -- quiet nodes rush to its end and the negatives read across the skew are
    // stale or premature.  Measured on the Higgs-shaped 60 k graph (k = 6, 40 batches from the dmap initialisation; ratio to the
    // sequential loop): 10-20 windows per batch CE 1.002-1.012 / quartiles within 3 %; 25-45 windows CE 1.02-1.04 / lower
    // quartiles -6 ... -11 %; 240 windows (half an event per node and window) CE 1.006 / within 2 % again, at 1.4x the time.
    double T = 4.0;
    T = std::max(T, std::ceil(per_node * (double)o->ev_pmax / 2.0));
    T = std::max(T, std::ceil(per_node * (double)o->ev_wave_rate_max[o->ev_npw == 64 ? 0 : (o->ev_npw == 32 ? 1 : 2)] / ((double)(o->ev_npw * kPoolPerNode) / 1.5)));
    T = std::max(T, std::ceil(per_node * (double)o->ev_node_rate_max / ((double)kCoopMax / 1.6)));
    if (knob("AE_EV_WINDOWS")) T = std::max(1.0, atof(knob("AE_EV_WINDOWS")));
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
    a.regather = knob("AE_EV_REGATHER") ? (uint32_t)atoi(knob("AE_EV_REGATHER")) : 2u;
    a.lookahead_min = knob("AE_EV_LOOKAHEAD_MIN") ? (uint32_t)atoi(knob("AE_EV_LOOKAHEAD_MIN")) : 0u;
    static DevBuf<unsigned long long> prof_buf;
    a.prof = nullptr;
    const bool prof = debug_knob("AE_CE_PROF") != nullptr;
    if (prof) {
        if (!prof_buf.n) { prof_buf.alloc(8); prof_buf.zero(); }
        a.prof = prof_buf.p;
    }
    const bool lookahead = !knob("AE_EV_NO_LOOKAHEAD");
    (void)lookahead;
    const unsigned grid = blocks_for(n, (unsigned)o->ev_npw);
    for (uint32_t w = 0; w < windows; w++) {
        a.window_key = (iter << 12) | w;
        AE_DISPATCH_DIM(o->dev.dim, event_kernel, o, o->ev_npw, 0, a, grid, nullptr);
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
