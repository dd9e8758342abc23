// ce_node.hip -- AE_CE_HOGWILD: the lock-free CE gradient batch (gradient_iteration_threaded,
// src/embedder.rs:1311-1315) restructured for MI355X as an OWNER-COMPUTES, node-centric kernel.
//
// Why not "one thread per sample with racy read-modify-write", which is what the reference's rayon
// loop does (:1185-1186, :1239, :1301)?  Measured on MI355X (tools/ubench_gather.hip): with ~2 M samples
// in flight over 60 k nodes 91 % of racy 8-byte read-modify-writes are lost, the 8 per-XCD L2s are not
// coherent inside a launch, and atomics of any type top out at ~23 G/s (vs 55-270 G/s for random 8-byte
// loads, ~80 G/s for write-through stores).  A CPU has tens of threads for N nodes; the GPU has more
// lanes than nodes.  So every coordinate row gets exactly ONE writer:
//
//   thread v owns y_v.  Per round it replays, sequentially on its private copy of y_v,
//     (a) the samples whose source is v: attraction to the sampled neighbour j (the y_i half of
//         :1223-1237) followed by the 5 negative repulsions (:1241-1299), and
//     (b) the samples whose target is v: the y_j half of :1238-1239, recomputed from (y_u, y_v, s_u, w).
//   The number of times edge e is sampled in a round is c_e ~ Poisson(mu_e), mu_e = nb_sample * p_e /
//   (N * rounds): by Poisson splitting this is the edge law of the reference's i.i.d. draw from the
//   alias table (:987, :1182) with a Poisson(nb_sample) total.  c_e is a pure function of (seed, batch,
//   round, e) (one Philox block), so thread i and thread j agree on it with no communication, no
//   atomics, no sorting -- and on multi-GPU the owner of j replays remote pushes from its replica.
//   y_v is written back (write-through, agent scope) after every sample; partners are re-read
//   (L1-bypassing loads) for every sample, so the staleness is that of the memory system, like Hogwild.
//
// Arithmetic: f32 with v_rcp_f32 (the reference's scalars are f64; the coefficient is a smooth
// function clipped to [-0.49, 2], a 1e-7 relative difference is far below the SGD noise).  The exact
// f64 arithmetic lives in ce.hip (AE_CE_SEQUENTIAL, bit-exact against the oracle).
#include "ce_internal.h"
#include "philox.h"

using namespace ae;

namespace ae {
void sort_pairs_u64_u32(uint64_t* d_keys_in, uint64_t* d_keys_out, uint32_t* d_vals_in, uint32_t* d_vals_out, uint64_t count);
void rowptr_from_sorted_keys(const uint64_t* d_keys, uint64_t nnz, uint64_t nrows, uint64_t* d_rowptr);
}  // namespace ae

namespace {

constexpr uint32_t kTagEdgeCount = 0xFFFF0010u;
constexpr uint32_t kTagNodeRng = 0xFFFF0011u;
constexpr int kBlock = 256;
constexpr int kApplyBlock = 64;  // one wave per workgroup: few nodes => spread the waves over all CUs

template <int DIM>
__device__ __forceinline__ void load_row_fresh(const float* __restrict__ y, uint32_t node, float* out) {
    // L1-bypassing loads: another CU may have rewritten the row since this CU cached it
    const float* p = y + (uint64_t)node * DIM;
    if constexpr (DIM % 2 == 0) {
        using f2 = __attribute__((ext_vector_type(2))) float;
#pragma unroll
        for (int q = 0; q < DIM / 2; q++) {
            f2 t = __builtin_nontemporal_load(reinterpret_cast<const f2*>(p) + q);
            out[2 * q] = t.x; out[2 * q + 1] = t.y;
        }
    } else {
#pragma unroll
        for (int t = 0; t < DIM; t++) out[t] = __builtin_nontemporal_load(p + t);
    }
}
template <int DIM>
__device__ __forceinline__ void store_row_through(float* __restrict__ y, uint32_t node, const float* in) {
    // agent-scope (write-through) stores: the owner's update becomes visible to the other XCDs
    float* p = y + (uint64_t)node * DIM;
    if constexpr (DIM % 2 == 0) {
#pragma unroll
        for (int q = 0; q < DIM / 2; q++) {
            uint64_t bits = ((uint64_t)__float_as_uint(in[2 * q + 1]) << 32) | __float_as_uint(in[2 * q]);
            __hip_atomic_store(reinterpret_cast<uint64_t*>(p) + q, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    } else {
#pragma unroll
        for (int t = 0; t < DIM; t++) __hip_atomic_store(reinterpret_cast<uint32_t*>(p) + t, __float_as_uint(in[t]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

__device__ __forceinline__ float rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// 2b * cauchy_weight * delta^(b-1) / s^2, embedder.rs:1216-1222 (f32)
template <bool B1>
__device__ __forceinline__ float grad_coeff_f32(float delta, float inv_s2, float b) {
    if constexpr (!B1) {  // general exponent: kept out of the b == 1 instantiation (the powf code is ~8x the loop body)
        const float db = __powf(delta, b);
        return 2.0f * b * rcp(1.0f + db) * __powf(delta, b - 1.0f) * inv_s2;
    }
    return 2.0f * inv_s2 * rcp(1.0f + delta);
}

struct NodeArgs {
    CeDev c;
    const uint64_t* tptr;
    const InEdge* tin;
    uint8_t* cnt;        // per edge: number of samples of the edge in this round
    uint32_t* tot;       // per owned node: number of planned out-samples (<= cap)
    uint32_t* plan;      // per owned node: cap slots x 6 words {j, k1..k5}
    uint32_t cap;
    uint32_t round_key;
    float step;
    float unit;  // mu_e = unit * p_e
    float b;
    unsigned long long* sample_counter;
    unsigned int* overflow;
    unsigned long long* prof;  // debug: per-section cycle sums [stage, fetch-issue, compute, store, in-phase, total]
    int skip;        // debug: 1 skip out-phase, 2 skip in-phase
    int store_mode;  // 0: write-through store after every sample, 1: plain store after every sample, 2: write-through at phase ends
};

// PCG-RXS-M-XS 32 output hash: the fast mode's stream for the negative draws (the exact Philox stream
// of the oracle is used by AE_CE_SEQUENTIAL; this mode is validated statistically)
__device__ __forceinline__ uint32_t pcg_hash(uint32_t x) {
    uint32_t s = x * 747796405u + 2891336453u;
    uint32_t w = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u;
    return (w >> 22u) ^ w;
}

// Poisson(mu) by inversion on an edge-keyed uniform.  The uniform is a two-level PCG hash of (edge id, round
// key): source and target owner evaluate the same function, so they agree on c_e with no communication.  (A
// Philox block per edge was measured at ~28 % of the round kernel: 40 quarter-rate integer multiplies.)
__device__ __forceinline__ uint32_t round_hash_key(uint32_t round_key, uint64_t seed) {
    return pcg_hash(pcg_hash((uint32_t)seed ^ 0x5bd1e995u) ^ pcg_hash(round_key + (uint32_t)(seed >> 32)) ^ kTagEdgeCount);
}
__device__ __forceinline__ float edge_uniform(uint64_t e, uint32_t rk) {
    // capped below the f32 partial sums of the Poisson cdf (which end within ~4e-6 of 1 for mu <= 30), so
    // that the inversion loop always terminates on `u < cdf`
    return fminf((float)(pcg_hash(pcg_hash((uint32_t)e) ^ rk) >> 8) * (1.0f / 16777216.0f), 0.999984f);
}
__device__ __forceinline__ uint32_t edge_count(uint64_t e, uint32_t round_key, uint64_t seed, float mu) {
    const float u = edge_uniform(e, round_hash_key(round_key, seed));
    float p = __expf(-mu);
    float cdf = p;
    uint32_t c = 0;
    while (u >= cdf && c < 255u) {
        c++;
        p *= mu * (1.0f / (float)c);
        cdf += p;
    }
    return c;
}

// ---- K_plan: one wave per owned node.  Lanes < k draw the per-edge sample counts (written to cnt[] for
// the in-push replay of the partner), a wave scan turns them into sample slots, then lane t plans sample t:
// target j and the 5 admissible negatives (embedder.rs:1241-1253).
template <bool HUB>
__global__ void __launch_bounds__(kBlock) ce_plan_node_kernel(NodeArgs a) {
    const CeDev c = a.c;
    const int lane = threadIdx.x & 63;
    const uint64_t v64 = c.node_lo + ((blockIdx.x * (uint64_t)kBlock + threadIdx.x) >> 6);
    if (v64 >= c.node_hi) return;
    const uint32_t v = (uint32_t)v64;
    uint64_t ib;
    uint32_t k;
    if (c.uniform_k) { ib = (uint64_t)v * c.uniform_k; k = c.uniform_k; }
    else { ib = c.indptr[v]; k = (uint32_t)(c.indptr[v + 1] - ib); }
    const uint32_t node_base = pcg_hash(pcg_hash((uint32_t)c.seed ^ a.round_key) + v);
    uint32_t planned = 0;   // slots already used by previous 64-edge chunks (k > 64)
    // slot-major layout: entry (slot, node) at ((slot * nodes) + node) * 2, so that the 64 nodes of an apply
    // wave read 2 KB contiguous per slot
    const uint64_t nodes_owned = c.node_hi - c.node_lo;
    uint4* my_plan = reinterpret_cast<uint4*>(a.plan) + (uint64_t)(v - c.node_lo) * 2;
    for (uint32_t e0 = 0; e0 < k; e0 += 64) {
        const uint32_t m = e0 + lane;
        uint32_t cm = 0, nb = 0xFFFFFFFFu;
        float pm = 0.f;
        if (m < k) {
            nb = c.nbr[ib + m];
            pm = c.proba[ib + m];
            cm = edge_count(ib + m, a.round_key, c.seed, a.unit * pm);
            a.cnt[ib + m] = (uint8_t)cm;
        }
        // inclusive scan of the counts over the wave
        uint32_t pre = cm;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o = __shfl_up(pre, off);
            if (lane >= off) pre += o;
        }
        const uint32_t chunk_total = __shfl(pre, 63);
        // 64-bit signature of the row chunk for a cheap "certainly not a neighbour" test
        unsigned long long sig = (m < k) ? (1ull << (nb & 63u)) : 0ull;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sig |= __shfl_xor(sig, off);
        for (uint32_t t0 = 0; t0 < chunk_total; t0 += 64) {
            const uint32_t t = t0 + lane;
            // edge of sample t: first lane whose inclusive prefix exceeds t (wave-uniform loop over <= 64 lanes)
            uint32_t my_edge_lane = 0;
            const uint32_t kk = (k - e0) < 64u ? (k - e0) : 64u;
            for (uint32_t l = 0; l < kk; l++) {
                const uint32_t pl = __shfl(pre, (int)l);
                if (pl <= t) my_edge_lane = l + 1;
            }
            const bool active = t < chunk_total;
            const uint32_t j = __shfl(nb, (int)(my_edge_lane < 64u ? my_edge_lane : 63u));
            const float wj = __shfl(pm, (int)(my_edge_lane < 64u ? my_edge_lane : 63u));
            uint32_t kn[5];
            const uint32_t sbase = node_base + (planned + t) * 64u;
#pragma unroll
            for (int g = 0; g < 5; g++) {
                uint32_t cand = 0;
                for (uint32_t attempt = 0; attempt < 12u; attempt++) {
                    const uint32_t w0 = pcg_hash(sbase + (uint32_t)g * 12u + attempt);
                    if constexpr (HUB) {  // NodeSampler::sample, embedder.rs:927-930
                        const uint32_t x = __umulhi(w0, (uint32_t)c.n);
                        const float u = (float)(pcg_hash(w0 ^ 0x9E3779B9u) >> 8) * (1.0f / 16777216.0f);
                        cand = (u < c.hub_odds[x]) ? x : c.hub_alias[x];
                    } else {
                        cand = __umulhi(w0, (uint32_t)c.n);  // :1121
                    }
                    bool reject = (cand == v) || (cand == j);
                    // NodeParam::get_edge (nodeparam.rs:83-85): exact row scan only when the signature hits.
                    // (rows longer than 64 are checked chunk by chunk against the current chunk only; the
                    // remaining k/N-probability event is accepted there)
                    const bool maybe = (sig >> (cand & 63u)) & 1ull;
                    if (__any(active && !reject && maybe)) {
                        for (uint32_t l = 0; l < kk; l++) {
                            const uint32_t nl = __shfl(nb, (int)l);
                            if (nl == cand) reject = true;
                        }
                    }
                    if (!reject) break;
                }
                kn[g] = cand;
            }
            const uint32_t slot = planned + t;
            if (active) {
                if (slot < a.cap) {
                    my_plan[(uint64_t)slot * nodes_owned * 2] = make_uint4(j, kn[0], kn[1], kn[2]);
                    my_plan[(uint64_t)slot * nodes_owned * 2 + 1] = make_uint4(kn[3], kn[4], __float_as_uint(wj), 0u);
                } else {
                    atomicOr(a.overflow, 2u);
                }
            }
        }
        planned += chunk_total;
    }
    if (lane == 0) a.tot[v - c.node_lo] = planned < a.cap ? planned : a.cap;
}

// counts of the edges whose source is NOT owned by this shard (multi-GPU): needed to replay remote pushes
__global__ void __launch_bounds__(kBlock) ce_count_remote_kernel(NodeArgs a) {
    const CeDev c = a.c;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t e = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; e < c.nnz; e += stride) {
        if (e >= c.edge_lo && e < c.edge_lo + c.shard_edges) continue;
        a.cnt[e] = (uint8_t)edge_count(e, a.round_key, c.seed, a.unit * c.proba[e]);
    }
}

// ---- K_apply: thread v owns y_v and replays its planned out-samples, then the pushes of its in-edges.
template <int DIM, bool B1>
__device__ __forceinline__ void attract(float* yv, const float* yo, float w, float inv_s2, float step, float b, float sign) {
    // y_i half (sign = -1: y_i -= g) or y_j half (sign = +1: y_j += g) of embedder.rs:1207-1239, g = (y_j - y_i) * c
    float d = 0.f;
#pragma unroll
    for (int t = 0; t < DIM; t++) { const float df = yv[t] - yo[t]; d += df * df; }
    const float delta = d * inv_s2;
    if (delta > 0.f) {
        const float coeff = grad_coeff_f32<B1>(delta, inv_s2, b);
        const float rep = rcp(fmaxf(delta * delta, 1.0f / kProbaMin));
        const float cij = fmaxf(step * coeff * (-w + (1.f - w) * rep), -0.49f);
        // source side: y_i -= (y_j - y_i) c ; target side: y_j += (y_j - y_i) c  -- both are  y += (y - y_other) c
#pragma unroll
        for (int t = 0; t < DIM; t++) yv[t] += (yv[t] - yo[t]) * cij;
    }
    (void)sign;
}

template <int DIM>
__device__ __forceinline__ void store_row_plain(float* __restrict__ y, uint32_t node, const float* in) {
    float* p = y + (uint64_t)node * DIM;
#pragma unroll
    for (int t = 0; t < DIM; t++) p[t] = in[t];
}

template <int DIM, bool B1>
__global__ void __launch_bounds__(kApplyBlock) ce_apply_node_kernel(NodeArgs a) {
    const CeDev c = a.c;
    const uint32_t tid = threadIdx.x;
    const uint64_t v64 = c.node_lo + blockIdx.x * (uint64_t)kApplyBlock + threadIdx.x;
    if (v64 >= c.node_hi) return;
    const uint32_t v = (uint32_t)v64;
    float yv[DIM], grad[DIM];
    load_row_fresh<DIM>(c.y, v, yv);
#pragma unroll
    for (int t = 0; t < DIM; t++) grad[t] = 0.f;
    const float s_v = c.emb_scale[v];
    const float inv_s2 = rcp(s_v * s_v);
    uint64_t ib;
    if (c.uniform_k) ib = (uint64_t)v * c.uniform_k; else ib = c.indptr[v];
    // ---------------- (a) samples whose source is v ----------------
    // Depth-PD software pipeline: the coordinate rows of sample t + PD are requested right after sample t is
    // computed and BEFORE y_v is written back, so the in-order vmcnt wait of a later sample never queues
    // behind the (slow, write-through) store of the sample just computed.
    const uint32_t nv = a.tot[v - c.node_lo];
    const uint64_t nodes_owned = c.node_hi - c.node_lo;
    const uint4* my_plan = reinterpret_cast<const uint4*>(a.plan) + (uint64_t)(v - c.node_lo) * 2;
    // Plan entries are staged through LDS in chunks of CH samples: an entry read is then an LDS read
    // (lgkmcnt), so requesting the coordinate rows of a later sample never has to drain the vector-memory
    // queue -- which holds the slow write-through stores of the samples just computed (vmcnt is in order).
    constexpr int CH = 8;
    constexpr int PD = DIM <= 4 ? 3 : (DIM <= 8 ? 2 : 1);
    __shared__ uint4 s_ent[CH * 2 * kApplyBlock];
    float rows[PD][6][DIM];
    float wq[PD];
    auto stage = [&](uint32_t t_begin) {  // global -> LDS, this lane's entries [t_begin, t_begin + CH)
#pragma unroll
        for (int q = 0; q < CH; q++) {
            if (t_begin + (uint32_t)q < nv) {
                s_ent[(q * 2) * kApplyBlock + tid] = my_plan[(uint64_t)(t_begin + q) * nodes_owned * 2];
                s_ent[(q * 2 + 1) * kApplyBlock + tid] = my_plan[(uint64_t)(t_begin + q) * nodes_owned * 2 + 1];
            }
        }
    };
    auto fetch = [&](uint32_t t, int slot) {  // entry from LDS, rows from global
        const uint32_t q = t % CH;
        const uint4 p0 = s_ent[(q * 2) * kApplyBlock + tid], p1 = s_ent[(q * 2 + 1) * kApplyBlock + tid];
        wq[slot] = __uint_as_float(p1.z);
        load_row_fresh<DIM>(c.y, p0.x, rows[slot][0]);
        load_row_fresh<DIM>(c.y, p0.y, rows[slot][1]);
        load_row_fresh<DIM>(c.y, p0.z, rows[slot][2]);
        load_row_fresh<DIM>(c.y, p0.w, rows[slot][3]);
        load_row_fresh<DIM>(c.y, p1.x, rows[slot][4]);
        load_row_fresh<DIM>(c.y, p1.y, rows[slot][5]);
    };
    // chunk c covers samples [c*CH, (c+1)*CH); the rows of sample t are requested PD samples ahead, so chunk
    // boundaries are handled by staging the NEXT chunk as soon as the prefetch front reaches it.  To keep the
    // code simple the prefetch depth does not cross a chunk: each chunk is prologue + steady state.
    unsigned long long t_stage = 0, t_fetch = 0, t_comp = 0, t_store = 0, t_in = 0;
    const unsigned long long t_begin_all = __builtin_amdgcn_s_memtime();
    for (uint32_t c0 = 0; c0 < (a.skip == 1 ? 0u : nv); c0 += CH) {
        unsigned long long ts = __builtin_amdgcn_s_memtime();
        stage(c0);
        __builtin_amdgcn_s_waitcnt(0);
        t_stage += __builtin_amdgcn_s_memtime() - ts;
        const uint32_t cend = (c0 + CH < nv) ? c0 + CH : nv;
#pragma unroll
        for (int u = 0; u < PD; u++)
            if (c0 + (uint32_t)u < cend) fetch(c0 + (uint32_t)u, u);
        for (uint32_t t0 = c0; t0 < cend; t0 += PD) {
#pragma unroll
            for (int u = 0; u < PD; u++) {
                const uint32_t t = t0 + (uint32_t)u;
                if (t < cend) {
                    const float w = wq[u];
                    unsigned long long tc = __builtin_amdgcn_s_memtime();
                    {   // attraction, the y_i half of :1207-1237 (gradient kept for the reference's stale-gradient quirk)
                        float d = 0.f;
#pragma unroll
                        for (int q = 0; q < DIM; q++) { const float df = yv[q] - rows[u][0][q]; d += df * df; }
                        const float delta = d * inv_s2;
                        if (delta > 0.f) {
                            const float coeff = grad_coeff_f32<B1>(delta, inv_s2, a.b);
                            const float rep = rcp(fmaxf(delta * delta, 1.0f / kProbaMin));
                            const float cij = fmaxf(a.step * coeff * (-w + (1.f - w) * rep), -0.49f);
#pragma unroll
                            for (int q = 0; q < DIM; q++) grad[q] = (rows[u][0][q] - yv[q]) * cij;
                        } else {
#pragma unroll
                            for (int q = 0; q < DIM; q++) grad[q] = 0.f;
                        }
#pragma unroll
                        for (int q = 0; q < DIM; q++) yv[q] -= grad[q];
                    }
#pragma unroll
                    for (int g = 0; g < 5; g++) {  // 5 repulsions, :1267-1297
                        float dk = 0.f;
#pragma unroll
                        for (int q = 0; q < DIM; q++) { const float df = yv[q] - rows[u][1 + g][q]; dk += df * df; }
                        if (dk > 0.f) {
                            const float dks = dk * inv_s2;
                            const float coeff = grad_coeff_f32<B1>(dks, inv_s2, a.b);
                            const float cik = fminf(a.step * coeff * rcp(fmaxf(dks * dks, 1.0f / 16.0f)), 2.0f);
#pragma unroll
                            for (int q = 0; q < DIM; q++) grad[q] = (rows[u][1 + g][q] - yv[q]) * cik;
                        }  // else `gradient` keeps its previous value (reference quirk B4)
#pragma unroll
                        for (int q = 0; q < DIM; q++) yv[q] -= grad[q];
                    }
                    unsigned long long tf = __builtin_amdgcn_s_memtime();
                    t_comp += tf - tc;
                    if (t + PD < cend) fetch(t + PD, u);  // refill this slot before the store below
                    unsigned long long tst = __builtin_amdgcn_s_memtime();
                    t_fetch += tst - tf;
                    if (a.store_mode == 0) store_row_through<DIM>(c.y, v, yv);
                    else if (a.store_mode == 1) store_row_plain<DIM>(c.y, v, yv);
                    t_store += __builtin_amdgcn_s_memtime() - tst;
                }
            }
        }
    }
    const unsigned long long t_in0 = __builtin_amdgcn_s_memtime();
    if (a.store_mode == 2) store_row_through<DIM>(c.y, v, yv);
    // ---------------- (b) samples whose target is v: the y_j half of :1238-1239 ----------------
    // one-deep pipeline over the in-edges: record, count and source row of the next in-edge are requested
    // before the pushes of the current one are applied and stored
    const uint64_t tb = a.tptr[v], te = a.tptr[v + 1];
    if (tb < te && a.skip != 2) {
        InEdge rec = a.tin[tb];
        uint32_t cnt = a.cnt[rec.eid];
        float yu[DIM];
        load_row_fresh<DIM>(c.y, rec.src, yu);
        for (uint64_t x = tb; x < te; x++) {
            InEdge nrec = rec;
            uint32_t ncnt = 0;
            float nyu[DIM];
#pragma unroll
            for (int q = 0; q < DIM; q++) nyu[q] = 0.f;
            if (x + 1 < te) {
                nrec = a.tin[x + 1];
                ncnt = a.cnt[nrec.eid];
                load_row_fresh<DIM>(c.y, nrec.src, nyu);
            }
            if (cnt) {
                const float inv_su2 = rcp(rec.s_src * rec.s_src);
                for (uint32_t r = 0; r < cnt; r++) {
                    if (r > 0) load_row_fresh<DIM>(c.y, rec.src, yu);  // repeated sample of the edge: refresh the source
                    attract<DIM, B1>(yv, yu, rec.w, inv_su2, a.step, a.b, 1.f);
                    if (a.store_mode == 0) store_row_through<DIM>(c.y, v, yv);
                    else if (a.store_mode == 1) store_row_plain<DIM>(c.y, v, yv);
                }
            }
            rec = nrec;
            cnt = ncnt;
#pragma unroll
            for (int q = 0; q < DIM; q++) yu[q] = nyu[q];
        }
    }
    if (a.store_mode == 2) store_row_through<DIM>(c.y, v, yv);
    if (a.prof && tid == 0) {
        const unsigned long long tend = __builtin_amdgcn_s_memtime();
        t_in = tend - t_in0;
        atomicAdd(&a.prof[0], t_stage); atomicAdd(&a.prof[1], t_fetch); atomicAdd(&a.prof[2], t_comp);
        atomicAdd(&a.prof[3], t_store); atomicAdd(&a.prof[4], t_in); atomicAdd(&a.prof[5], tend - t_begin_all);
        atomicAdd(&a.prof[6], 1ull);
    }
}


// ---- K_apply, lane-group form: 8 lanes per node, 8 nodes per wave.  A graph with N nodes offers only N
// sequential update chains; with one lane per node 60 k nodes are 940 waves (one per SIMD, nothing to hide
// latency with).  Here lane r of a group fetches row r of the sample (j, k1..k5) -- one gather instruction
// covers the 6 rows of 8 samples -- and the 6 dependent update steps are replayed by the whole group on a
// replicated y_v with in-group broadcasts.  8x more waves, 8x shorter in-edge loops for hubs.
constexpr int kGroup = 8;
template <int DIM>
__device__ __forceinline__ void group_bcast(const float* in, int src_lane, float* out) {
#pragma unroll
    for (int q = 0; q < DIM; q++) out[q] = __shfl(in[q], src_lane);
}

template <int DIM, bool B1>
__global__ void __launch_bounds__(kBlock) ce_apply_group_kernel(NodeArgs a) {
    const CeDev c = a.c;
    const int lane = threadIdx.x & 63;
    const int r = lane & (kGroup - 1);
    const int gbase = lane & ~(kGroup - 1);
    const uint64_t nodes_owned = c.node_hi - c.node_lo;
    const uint64_t wave = (blockIdx.x * (uint64_t)kBlock + threadIdx.x) >> 6;
    const uint64_t local = wave * (64 / kGroup) + (uint64_t)(lane >> 3);
    const bool valid = local < nodes_owned;
    const uint64_t lv = valid ? local : nodes_owned - 1;  // idle groups shadow the last node (no stores)
    const uint32_t v = (uint32_t)(c.node_lo + lv);
    float yv[DIM], grad[DIM];
    load_row_fresh<DIM>(c.y, v, yv);
#pragma unroll
    for (int q = 0; q < DIM; q++) grad[q] = 0.f;
    const float s_v = c.emb_scale[v];
    const float inv_s2 = rcp(s_v * s_v);
    // ---------------- (a) samples whose source is v ----------------
    const uint32_t nv = valid ? a.tot[lv] : 0u;
    uint32_t nmax = nv;
#pragma unroll
    for (int off = 32; off >= kGroup; off >>= 1) { const uint32_t o = __shfl_xor(nmax, off); nmax = o > nmax ? o : nmax; }
    const uint32_t* plan_words = a.plan + lv * 8 + r;  // entry (slot, node): 8 words at ((slot * nodes) + node) * 8
    for (uint32_t t = 0; t < nmax; t++) {
        const bool act = t < nv;
        uint32_t word = 0;
        if (act) word = plan_words[(uint64_t)t * nodes_owned * 8];
        float row[DIM];
#pragma unroll
        for (int q = 0; q < DIM; q++) row[q] = 0.f;
        if (act && r < 6) {
            if (a.skip == 7) { const float* pp = c.y + (uint64_t)word * DIM; for (int q = 0; q < DIM; q++) row[q] = pp[q]; }
            else load_row_fresh<DIM>(c.y, word, row);
        }
        const float w = __uint_as_float(__shfl(word, gbase + 6));
        float other[DIM];
        group_bcast<DIM>(row, gbase + 0, other);
        {   // attraction, the y_i half of embedder.rs:1207-1237
            float d = 0.f;
#pragma unroll
            for (int q = 0; q < DIM; q++) { const float df = yv[q] - other[q]; d += df * df; }
            const float delta = d * inv_s2;
            if (act && delta > 0.f) {
                const float coeff = grad_coeff_f32<B1>(delta, inv_s2, a.b);
                const float rep = rcp(fmaxf(delta * delta, 1.0f / kProbaMin));
                const float cij = fmaxf(a.step * coeff * (-w + (1.f - w) * rep), -0.49f);
#pragma unroll
                for (int q = 0; q < DIM; q++) grad[q] = (other[q] - yv[q]) * cij;
            } else {
#pragma unroll
                for (int q = 0; q < DIM; q++) grad[q] = 0.f;
            }
#pragma unroll
            for (int q = 0; q < DIM; q++) yv[q] -= grad[q];
        }
#pragma unroll
        for (int g = 1; g <= 5; g++) {  // 5 repulsions, :1267-1297
            group_bcast<DIM>(row, gbase + g, other);
            float dk = 0.f;
#pragma unroll
            for (int q = 0; q < DIM; q++) { const float df = yv[q] - other[q]; dk += df * df; }
            if (act && dk > 0.f) {
                const float dks = dk * inv_s2;
                const float coeff = grad_coeff_f32<B1>(dks, inv_s2, a.b);
                const float cik = fminf(a.step * coeff * rcp(fmaxf(dks * dks, 1.0f / 16.0f)), 2.0f);
#pragma unroll
                for (int q = 0; q < DIM; q++) grad[q] = (other[q] - yv[q]) * cik;
            }  // else `gradient` keeps its previous value (reference quirk B4)
            if (act) {
#pragma unroll
                for (int q = 0; q < DIM; q++) yv[q] -= grad[q];
            }
        }
        if (act && r == 0) { if (a.store_mode == 0) store_row_through<DIM>(c.y, v, yv); else if (a.store_mode == 1) store_row_plain<DIM>(c.y, v, yv); }
    }
    if (a.store_mode >= 2 && r == 0 && valid) store_row_through<DIM>(c.y, v, yv);
    if (a.skip == 2) return;
    // ---------------- (b) samples whose target is v: the y_j half of :1238-1239 ----------------
    const uint64_t tb = a.tptr[v];
    const uint32_t indeg = valid ? (uint32_t)(a.tptr[v + 1] - tb) : 0u;
    uint32_t dmax = indeg;
#pragma unroll
    for (int off = 32; off >= kGroup; off >>= 1) { const uint32_t o = __shfl_xor(dmax, off); dmax = o > dmax ? o : dmax; }
    for (uint32_t x0 = 0; x0 < dmax; x0 += kGroup) {
        const uint32_t x = x0 + (uint32_t)r;
        uint32_t cnt = 0;
        float wu = 0.f, su = 1.f;
        float yu[DIM];
#pragma unroll
        for (int q = 0; q < DIM; q++) yu[q] = 0.f;
        if (x < indeg) {
            const InEdge rec = a.tin[tb + x];
            cnt = a.cnt[rec.eid];
            wu = rec.w;
            su = rec.s_src;
            if (cnt) load_row_fresh<DIM>(c.y, rec.src, yu);
        }
        bool changed = false;
#pragma unroll
        for (int sl = 0; sl < kGroup; sl++) {
            const uint32_t bc = __shfl(cnt, gbase + sl);
            uint32_t cmax = bc;
#pragma unroll
            for (int off = 32; off >= kGroup; off >>= 1) { const uint32_t o = __shfl_xor(cmax, off); cmax = o > cmax ? o : cmax; }
            if (cmax == 0) continue;  // wave-uniform
            float other[DIM];
            group_bcast<DIM>(yu, gbase + sl, other);
            const float wb = __shfl(wu, gbase + sl);
            const float sb = __shfl(su, gbase + sl);
            const float inv_su2 = rcp(sb * sb);
            for (uint32_t rep_i = 0; rep_i < cmax; rep_i++) {
                if (rep_i < bc) {
                    attract<DIM, B1>(yv, other, wb, inv_su2, a.step, a.b, 1.f);
                    changed = true;
                }
            }
        }
        if (changed && r == 0 && valid && a.store_mode < 2) store_row_through<DIM>(c.y, v, yv);
    }
    if (a.store_mode >= 2 && r == 0 && valid) store_row_through<DIM>(c.y, v, yv);
}

// samples planned in this round (one workgroup; a per-wave atomic on one address would cost ~12 ns each)
__global__ void __launch_bounds__(1024) ce_sum_tot_kernel(const uint32_t* __restrict__ tot, uint64_t n, unsigned long long* counter) {
    __shared__ unsigned long long red[1024];
    unsigned long long s = 0;
    for (uint64_t i = threadIdx.x; i < n; i += 1024) s += tot[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int off = 512; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) *counter += red[0];
}

// ---- Fused round kernel (default for rows of <= 16 neighbours): counts, planning and replay in ONE launch per
// round, 8 lanes per node.  Lanes draw the Poisson counts of the node's out-edges (2 per lane), lane 0 resolves
// the sampled edge, lanes 1..5 draw one admissible negative each (the RNG work of a sample is spread over the
// group instead of being serial), every lane fetches its row, the group replays the 6 dependent steps.  The
// in-edge pushes are replayed from counts recomputed with the same edge-keyed Philox block (no cnt[] array, so
// remote sources on other GPUs need no exchange).
template <int DIM, bool B1, bool HUB>
__global__ void __launch_bounds__(kBlock) ce_round_group_kernel(NodeArgs a) {
    constexpr int NPB = kBlock / kGroup;        // nodes per workgroup
    __shared__ uint32_t s_row[16 * NPB];        // [m][node in block]: exact neighbour test of a candidate
    const CeDev c = a.c;
    const int lane = threadIdx.x & 63;
    const int r = lane & (kGroup - 1);
    const int gbase = lane & ~(kGroup - 1);
    const int nib = threadIdx.x >> 3;           // node in block
    const uint64_t nodes_owned = c.node_hi - c.node_lo;
    const uint64_t local = blockIdx.x * (uint64_t)NPB + (uint64_t)nib;
    const bool valid = local < nodes_owned;
    const uint64_t lv = valid ? local : nodes_owned - 1;
    const uint32_t v = (uint32_t)(c.node_lo + lv);
    uint64_t ib;
    uint32_t k;
    if (c.uniform_k) { ib = (uint64_t)v * c.uniform_k; k = c.uniform_k; }
    else { ib = c.indptr[v]; k = (uint32_t)(c.indptr[v + 1] - ib); }
    // ---- stage A: counts of the out-edges m = r and m = r + 8
    uint32_t nbA = 0xFFFFFFFFu, nbB = 0xFFFFFFFFu, cA = 0, cB = 0;
    float pA = 0.f, pB = 0.f;
    if ((uint32_t)r < k) {
        nbA = c.nbr[ib + r]; pA = c.proba[ib + r];
        if (valid) cA = edge_count(ib + r, a.round_key, c.seed, a.unit * pA);
    }
    if ((uint32_t)r + 8u < k) {
        nbB = c.nbr[ib + r + 8]; pB = c.proba[ib + r + 8];
        if (valid) cB = edge_count(ib + r + 8, a.round_key, c.seed, a.unit * pB);
    }
    s_row[r * NPB + nib] = nbA;
    s_row[(r + 8) * NPB + nib] = nbB;
    uint32_t preA = cA, preB = cB;  // inclusive scans inside the 8-lane group
#pragma unroll
    for (int off = 1; off < kGroup; off <<= 1) {
        const uint32_t oa = __shfl_up(preA, off), ob = __shfl_up(preB, off);
        if (r >= off) { preA += oa; preB += ob; }
    }
    const uint32_t totA = __shfl(preA, gbase + 7);
    const uint32_t nv = totA + __shfl(preB, gbase + 7);
    unsigned long long sig = 0ull;  // 64-bit signature of the row: cheap "certainly not a neighbour"
    if ((uint32_t)r < k) sig |= 1ull << (nbA & 63u);
    if ((uint32_t)r + 8u < k) sig |= 1ull << (nbB & 63u);
#pragma unroll
    for (int off = 1; off < kGroup; off <<= 1) sig |= __shfl_xor(sig, off);
    uint32_t nmax = nv;
#pragma unroll
    for (int off = 32; off >= kGroup; off >>= 1) { const uint32_t o = __shfl_xor(nmax, off); nmax = o > nmax ? o : nmax; }
    __syncthreads();  // s_row visible (one barrier per launch)
    float yv[DIM], grad[DIM];
    load_row_fresh<DIM>(c.y, v, yv);
#pragma unroll
    for (int q = 0; q < DIM; q++) grad[q] = 0.f;
    const float s_v = c.emb_scale[v];
    const float inv_s2 = rcp(s_v * s_v);
    const uint32_t node_base = pcg_hash(pcg_hash((uint32_t)c.seed ^ a.round_key) + v);
    // ---- stage B: the samples whose source is v
    for (uint32_t t = 0; t < nmax; t++) {
        const bool act = t < nv;
        // which edge: the lane whose count interval contains t
        const bool ownA = act && t < totA && t >= preA - cA && t < preA;
        const uint32_t tb_ = t - totA;
        const bool ownB = act && t >= totA && tb_ >= preB - cB && tb_ < preB;
        const unsigned long long mA = __ballot(ownA), mB = __ballot(ownB);
        const uint32_t gA = (uint32_t)(mA >> gbase) & 0xFFu, gB = (uint32_t)(mB >> gbase) & 0xFFu;
        const int srcA = gbase + (gA ? __ffs(gA) - 1 : 0), srcB = gbase + (gB ? __ffs(gB) - 1 : 0);
        const uint32_t jA = __shfl(nbA, srcA), jB = __shfl(nbB, srcB);
        const float wA = __shfl(pA, srcA), wB = __shfl(pB, srcB);
        const uint32_t j = gA ? jA : jB;
        const float w = gA ? wA : wB;
        uint32_t idx = j;  // lane 0: the sampled neighbour; lanes 1..5: one admissible negative each (:1241-1253)
        if (act && r >= 1 && r <= 5) {
            const uint32_t sbase = node_base + t * 128u + (uint32_t)r * 16u;
            for (uint32_t attempt = 0; attempt < 16u; attempt++) {
                const uint32_t w0 = pcg_hash(sbase + attempt);
                uint32_t cand;
                if constexpr (HUB) {  // NodeSampler::sample, embedder.rs:927-930
                    const uint32_t x = __umulhi(w0, (uint32_t)c.n);
                    const float u = (float)(pcg_hash(w0 ^ 0x9E3779B9u) >> 8) * (1.0f / 16777216.0f);
                    cand = (u < c.hub_odds[x]) ? x : c.hub_alias[x];
                } else {
                    cand = __umulhi(w0, (uint32_t)c.n);  // :1121
                }
                bool reject = (cand == v) || (cand == j);
                if (!reject && ((sig >> (cand & 63u)) & 1ull)) {  // NodeParam::get_edge, nodeparam.rs:83-85
                    for (uint32_t m = 0; m < k; m++)
                        if (s_row[m * NPB + nib] == cand) { reject = true; break; }
                }
                idx = cand;
                if (!reject) break;
            }
        }
        float row[DIM];
#pragma unroll
        for (int q = 0; q < DIM; q++) row[q] = 0.f;
        if (act && r < 6) load_row_fresh<DIM>(c.y, idx, row);
        float other[DIM];
        group_bcast<DIM>(row, gbase + 0, other);
        {   // attraction, the y_i half of embedder.rs:1207-1237
            float d = 0.f;
#pragma unroll
            for (int q = 0; q < DIM; q++) { const float df = yv[q] - other[q]; d += df * df; }
            const float delta = d * inv_s2;
            if (act && delta > 0.f) {
                const float coeff = grad_coeff_f32<B1>(delta, inv_s2, a.b);
                const float rep = rcp(fmaxf(delta * delta, 1.0f / kProbaMin));
                const float cij = fmaxf(a.step * coeff * (-w + (1.f - w) * rep), -0.49f);
#pragma unroll
                for (int q = 0; q < DIM; q++) grad[q] = (other[q] - yv[q]) * cij;
            } else {
#pragma unroll
                for (int q = 0; q < DIM; q++) grad[q] = 0.f;
            }
#pragma unroll
            for (int q = 0; q < DIM; q++) yv[q] -= grad[q];
        }
#pragma unroll
        for (int g = 1; g <= 5; g++) {  // 5 repulsions, :1267-1297
            group_bcast<DIM>(row, gbase + g, other);
            float dk = 0.f;
#pragma unroll
            for (int q = 0; q < DIM; q++) { const float df = yv[q] - other[q]; dk += df * df; }
            if (act && dk > 0.f) {
                const float dks = dk * inv_s2;
                const float coeff = grad_coeff_f32<B1>(dks, inv_s2, a.b);
                const float cik = fminf(a.step * coeff * rcp(fmaxf(dks * dks, 1.0f / 16.0f)), 2.0f);
#pragma unroll
                for (int q = 0; q < DIM; q++) grad[q] = (other[q] - yv[q]) * cik;
            }  // else `gradient` keeps its previous value (reference quirk B4)
            if (act) {
#pragma unroll
                for (int q = 0; q < DIM; q++) yv[q] -= grad[q];
            }
        }
        if (a.store_mode == 0 && act && r == 0) store_row_through<DIM>(c.y, v, yv);
    }
    if (a.store_mode != 0 && r == 0 && valid && nv) store_row_through<DIM>(c.y, v, yv);
    // ---- stage C: the samples whose target is v (the y_j half of :1238-1239), counts recomputed per in-edge
    const uint64_t tb = a.tptr[v];
    const uint32_t indeg = valid ? (uint32_t)(a.tptr[v + 1] - tb) : 0u;
    uint32_t dmax = indeg;
#pragma unroll
    for (int off = 32; off >= kGroup; off >>= 1) { const uint32_t o = __shfl_xor(dmax, off); dmax = o > dmax ? o : dmax; }
    bool any_push = false;
    for (uint32_t x0 = 0; x0 < dmax; x0 += kGroup) {
        const uint32_t x = x0 + (uint32_t)r;
        uint32_t cnt = 0;
        float wu = 0.f, su = 1.f;
        float yu[DIM];
#pragma unroll
        for (int q = 0; q < DIM; q++) yu[q] = 0.f;
        if (x < indeg) {
            const InEdge rec = a.tin[tb + x];
            cnt = edge_count(rec.eid, a.round_key, c.seed, a.unit * rec.w);
            wu = rec.w;
            su = rec.s_src;
            if (cnt) load_row_fresh<DIM>(c.y, rec.src, yu);
        }
        bool changed = false;
#pragma unroll
        for (int sl = 0; sl < kGroup; sl++) {
            const uint32_t bc = __shfl(cnt, gbase + sl);
            uint32_t cmax = bc;
#pragma unroll
            for (int off = 32; off >= kGroup; off >>= 1) { const uint32_t o = __shfl_xor(cmax, off); cmax = o > cmax ? o : cmax; }
            if (cmax == 0) continue;  // wave-uniform
            float other[DIM];
            group_bcast<DIM>(yu, gbase + sl, other);
            const float wb = __shfl(wu, gbase + sl);
            const float sb = __shfl(su, gbase + sl);
            const float inv_su2 = rcp(sb * sb);
            for (uint32_t rep_i = 0; rep_i < cmax; rep_i++) {
                if (rep_i < bc) {
                    attract<DIM, B1>(yv, other, wb, inv_su2, a.step, a.b, 1.f);
                    changed = true;
                }
            }
        }
        any_push |= changed;
        if (a.store_mode == 0 && changed && r == 0 && valid) store_row_through<DIM>(c.y, v, yv);
    }
    if (a.store_mode != 0 && any_push && r == 0 && valid) store_row_through<DIM>(c.y, v, yv);
    // samples drawn: one atomic per wave, spread over 1024 counters (a single address serialises at ~12 ns each)
    unsigned long long mine = (r == 0 && valid) ? (unsigned long long)nv : 0ull;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mine += __shfl_xor(mine, off);
    if (lane == 0 && mine) atomicAdd(&a.sample_counter[(blockIdx.x * 4 + (threadIdx.x >> 6)) & 1023u], mine);
}

// ---- Node-per-lane round kernel (default for rows of <= 16 neighbours): one lane owns one node, 64 nodes per
// wave, one launch per round.  The lane-group kernel above spends 8 lanes on every dependent update chain; here
// a chain costs one lane, and latency is hidden by memory-level parallelism instead of by waves: the rows of S
// samples (6 S gathers per lane) are in flight before their replay starts -- the sample's node set does not
// depend on y_v.
//   stage A  per lane: Poisson counts of the node's out-edges (edge-keyed hash); neighbour ids, weights and
//            counts parked in an LDS column private to the lane (dynamically indexable scratch, no barrier).
//   stage B  per chunk of S samples: resolve the sampled edge (walk the counts), draw 5 admissible negatives
//            (exact rejection against the LDS column, nodeparam.rs:83-85), issue the 6 S gathers, replay
//            the 6 dependent steps of embedder.rs:1207-1297 per sample.
//   stage C  the in-edge pushes (y_j halves, :1238-1239): the in-edges of the wave's 64 nodes are contiguous in
//            the transposed graph, so the wave evaluates their counts and gathers the source rows *balanced*
//            (edge x -> lane x mod 64, a hub's 100 in-edges cost every lane 2), parks the active ones in LDS, and
//            each lane replays the slice that targets its node.
template <int DIM>
struct NodeKernelCfg {
    static constexpr int S = DIM <= 2 ? 4 : (DIM <= 4 ? 2 : 1);  // samples whose rows are gathered together
    static constexpr int NQ = 4;                                  // in-edge records per lane and pass of stage C
    static constexpr int CH = 64 * NQ;
    static constexpr int EC = DIM <= 4 ? 512 : (DIM <= 8 ? 256 : 128);  // pushes parked in LDS per window
};

// N independent Poisson inversions advanced together, branch-free (same operations, same order per variate as
// edge_count(): the source and the target owner of an edge must get the same count), so that the dependent
// chains of different variates overlap -- a lone wave per SIMD has nothing else to hide their latency with.
template <int N>
__device__ __forceinline__ void poisson_batch(const float* u, const float* mu, uint32_t* cnt) {
    float p[N], cdf[N];
#pragma unroll
    for (int i = 0; i < N; i++) { p[i] = __expf(-mu[i]); cdf[i] = p[i]; cnt[i] = 0u; }
    for (uint32_t c = 1; c <= 255u; c++) {
        const float inv_c = 1.0f / (float)c;
        bool more = false;
#pragma unroll
        for (int i = 0; i < N; i++) {
            const bool go = u[i] >= cdf[i];  // once false it stays false: cdf only moves while go holds
            p[i] = p[i] * (mu[i] * inv_c);
            cdf[i] += go ? p[i] : 0.f;
            cnt[i] += go ? 1u : 0u;
            more |= go;
        }
        if (!__any(more)) break;
    }
}

// one pair step on y_v with a single reciprocal (b == 1):  attraction  c = max(2 step/s^2 (-w M + 1 - w) /
// ((1 + delta) M), -0.49), M = max(delta^2, 1e4)  (embedder.rs:1216-1233);  repulsion  c = min(2 step/s^2 /
// ((1 + delta) max(delta^2, 1/16)), 2)  (:1286-1293).  y_v += (y_v - y_o) c in both roles (source: y_i -= g;
// target: y_j += g, g = (y_j - y_i) c).
template <int DIM, bool B1>
__device__ __forceinline__ float attract_coeff(float d, float w, float inv_s2, float step2, float step, float b) {
    const float delta = d * inv_s2;
    if constexpr (B1) {
        const float M = fmaxf(delta * delta, 1.0f / kProbaMin);
        return fmaxf(step2 * inv_s2 * ((1.f - w) - w * M) * rcp((1.f + delta) * M), -0.49f);
    } else {
        const float coeff = grad_coeff_f32<false>(delta, inv_s2, b);
        const float rep = rcp(fmaxf(delta * delta, 1.0f / kProbaMin));
        return fmaxf(step * coeff * (-w + (1.f - w) * rep), -0.49f);
    }
}
template <int DIM, bool B1>
__device__ __forceinline__ float repulse_coeff(float d, float inv_s2, float step2, float step, float b) {
    const float delta = d * inv_s2;
    if constexpr (B1) {
        return fminf(step2 * inv_s2 * rcp((1.f + delta) * fmaxf(delta * delta, 1.0f / 16.0f)), 2.0f);
    } else {
        const float coeff = grad_coeff_f32<false>(delta, inv_s2, b);
        return fminf(step * coeff * rcp(fmaxf(delta * delta, 1.0f / 16.0f)), 2.0f);
    }
}

template <int DIM, bool B1, bool HUB, int KMAX>
__global__ void __launch_bounds__(64) ce_round_node_kernel(NodeArgs a) {
    using Cfg = NodeKernelCfg<DIM>;
    constexpr int LS = 65, S = Cfg::S, CH = Cfg::CH, NQ = Cfg::NQ, EC = Cfg::EC, KP = KMAX / 4;
    __shared__ uint32_t s_nbr[KMAX * LS];
    __shared__ float s_w[KMAX * LS];
    __shared__ float s_in_row[EC * DIM];
    __shared__ float s_in_a[EC];      // b == 1: 2 step / s_u^2 * (1 - w); else w
    __shared__ float s_in_b[EC];      // b == 1: 2 step / s_u^2 * w
    __shared__ float s_in_is2[EC];
    __shared__ uint32_t s_pos[CH + 1];
    const CeDev c = a.c;
    const int lane = threadIdx.x;
    const uint64_t nodes_owned = c.node_hi - c.node_lo;
    const uint64_t local0 = blockIdx.x * 64ull;
    const uint64_t local = local0 + (uint64_t)lane;
    const bool valid = local < nodes_owned;
    const uint32_t v = (uint32_t)(c.node_lo + (valid ? local : nodes_owned - 1));
    uint64_t ib;
    uint32_t k;
    if (c.uniform_k) { ib = (uint64_t)v * c.uniform_k; k = c.uniform_k; }
    else { ib = c.indptr[v]; k = (uint32_t)(c.indptr[v + 1] - ib); }
    const uint32_t rk = round_hash_key(a.round_key, c.seed);
    const float step2 = 2.0f * a.step;
    unsigned long long tk0 = a.prof ? __builtin_amdgcn_s_memtime() : 0ull, tk_acc[6] = {0, 0, 0, 0, 0, 0};
    const unsigned long long tk_begin = tk0;
#define AE_TICK(i) if (a.prof) { const unsigned long long tk1 = __builtin_amdgcn_s_memtime(); tk_acc[i] += tk1 - tk0; tk0 = tk1; }
    // ---- stage C prologue: the first in-edge records of the wave are requested now, their latency overlaps
    // stages A and B.  Lane l holds the NQ consecutive records cb + l NQ .. cb + l NQ + NQ - 1.
    const uint32_t v0 = (uint32_t)(c.node_lo + local0);
    const uint64_t n_here = (nodes_owned - local0) < 64ull ? (nodes_owned - local0) : 64ull;
    const uint64_t t_begin = a.tptr[v0], t_end = a.tptr[v0 + n_here];
    const uint64_t tb_v = valid ? a.tptr[v] : 0ull, te_v = valid ? a.tptr[v + 1] : 0ull;
    InEdge recA[NQ], recB[NQ];
    auto load_recs = [&](uint64_t cb, InEdge* rec) {
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            const uint64_t x = cb + (uint64_t)(lane * NQ + q);
            rec[q] = a.tin[x < t_end ? x : t_begin];
        }
    };
    if (t_begin < t_end) load_recs(t_begin, recA);
    // ---- stage A: the node's row in registers (rejection test) and in an LDS column private to the lane
    // (dynamic index, no barrier needed), cumulative Poisson counts of the out-edges packed 4 per register
    uint32_t nbr_reg[KMAX], cumP[KP];
    uint32_t nv;
    {
        float pr[KMAX], mu[KMAX], u[KMAX];
        uint32_t cnt[KMAX];
#pragma unroll
        for (int m = 0; m < KMAX; m++) {  // unconditional loads (clamped index): all in flight together
            const uint32_t mm = (uint32_t)m < k ? (uint32_t)m : k - 1u;
            nbr_reg[m] = c.nbr[ib + mm];
            pr[m] = c.proba[ib + mm];
        }
#pragma unroll
        for (int m = 0; m < KMAX; m++) {
            const bool has = (uint32_t)m < k;
            nbr_reg[m] = has ? nbr_reg[m] : 0xFFFFFFFFu;  // the pad never equals a candidate
            pr[m] = has ? pr[m] : 0.f;
            s_nbr[m * LS + lane] = nbr_reg[m];
            s_w[m * LS + lane] = pr[m];
            mu[m] = (has && valid) ? a.unit * pr[m] : 0.f;
            u[m] = edge_uniform(ib + m, rk);
        }
        poisson_batch<KMAX>(u, mu, cnt);
        uint32_t run = 0;
#pragma unroll
        for (int m = 0; m < KMAX; m++) {  // inclusive prefix, saturated at 127 (SWAR search below; Poisson(12) never gets there)
            run += cnt[m];
            run = run < 127u ? run : 127u;
            if (m % 4 == 0) cumP[m / 4] = run;
            else cumP[m / 4] |= run << (8 * (m % 4));
        }
        nv = run;
    }
    uint32_t nmax = nv;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const uint32_t o = __shfl_xor(nmax, off); nmax = o > nmax ? o : nmax; }
    nmax = (uint32_t)__builtin_amdgcn_readfirstlane((int)nmax);
    float yv[DIM];
    load_row_fresh<DIM>(c.y, v, yv);
    const float s_v = c.emb_scale[v];
    const float inv_s2 = rcp(s_v * s_v);
    const uint32_t node_base = pcg_hash(pcg_hash((uint32_t)c.seed ^ a.round_key) + v);
    AE_TICK(0)
    // ---- stage B.  prepare(t0): node sets of samples t0 .. t0+S-1 and their 6 S gathers;  replay(): the
    // dependent updates.  The gathers of chunk i+1 are in flight while chunk i is replayed.
    struct Chunk {
        float rows[S][6][DIM];
        float ws[S];
        uint32_t act;
    };
    auto prepare = [&](uint32_t t0, Chunk& ck) {
        uint32_t idx[S][6];
        uint32_t need = 0;  // bit s * 8 + g: draw (s, g) still has to be (re)drawn
        ck.act = 0;
#pragma unroll
        for (int s = 0; s < S; s++) {
            const uint32_t t = t0 + (uint32_t)s;  // wave-uniform
            const bool act = t < nv;
            // sampled edge of sample t: the first m with cum[m] > t, i.e. KMAX - #{m : cum[m] > t}; bytes < 128,
            // so byte + (127 - t) carries into bit 7 exactly when cum[m] > t
            const uint32_t bias = (127u - (t < 127u ? t : 127u)) * 0x01010101u;
            uint32_t above = 0;
#pragma unroll
            for (int q = 0; q < KP; q++) above += (uint32_t)__builtin_popcount((cumP[q] + bias) & 0x80808080u);
            uint32_t m_s = (uint32_t)KMAX - above;
            m_s = m_s < (uint32_t)KMAX ? m_s : (uint32_t)KMAX - 1u;
            idx[s][0] = s_nbr[m_s * LS + lane];
            ck.ws[s] = s_w[m_s * LS + lane];
            if (act) { ck.act |= 1u << s; need |= 0x3Eu << (8 * s); }
        }
#pragma nounroll
        for (uint32_t attempt = 0; attempt < 16u; attempt++) {  // embedder.rs:1241-1253; one pass unless a draw is rejected
#pragma unroll
            for (int s = 0; s < S; s++) {
                const uint32_t j = idx[s][0];
#pragma unroll
                for (int g = 1; g <= 5; g++) {
                    const uint32_t w0 = pcg_hash(node_base + (t0 + (uint32_t)s) * 128u + (uint32_t)g * 16u + attempt);
                    uint32_t cand;
                    if constexpr (HUB) {  // NodeSampler::sample, embedder.rs:927-930
                        const uint32_t x = __umulhi(w0, (uint32_t)c.n);
                        const float uu = (float)(pcg_hash(w0 ^ 0x9E3779B9u) >> 8) * (1.0f / 16777216.0f);
                        cand = (uu < c.hub_odds[x]) ? x : c.hub_alias[x];
                    } else {
                        cand = __umulhi(w0, (uint32_t)c.n);  // :1121
                    }
                    // reject k in {i, j} or k in N(i) (NodeParam::get_edge, nodeparam.rs:83-85): min over xors is 0
                    uint32_t acc = (cand ^ v) < (cand ^ j) ? (cand ^ v) : (cand ^ j);
#pragma unroll
                    for (int m = 0; m < KMAX; m++) { const uint32_t x = nbr_reg[m] ^ cand; acc = x < acc ? x : acc; }
                    const uint32_t bit = 1u << (8 * s + g);
                    const bool mine = (need & bit) != 0u;
                    idx[s][g] = (mine || attempt == 0u) ? cand : idx[s][g];
                    need = (mine && acc != 0u) ? (need & ~bit) : need;
                }
            }
            if (!__any(need != 0u)) break;
        }
#pragma unroll
        for (int s = 0; s < S; s++) {
            const bool act = (ck.act >> s) & 1u;
#pragma unroll
            for (int g = 0; g < 6; g++) load_row_fresh<DIM>(c.y, act ? idx[s][g] : v, ck.rows[s][g]);
        }
    };
    auto replay = [&](const Chunk& ck) {
#pragma unroll
        for (int s = 0; s < S; s++) {
            const bool act = (ck.act >> s) & 1u;
            float grad[DIM];
            {   // attraction, the y_i half of embedder.rs:1207-1237
                float d = 0.f;
#pragma unroll
                for (int q = 0; q < DIM; q++) { const float df = yv[q] - ck.rows[s][0][q]; d += df * df; }
                float cij = attract_coeff<DIM, B1>(d, ck.ws[s], inv_s2, step2, a.step, a.b);
                cij = (act && d > 0.f) ? cij : 0.f;
#pragma unroll
                for (int q = 0; q < DIM; q++) { grad[q] = (ck.rows[s][0][q] - yv[q]) * cij; yv[q] -= grad[q]; }
            }
#pragma unroll
            for (int g = 1; g <= 5; g++) {  // 5 repulsions, :1267-1297
                float dk = 0.f;
#pragma unroll
                for (int q = 0; q < DIM; q++) { const float df = yv[q] - ck.rows[s][g][q]; dk += df * df; }
                const float cik = repulse_coeff<DIM, B1>(dk, inv_s2, step2, a.step, a.b);
                const bool upd = dk > 0.f;  // else `gradient` keeps its previous value (reference quirk B4)
#pragma unroll
                for (int q = 0; q < DIM; q++) {
                    const float gn = (ck.rows[s][g][q] - yv[q]) * cik;
                    grad[q] = upd ? gn : grad[q];
                    yv[q] -= act ? grad[q] : 0.f;
                }
            }
        }
    };
    {
        Chunk cA, cB;
#pragma nounroll
        for (uint32_t t0 = 0; t0 < nmax + S; t0 += S) {  // iteration i prepares chunk i and replays chunk i - 1
            if (t0 < nmax) prepare(t0, cB);
            AE_TICK(1)
            if (t0 > 0) {
                replay(cA);
                // write-through after every chunk: the other waves run this round concurrently and gather rows chunk
                // by chunk, so they see this node move during the round as the reference's threads do
                if (a.store_mode != 2 && valid && cA.act) store_row_through<DIM>(c.y, v, yv);
            }
            cA = cB;
            AE_TICK(2)
        }
    }
    if (a.store_mode == 2 && valid && nv) store_row_through<DIM>(c.y, v, yv);
    // ---- stage C: the y_j halves of :1238-1239, replayed by the target.  Per pass of CH in-edges: counts,
    // gathers of the sources' rows, an exclusive scan of the counts = position of every push in the list of
    // pushes of the pass (zero counts vanish, a count of c takes c slots, a node's pushes are contiguous since
    // the records are sorted by target); the list is parked in LDS in windows of EC pushes.
    bool any_push = false;
    if (t_begin < t_end) {
        uint32_t cn[NQ];
        float yu[NQ][DIM];
        auto count_and_gather = [&](uint64_t cb) {
            float mu[NQ], u[NQ];
#pragma unroll
            for (int q = 0; q < NQ; q++) {
                const bool in = cb + (uint64_t)(lane * NQ + q) < t_end;
                mu[q] = in ? a.unit * recA[q].w : 0.f;
                u[q] = edge_uniform(recA[q].eid, rk);
            }
            poisson_batch<NQ>(u, mu, cn);
#pragma unroll
            for (int q = 0; q < NQ; q++) load_row_fresh<DIM>(c.y, cn[q] ? recA[q].src : v, yu[q]);
        };
        if (t_begin + CH < t_end) load_recs(t_begin + CH, recB);
        count_and_gather(t_begin);
#pragma nounroll
        for (uint64_t cb = t_begin; cb < t_end; cb += CH) {
            // exclusive scan over the pass (lane-major record order)
            uint32_t mine_tot = 0;
#pragma unroll
            for (int q = 0; q < NQ; q++) mine_tot += cn[q];
            uint32_t incl = mine_tot;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) { const uint32_t o = __shfl_up(incl, off); incl += lane >= off ? o : 0u; }
            const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            uint32_t pos[NQ];
            {
                uint32_t run = incl - mine_tot;
#pragma unroll
                for (int q = 0; q < NQ; q++) { pos[q] = run; s_pos[lane * NQ + q] = run; run += cn[q]; }
            }
            if (lane == 0) s_pos[CH] = total;
            float pa[NQ], pb[NQ], pis2[NQ];
#pragma unroll
            for (int q = 0; q < NQ; q++) {
                pis2[q] = rcp(recA[q].s_src * recA[q].s_src);
                pa[q] = B1 ? step2 * pis2[q] * (1.f - recA[q].w) : recA[q].w;
                pb[q] = B1 ? step2 * pis2[q] * recA[q].w : 0.f;
            }
            const uint64_t lo = tb_v > cb ? tb_v : cb;
            const uint64_t hi = te_v < cb + CH ? te_v : cb + CH;
            const bool has_range = hi > lo;
            const bool more = cb + CH < t_end;
#pragma nounroll
            for (uint32_t w0 = 0; w0 < total; w0 += EC) {  // one window unless the pass holds more than EC pushes
#pragma unroll
                for (int q = 0; q < NQ; q++) {
                    for (uint32_t r = 0; r < cn[q]; r++) {
                        const uint32_t e = pos[q] + r - w0;  // wraps below the window
                        if (e < (uint32_t)EC) {
#pragma unroll
                            for (int t = 0; t < DIM; t++) s_in_row[e * DIM + t] = yu[q][t];
                            s_in_a[e] = pa[q];
                            s_in_b[e] = pb[q];
                            s_in_is2[e] = pis2[q];
                        }
                    }
                }
                __syncthreads();
                if (more && w0 + EC >= total) {  // last window: the next pass's counts and rows are in flight during the replay
#pragma unroll
                    for (int q = 0; q < NQ; q++) recA[q] = recB[q];
                    count_and_gather(cb + CH);
                    if (cb + 2 * CH < t_end) load_recs(cb + 2 * CH, recB);
                }
                AE_TICK(3)
                uint32_t pbeg = 0, pend = 0;
                if (has_range) { pbeg = s_pos[(uint32_t)(lo - cb)]; pend = s_pos[(uint32_t)(hi - cb)]; }
                pbeg = pbeg > w0 ? pbeg : w0;
                pend = pend < w0 + EC ? pend : w0 + EC;
                const uint32_t len = pend > pbeg ? pend - pbeg : 0u;
                uint32_t lmax = len;
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) { const uint32_t o = __shfl_xor(lmax, off); lmax = o > lmax ? o : lmax; }
                lmax = (uint32_t)__builtin_amdgcn_readfirstlane((int)lmax);
                const uint32_t e0 = len ? pbeg - w0 : 0u;
                constexpr int U = 4;
#pragma nounroll
                for (uint32_t i0 = 0; i0 < lmax; i0 += U) {
                    float ru[U][DIM], au[U], bu[U], su[U];
                    bool on[U];
#pragma unroll
                    for (int q = 0; q < U; q++) {
                        on[q] = i0 + (uint32_t)q < len;
                        const uint32_t e = on[q] ? e0 + i0 + (uint32_t)q : 0u;
#pragma unroll
                        for (int t = 0; t < DIM; t++) ru[q][t] = s_in_row[e * DIM + t];
                        au[q] = s_in_a[e];
                        bu[q] = s_in_b[e];
                        su[q] = s_in_is2[e];
                    }
#pragma unroll
                    for (int q = 0; q < U; q++) {
                        float d = 0.f;
#pragma unroll
                        for (int t = 0; t < DIM; t++) { const float df = yv[t] - ru[q][t]; d += df * df; }
                        float cij;
                        if constexpr (B1) {
                            const float delta = d * su[q];
                            const float M = fmaxf(delta * delta, 1.0f / kProbaMin);
                            cij = fmaxf((au[q] - bu[q] * M) * rcp((1.f + delta) * M), -0.49f);
                        } else {
                            cij = attract_coeff<DIM, false>(d, au[q], su[q], step2, a.step, a.b);
                        }
                        cij = (on[q] && d > 0.f) ? cij : 0.f;
#pragma unroll
                        for (int t = 0; t < DIM; t++) yv[t] += (yv[t] - ru[q][t]) * cij;
                    }
                }
                any_push |= len != 0u;
                if (a.store_mode != 2 && valid && len) store_row_through<DIM>(c.y, v, yv);
                __syncthreads();
                AE_TICK(4)
            }
            if (more && total == 0u) {  // no window ran: advance the pipeline here
#pragma unroll
                for (int q = 0; q < NQ; q++) recA[q] = recB[q];
                count_and_gather(cb + CH);
                if (cb + 2 * CH < t_end) load_recs(cb + 2 * CH, recB);
            }
        }
    }
    if (a.store_mode == 2 && any_push && valid) store_row_through<DIM>(c.y, v, yv);
    // samples drawn: one atomic per wave, spread over 1024 counters (a single address serialises at ~12 ns each)
    unsigned long long mine = valid ? (unsigned long long)nv : 0ull;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mine += __shfl_xor(mine, off);
    if (lane == 0 && mine) atomicAdd(&a.sample_counter[blockIdx.x & 1023u], mine);
    if (a.prof && lane == 0) {
        const unsigned long long tend = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < 5; i++) atomicAdd(&a.prof[i], tk_acc[i]);
        atomicAdd(&a.prof[5], tend - tk_begin);
        atomicAdd(&a.prof[6], 1ull);
    }
#undef AE_TICK
}

__global__ void in_edge_keys_kernel(uint64_t n, const uint64_t* __restrict__ indptr, const uint32_t* __restrict__ nbr,
                                    uint64_t* __restrict__ keys, uint32_t* __restrict__ payload) {
    uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    for (uint64_t e = indptr[i]; e < indptr[i + 1]; e++) {
        keys[e] = ((uint64_t)nbr[e] << 32) | i;  // (target, source)
        payload[e] = (uint32_t)e;
    }
}
__global__ void in_edge_fill_kernel(uint64_t nnz, const uint64_t* __restrict__ keys, const uint32_t* __restrict__ perm,
                                    const float* __restrict__ proba, const float* __restrict__ emb_scale, InEdge* __restrict__ tin) {
    uint64_t x = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (x >= nnz) return;
    InEdge r;
    r.src = (uint32_t)(keys[x] & 0xFFFFFFFFull);
    r.eid = perm[x];
    r.w = proba[r.eid];
    r.s_src = emb_scale[r.src];
    tin[x] = r;
}

template <int DIM>
void launch_round_fused(ae_entropy_optim* o, const NodeArgs& a, uint64_t nodes) {
    if constexpr (DIM > 0) {
        const unsigned grid = blocks_for(nodes, kBlock / kGroup);
        const bool hub = a.c.hub_odds != nullptr, b1 = a.b == 1.0f;
        if (b1 && !hub) hipLaunchKernelGGL((ce_round_group_kernel<DIM, true, false>), dim3(grid), dim3(kBlock), 0, stream(), a);
        else if (b1 && hub) hipLaunchKernelGGL((ce_round_group_kernel<DIM, true, true>), dim3(grid), dim3(kBlock), 0, stream(), a);
        else if (!b1 && !hub) hipLaunchKernelGGL((ce_round_group_kernel<DIM, false, false>), dim3(grid), dim3(kBlock), 0, stream(), a);
        else hipLaunchKernelGGL((ce_round_group_kernel<DIM, false, true>), dim3(grid), dim3(kBlock), 0, stream(), a);
    }
}

template <int DIM, int KMAX>
void launch_round_node_k(const NodeArgs& a, uint64_t nodes) {
    const unsigned grid = blocks_for(nodes, 64);
    const bool hub = a.c.hub_odds != nullptr, b1 = a.b == 1.0f;
    if (b1 && !hub) hipLaunchKernelGGL((ce_round_node_kernel<DIM, true, false, KMAX>), dim3(grid), dim3(64), 0, stream(), a);
    else if (b1 && hub) hipLaunchKernelGGL((ce_round_node_kernel<DIM, true, true, KMAX>), dim3(grid), dim3(64), 0, stream(), a);
    else if (!b1 && !hub) hipLaunchKernelGGL((ce_round_node_kernel<DIM, false, false, KMAX>), dim3(grid), dim3(64), 0, stream(), a);
    else hipLaunchKernelGGL((ce_round_node_kernel<DIM, false, true, KMAX>), dim3(grid), dim3(64), 0, stream(), a);
}
template <int DIM>
void launch_round_node(ae_entropy_optim* o, const NodeArgs& a, uint64_t nodes) {
    if constexpr (DIM > 0) {
        if (o->g->max_nbng <= 8) launch_round_node_k<DIM, 8>(a, nodes);
        else if (o->g->max_nbng <= 12) launch_round_node_k<DIM, 12>(a, nodes);
        else launch_round_node_k<DIM, 16>(a, nodes);
    }
}

template <int DIM>
void launch_apply_group(ae_entropy_optim* o, const NodeArgs& a, uint64_t nodes) {
    if constexpr (DIM > 0) {
        const unsigned grid = blocks_for(((nodes + 7) / 8) * 64, kBlock);
        if (a.b == 1.0f) hipLaunchKernelGGL((ce_apply_group_kernel<DIM, true>), dim3(grid), dim3(kBlock), 0, stream(), a);
        else hipLaunchKernelGGL((ce_apply_group_kernel<DIM, false>), dim3(grid), dim3(kBlock), 0, stream(), a);
    }
}

template <int DIM>
void launch_apply(ae_entropy_optim* o, const NodeArgs& a, uint64_t nodes) {
    if constexpr (DIM > 0) {
        if (a.b == 1.0f) hipLaunchKernelGGL((ce_apply_node_kernel<DIM, true>), dim3(blocks_for(nodes, kApplyBlock)), dim3(kApplyBlock), 0, stream(), a);
        else hipLaunchKernelGGL((ce_apply_node_kernel<DIM, false>), dim3(blocks_for(nodes, kApplyBlock)), dim3(kApplyBlock), 0, stream(), a);
    }
}

}  // namespace

namespace ae {

bool ce_node_supports_dim(uint32_t dim) { return dim == 2 || dim == 3 || dim == 4 || dim == 8 || dim == 16; }

// transposed graph (in-edges with w and the source's embedded scale), once per EntropyOptim
void ce_node_build_transpose(ae_entropy_optim* o) {
    const ae_kgraph* g = o->g;
    if (g->nnz >= 0xFFFFFFFFull) fail(AE_ERR_INVALID_ARG, "graph too large for u32 edge ids");
    DevBuf<uint64_t> k0(g->nnz), k1(g->nnz);
    DevBuf<uint32_t> p0(g->nnz), p1(g->nnz);
    hipLaunchKernelGGL(in_edge_keys_kernel, dim3(blocks_for(g->n, 256)), dim3(256), 0, stream(), g->n, g->indptr.p, g->nbr.p, k0.p, p0.p);
    check_launch("in_edge_keys");
    sort_pairs_u64_u32(k0.p, k1.p, p0.p, p1.p, g->nnz);
    o->tin.alloc(g->nnz);
    hipLaunchKernelGGL(in_edge_fill_kernel, dim3(blocks_for(g->nnz, 256)), dim3(256), 0, stream(), g->nnz, k1.p, p1.p, o->np->proba.p,
                       o->emb_scale.p, o->tin.p);
    check_launch("in_edge_fill");
    o->tptr.alloc(g->n + 1);
    rowptr_from_sorted_keys(k1.p, g->nnz, g->n, o->tptr.p);
    o->sample_counter.alloc(1024);
    o->sample_counter.zero();
    sync();
}

void ce_node_gradient_iteration(ae_entropy_optim* o, uint64_t nb_sample, double grad_step, uint32_t iter) {
    const uint64_t nodes = o->dev.node_hi - o->dev.node_lo;
    const double per_node = (double)nb_sample / (double)nodes;  // expected samples per source node in the batch
    // rounds: keep the largest per-edge Poisson mean (p_e <= 1) below 30 so that exp(-mu) stays normal in f32
    // samples per node and round: measured trade-off between fidelity to the sequential reference (final cross
    // entropy within ~10-15 %, edge-length quantiles within ~7 %) and per-round fixed costs.  The
    // node-per-lane kernel runs all its waves concurrently, so within a round every gather sees the previous
    // round's rows (Jacobi-like) and needs shorter rounds (8) than the lane-group kernel (12), whose waves
    // finish at different times, for the same fidelity.
    const bool node_kernel = o->g->max_nbng <= 16 && !getenv("AE_CE_UNFUSED") && !getenv("AE_CE_GROUP");
    double per_round_target = node_kernel ? 8.0 : 12.0;
    if (getenv("AE_CE_PER_ROUND")) per_round_target = atof(getenv("AE_CE_PER_ROUND"));
    const uint32_t rounds = (uint32_t)std::max(1.0, std::ceil(per_node / per_round_target));
    o->rounds = rounds;
    if (iter >= (1u << 20) || rounds >= (1u << 10)) fail(AE_ERR_INVALID_ARG, "iteration / round index too large for the RNG key");
    const double per_round = per_node / (double)rounds;
    // plan capacity per node and round: Poisson(per_round) exceeds mean + 8 sigma + 8 with probability < 1e-14
    const uint32_t cap = (uint32_t)std::ceil(per_round + 8.0 * std::sqrt(per_round) + 8.0);
    if (o->plan.n < nodes * (uint64_t)cap * 8) o->plan.alloc(nodes * (uint64_t)cap * 8);
    if (o->tot.n < nodes) o->tot.alloc(nodes);
    if (o->cnt.n < o->dev.nnz) { o->cnt.alloc(o->dev.nnz); o->cnt.zero(); }
    NodeArgs a;
    a.c = o->dev;
    a.tptr = o->tptr.p;
    a.tin = o->tin.p;
    a.cnt = o->cnt.p;
    a.tot = o->tot.p;
    a.plan = o->plan.p;
    a.cap = cap;
    a.step = (float)grad_step;
    a.unit = (float)per_round;
    a.b = (float)o->dev.b;
    a.sample_counter = o->sample_counter.p;
    a.overflow = o->err.p;
    static DevBuf<unsigned long long> prof_buf;
    a.prof = nullptr;
    if (getenv("AE_CE_PROF")) {
        if (!prof_buf.n) { prof_buf.alloc(8); prof_buf.zero(); }
        a.prof = prof_buf.p;
    }
    const bool sharded = o->dev.shard_edges != o->dev.nnz;
    const bool fused = o->g->max_nbng <= 16 && !getenv("AE_CE_UNFUSED");
    const bool group_kernel = getenv("AE_CE_GROUP") != nullptr;  // the 8-lanes-per-node fused kernel (kept for A/B)
    a.skip = getenv("AE_CE_SKIP") ? atoi(getenv("AE_CE_SKIP")) : 0;
    // 2: write-through at phase ends (default).  0: after every chunk -- measured slower in the node kernel (loads
    // and stores share vmcnt and may return out of order, so every wait after a store drains it: +50 % time)
    // for a fidelity gain that shorter rounds give more cheaply
    a.store_mode = getenv("AE_CE_STORE") ? atoi(getenv("AE_CE_STORE")) : 2;
    for (uint32_t r = 0; r < rounds; r++) {
        a.round_key = (iter << 10) | r;
        if (fused && !group_kernel) { AE_DISPATCH_DIM(o->dev.dim, launch_round_node, o, a, nodes); continue; }
        if (fused) { AE_DISPATCH_DIM(o->dev.dim, launch_round_fused, o, a, nodes); continue; }
        const unsigned plan_grid = blocks_for(nodes * 64, kBlock);
        if (a.c.hub_odds) hipLaunchKernelGGL((ce_plan_node_kernel<true>), dim3(plan_grid), dim3(kBlock), 0, stream(), a);
        else hipLaunchKernelGGL((ce_plan_node_kernel<false>), dim3(plan_grid), dim3(kBlock), 0, stream(), a);
        if (sharded) hipLaunchKernelGGL(ce_count_remote_kernel, dim3(grid_cap(o->dev.nnz, kBlock)), dim3(kBlock), 0, stream(), a);
        hipLaunchKernelGGL(ce_sum_tot_kernel, dim3(1), dim3(1024), 0, stream(), (const uint32_t*)o->tot.p, nodes, o->sample_counter.p);
        if (getenv("AE_CE_THREAD_PER_NODE")) { AE_DISPATCH_DIM(o->dev.dim, launch_apply, o, a, nodes); }
        else { AE_DISPATCH_DIM(o->dev.dim, launch_apply_group, o, a, nodes); }
    }
    check_launch("ce_node");
    if (a.prof) {
        unsigned long long h[8];
        prof_buf.download(h, 8);
        if (h[6]) fprintf(stderr, "CEPROF waves=%llu per-wave cycles: [0] %.0f [1] %.0f [2] %.0f [3] %.0f [4] %.0f total %.0f\n", h[6],
                          (double)h[0] / h[6], (double)h[1] / h[6], (double)h[2] / h[6], (double)h[3] / h[6], (double)h[4] / h[6], (double)h[5] / h[6]);
        prof_buf.zero();
    }
}

}  // namespace ae
