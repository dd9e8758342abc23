// ce_node_round.h -- the node-per-lane CE round kernel (template) and its launcher; instantiated for the exact
// dimensions in ce_node_round_exact.hip and for the zero-padded ones in ce_node_round_pad.hip (two translation
// units: the kernel is large and hipcc compiles them in parallel).
#pragma once
#include <type_traits>
#include "ce_node_common.h"

namespace ae {

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
    static constexpr int S = DIM <= 4 ? 2 : 1;  // samples whose rows are gathered together (measured on C2: S = 2 0.73 ms, 3 0.77, 4 0.78, 1 0.79)
    static constexpr int NQ = DIM <= 4 ? 8 : 4;                   // in-edge records per lane and pass of stage C
    static constexpr int CH = 64 * NQ;
    static constexpr int EC = DIM <= 4 ? 1024 : (DIM <= 8 ? 256 : 128);  // pushes parked in LDS per window
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

// Wide rows (d = 8 / 16: 32 / 64 bytes).  With the row held by one lane, a gather instruction puts every lane on a cache
// line of its own and a row costs 2 / 4 requests; out of a footprint beyond the Infinity Cache the memory system then
// delivers 40 / 24 G rows/s, against 55 G rows/s when each row is ONE request (tools/ubench_rowgather.hip: the bound is
// requests, not bytes).  So the rows of a lane group (G = d/4 lanes) are fetched cooperatively -- in step j every lane of
// the group loads its 16-byte quarter of the row wanted by the group's lane j -- and handed to their owners through LDS
// when they are consumed (`untangle`).
template <int CTRL>
__device__ __forceinline__ uint32_t quad_perm(uint32_t x) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, CTRL, 0xF, 0xF, true);
}
template <int G>
__device__ __forceinline__ uint32_t group_bcast(uint32_t x, int j) {  // value of the group's lane j (j is a constant after unrolling)
    if constexpr (G == 4) {
        switch (j) {
            case 0: return quad_perm<0x00>(x);
            case 1: return quad_perm<0x55>(x);
            case 2: return quad_perm<0xAA>(x);
            default: return quad_perm<0xFF>(x);
        }
    } else {
        return j == 0 ? quad_perm<0xA0>(x) : quad_perm<0xF5>(x);
    }
}

// The workgroup is ONE wave: its LDS operations execute in program order, so making one lane's LDS writes visible to
// the others needs no s_barrier -- and must not use __syncthreads(), whose workgroup-scope fence drains every
// outstanding global load (vmcnt(0)) and would serialise the gathers that are deliberately left in flight.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// TILE: the negatives of the uniform sampler come from an LDS tile (below); a kernel of its own, not a run-time branch -- the
// registers of a kernel are those of its hungriest path, and the tile path does without the 5 S gathered rows per chunk
template <int DIM, bool PAD, bool B1, int KMAX, bool TILE>
__global__ void __launch_bounds__(64, (DIM == 8 ? 2 : 1)) ce_round_node_kernel(NodeArgs a) {  // d = 8: two waves per SIMD (gathered negatives: 10 registers spilled, the kernel otherwise lands on 267 registers, i.e. one wave, and the C4 shape is latency-bound; tile negatives: 236 registers, nothing spilled).  d = 16 with tile negatives needs 338 registers: one wave (two with 83 spilled: 4 % slower)
    using Cfg = NodeKernelCfg<DIM>;
    constexpr int LS = 65, S = Cfg::S, CH = Cfg::CH, NQ = Cfg::NQ, EC = Cfg::EC, KP = KMAX / 4;
    __shared__ uint32_t s_nbr[KMAX * LS];
    __shared__ float s_w[KMAX * LS];
    __shared__ float s_in_row[EC * DIM];
    __shared__ float s_in_a[EC];      // b == 1: 2 step / s_u^2 * (1 - w); else w
    __shared__ float s_in_b[EC];      // b == 1: 2 step / s_u^2 * w
    __shared__ float s_in_is2[EC];
    __shared__ uint32_t s_pos[CH + 1];
    constexpr bool COOP = !PAD && (DIM == 8 || DIM == 16);  // rows gathered by lane groups (see group_bcast above)
    constexpr int G = COOP ? DIM / 4 : 1, RS = DIM + 4;     // LDS row stride of the hand-over buffer: 16-byte aligned, spreads the banks
    using f4 = __attribute__((ext_vector_type(4))) float;
    __shared__ __attribute__((aligned(16))) float s_tr[COOP ? 64 * RS : 4];
    const CeDev c = a.c;
    const int lane = threadIdx.x;
    // row access: exact dimension = vector loads of the whole row; PAD = asked_dim < DIM, the registers beyond
    // asked_dim stay 0 (they add nothing to a distance and never move)
    auto ld = [&](uint32_t node, float* out) {
        if constexpr (!PAD) load_row_fresh<DIM>(c.y, node, out);
        else {
            const float* p = c.y + (uint64_t)node * c.dim;
#pragma unroll
            for (int t = 0; t < DIM; t++) out[t] = (uint32_t)t < c.dim ? __builtin_nontemporal_load(p + t) : 0.f;
        }
    };
    auto untangle = [&](float* r) {
        if constexpr (COOP) {
            const uint32_t sub = (uint32_t)lane & (uint32_t)(G - 1), base = (uint32_t)lane & ~(uint32_t)(G - 1);
#pragma unroll
            for (int j = 0; j < G; j++) {
                f4 t; t.x = r[4 * j]; t.y = r[4 * j + 1]; t.z = r[4 * j + 2]; t.w = r[4 * j + 3];
                *reinterpret_cast<f4*>(&s_tr[(base + (uint32_t)j) * RS + sub * 4u]) = t;
            }
            wave_lds_sync();
#pragma unroll
            for (int q = 0; q < G; q++) {
                const f4 t = *reinterpret_cast<const f4*>(&s_tr[(uint32_t)lane * RS + (uint32_t)q * 4u]);
                r[4 * q] = t.x; r[4 * q + 1] = t.y; r[4 * q + 2] = t.z; r[4 * q + 3] = t.w;
            }
            wave_lds_sync();
        }
    };
    auto st = [&](uint32_t node, const float* in) {
        if constexpr (!PAD) store_row_through<DIM>(c.y, node, in);
        else {
            float* p = c.y + (uint64_t)node * c.dim;
#pragma unroll
            for (int t = 0; t < DIM; t++)
                if ((uint32_t)t < c.dim) __hip_atomic_store(reinterpret_cast<uint32_t*>(p) + t, __float_as_uint(in[t]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    const bool hub = c.hub_odds != nullptr;
    const uint64_t nodes_owned = c.node_hi - c.node_lo;
    const uint64_t local0 = blockIdx.x * 64ull;
    const uint64_t local = local0 + (uint64_t)lane;
    const bool valid = local < nodes_owned;
    const uint32_t v = (uint32_t)(c.node_lo + (valid ? local : nodes_owned - 1));
    // gather of a partner row.  COOP: `out` receives the group's rows in quarters (out[4 j ..] = this lane's quarter of the
    // row wanted by the group's lane j) until untangle() swaps them into place; every lane takes part (no divergence here)
    auto ldg = [&](uint32_t node, bool want, float* out) {
        if constexpr (!COOP) ld(want ? node : v, out);
        else {
            // a row nobody wants (idle lane, in-edge without a push) is not requested at all: `out` keeps its previous
            // (finite) content, which the masked arithmetic ignores.  The predicate is uniform over the lane group.
            const uint32_t sub = (uint32_t)lane & (uint32_t)(G - 1);
#pragma unroll
            for (int j = 0; j < G; j++) {
                const uint32_t rj = group_bcast<G>(node, j);
                const uint32_t wj = group_bcast<G>(want ? 1u : 0u, j);
                if (wj) {
                    const f4 t = __builtin_nontemporal_load(reinterpret_cast<const f4*>(c.y + (uint64_t)rj * DIM) + sub);
                    out[4 * j] = t.x; out[4 * j + 1] = t.y; out[4 * j + 2] = t.z; out[4 * j + 3] = t.w;
                }
            }
        }
    };
    uint64_t ib;
    uint32_t k;
    if (c.uniform_k) { ib = (uint64_t)v * c.uniform_k; k = c.uniform_k; }
    else { ib = c.indptr[v]; k = (uint32_t)(c.indptr[v + 1] - ib); }
    const uint32_t rk = round_hash_key(a.round_key, c.seed);
    const float step2 = 2.0f * a.step;
    // ---- tile negatives (uniform sampler only): the 5 negatives of every sample of this wave are drawn from T consecutive
    // rows starting at a random node (per wave and round; the start is uniform over the nodes and the tile wraps around, so
    // every node is equally likely -- the marginal law of embedder.rs:1121), read ONCE, coalesced, into the LDS buffer that
    // stage C uses later.  A negative then costs an LDS read instead of a 128-byte line fetched for a 4 DIM-byte row: at the
    // C4 shape 5 of the 6 partner rows of a sample, ~70 % of the kernel's memory traffic.
    // Hubness-weighted sampler (NodeSampler, embedder.rs:915-930): a run of consecutive rows cannot carry the weights, so the
    // tile is T rows DRAWN from the alias table (every slot an independent draw of the reference's law: picking a slot
    // uniformly afterwards is again a draw of that law), their ids kept beside them for the rejection test -- 3 T random
    // accesses per wave and round (two table entries and a row per slot) instead of 3 per negative (C4 shape, ~20 samples
    // per lane and batch ... 360 -> 70 ms per batch).
    constexpr int T = EC, TL = T * DIM / 64, TPL = T / 64;  // TPL: slots drawn per lane in the hubness variant
    constexpr bool tile_on = TILE;  // (the launcher checks n >= 2 T)
    const uint32_t tbase = __umulhi(pcg_hash(rk ^ pcg_hash((uint32_t)blockIdx.x + 0x51ED270Bu)), (uint32_t)c.n);
    uint32_t* s_tile_id = reinterpret_cast<uint32_t*>(s_in_a);  // hubness variant: node id of every slot (stage C reuses the buffer later)
    float tile_raw[TL];
    auto tile_loads = [&] {
        if (hub) return;  // (drawn and loaded in tile_store: nothing is held in registers across stage A)
        if constexpr (!PAD && DIM % 4 == 0) {
#pragma unroll
            for (int i = 0; i < TL / 4; i++) {
                const uint32_t e = (uint32_t)(i * 64 + lane) * 4u, row = e / (uint32_t)DIM, col = e % (uint32_t)DIM;
                uint32_t node = tbase + row;
                node -= node >= (uint32_t)c.n ? (uint32_t)c.n : 0u;
                const f4 t = __builtin_nontemporal_load(reinterpret_cast<const f4*>(c.y + (uint64_t)node * DIM + col));
                tile_raw[4 * i] = t.x; tile_raw[4 * i + 1] = t.y; tile_raw[4 * i + 2] = t.z; tile_raw[4 * i + 3] = t.w;
            }
        } else if constexpr (!PAD && DIM == 2) {
            using f2 = __attribute__((ext_vector_type(2))) float;
#pragma unroll
            for (int i = 0; i < TL / 2; i++) {
                uint32_t node = tbase + (uint32_t)(i * 64 + lane);
                node -= node >= (uint32_t)c.n ? (uint32_t)c.n : 0u;
                const f2 t = __builtin_nontemporal_load(reinterpret_cast<const f2*>(c.y + (uint64_t)node * 2u));
                tile_raw[2 * i] = t.x; tile_raw[2 * i + 1] = t.y;
            }
        } else {  // odd or padded rows: element-wise, consecutive lanes on consecutive floats
            const uint32_t cd = PAD ? c.dim : (uint32_t)DIM;
#pragma unroll
            for (int i = 0; i < TL; i++) {
                const uint32_t e = (uint32_t)(i * 64 + lane), row = e / cd, col = e % cd;
                uint32_t node = tbase + (row < (uint32_t)T ? row : 0u);
                node -= node >= (uint32_t)c.n ? (uint32_t)c.n : 0u;
                tile_raw[i] = __builtin_nontemporal_load(c.y + (uint64_t)node * cd + col);
            }
        }
    };
    auto tile_store = [&] {
        if (hub) {
            uint32_t node[TPL], hx[TPL], hal[TPL];
            float hod[TPL], hu[TPL];
#pragma unroll
            for (int i = 0; i < TPL; i++) {  // the alias look-ups of this lane's slots, all in flight together
                const uint32_t w0 = pcg_hash(rk ^ pcg_hash((uint32_t)blockIdx.x * (uint32_t)T + (uint32_t)(i * 64 + lane) + 0x51ED270Bu));
                hx[i] = __umulhi(w0, (uint32_t)c.n);
                hu[i] = (float)(pcg_hash(w0 ^ 0x9E3779B9u) >> 8) * (1.0f / 16777216.0f);
                const uint2 he = c.hub_tab[hx[i]];
                hod[i] = __uint_as_float(he.x);
                hal[i] = he.y;
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < TPL; i++) {
                node[i] = (hu[i] < hod[i]) ? hx[i] : hal[i];  // NodeSampler::sample, embedder.rs:927-930
                ld(node[i], &tile_raw[i * DIM]);
            }
#pragma unroll
            for (int i = 0; i < TPL; i++) {
                const uint32_t slot = (uint32_t)(i * 64 + lane);
                s_tile_id[slot] = node[i];
#pragma unroll
                for (int t = 0; t < DIM; t++) s_in_row[slot * (uint32_t)DIM + (uint32_t)t] = tile_raw[i * DIM + t];
            }
            wave_lds_sync();
            return;
        }
        if constexpr (!PAD && DIM % 4 == 0) {
#pragma unroll
            for (int i = 0; i < TL / 4; i++) {
                f4 t; t.x = tile_raw[4 * i]; t.y = tile_raw[4 * i + 1]; t.z = tile_raw[4 * i + 2]; t.w = tile_raw[4 * i + 3];
                *reinterpret_cast<f4*>(&s_in_row[(uint32_t)(i * 64 + lane) * 4u]) = t;
            }
        } else if constexpr (!PAD && DIM == 2) {
#pragma unroll
            for (int i = 0; i < TL / 2; i++) { s_in_row[(uint32_t)(i * 64 + lane) * 2u] = tile_raw[2 * i]; s_in_row[(uint32_t)(i * 64 + lane) * 2u + 1u] = tile_raw[2 * i + 1]; }
        } else {
            const uint32_t cd = PAD ? c.dim : (uint32_t)DIM;
            if constexpr (PAD) {
#pragma unroll
                for (int i = 0; i < TL; i++) s_in_row[i * 64 + lane] = 0.f;  // the columns beyond asked_dim
                wave_lds_sync();
            }
#pragma unroll
            for (int i = 0; i < TL; i++) {
                const uint32_t e = (uint32_t)(i * 64 + lane), row = e / cd, col = e % cd;
                if (row < (uint32_t)T) s_in_row[row * (uint32_t)DIM + col] = tile_raw[i];
            }
        }
        wave_lds_sync();
    };
    unsigned long long tk0 = a.prof ? __builtin_amdgcn_s_memtime() : 0ull, tk_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long tk_begin = tk0;
#define AE_TICK(i) if (a.prof) { const unsigned long long tk1 = __builtin_amdgcn_s_memtime(); tk_acc[i] += tk1 - tk0; tk0 = tk1; }
    // ---- loads of stage A: they depend on v only.  Where they are issued is a measured choice per kernel family:
    // d >= 8 (C4 / C5 shapes, memory system saturated, a dependent round trip ~36 k cycles): ahead of everything else;
    // d <= 4: after the in-edge record prefetch, the own row after the counts (C2: 48.3 us per launch against 48.8 us)
    constexpr bool LATE_RECS = DIM >= 8;
    uint32_t nbr_reg[KMAX];
    float pr_raw[KMAX], yv[DIM], s_v;
    const uint32_t v0 = (uint32_t)(c.node_lo + local0);
    const uint64_t n_here = (nodes_owned - local0) < 64ull ? (nodes_owned - local0) : 64ull;
    f4 own_raw[G];
    auto row_loads = [&] {
#pragma unroll
        for (int m = 0; m < KMAX; m++) {  // unconditional loads (clamped index): all in flight together
            const uint32_t mm = (uint32_t)m < k ? (uint32_t)m : k - 1u;
            nbr_reg[m] = c.nbr[ib + mm];
            pr_raw[m] = c.proba[ib + mm];
        }
    };
    // the wave's own 64 rows are contiguous in memory: for wide rows they are read (and written back, st_rows) as one
    // coalesced block through the hand-over buffer -- 8 lines per instruction instead of one line per lane
    auto own_loads = [&] {
        if constexpr (COOP) {
#pragma unroll
            for (int i = 0; i < G; i++) {
                const uint32_t e = (uint32_t)(i * 64 + lane), row = e / (uint32_t)G, qq = e % (uint32_t)G;
                const uint32_t rc = row < (uint32_t)n_here ? row : (uint32_t)n_here - 1u;
                own_raw[i] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(c.y + (uint64_t)(v0 + rc) * DIM) + qq);
            }
        } else {
            ld(v, yv);
        }
        s_v = c.emb_scale[v];
    };
    if constexpr (LATE_RECS) { row_loads(); own_loads(); }
    if constexpr (tile_on) tile_loads();
    // ---- stage C prologue: the first in-edge records of the wave are requested now, their latency overlaps
    // stages A and B.  Lane l holds the NQ consecutive records cb + l NQ .. cb + l NQ + NQ - 1.
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
    // d >= 8: the kernel is at the register limit and the compiler parks the records in AGPRs as soon as they arrive, i.e.
    // it waits for them on the spot -- a second exposed round trip in front of stage A.  There the first records are
    // requested together with the gathers of the first chunk of stage B (one wait for both).
    auto late_recs = [&] {
        if constexpr (LATE_RECS) { if (t_begin < t_end) load_recs(t_begin, recA); }
    };
    if constexpr (!LATE_RECS) { if (t_begin < t_end) load_recs(t_begin, recA); }
    if constexpr (!LATE_RECS) row_loads();
    // ---- stage A: the node's row in registers (rejection test) and in an LDS column private to the lane
    // (dynamic index, no barrier needed), cumulative Poisson counts of the out-edges packed 4 per register
    uint32_t cumP[KP];
    uint32_t nv;
    {
        float pr[KMAX], mu[KMAX], u[KMAX];
        uint32_t cnt[KMAX];
#pragma unroll
        for (int m = 0; m < KMAX; m++) {
            const bool has = (uint32_t)m < k;
            nbr_reg[m] = has ? nbr_reg[m] : 0xFFFFFFFFu;  // the pad never equals a candidate
            pr[m] = has ? pr_raw[m] : 0.f;
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
    if constexpr (!LATE_RECS) own_loads();
    const float inv_s2 = rcp(s_v * s_v);
    if constexpr (COOP) {
#pragma unroll
        for (int i = 0; i < G; i++) {
            const uint32_t e = (uint32_t)(i * 64 + lane), row = e / (uint32_t)G, qq = e % (uint32_t)G;
            *reinterpret_cast<f4*>(&s_tr[row * RS + qq * 4u]) = own_raw[i];
        }
        wave_lds_sync();
#pragma unroll
        for (int q = 0; q < G; q++) {
            const f4 t = *reinterpret_cast<const f4*>(&s_tr[(uint32_t)lane * RS + (uint32_t)q * 4u]);
            yv[4 * q] = t.x; yv[4 * q + 1] = t.y; yv[4 * q + 2] = t.z; yv[4 * q + 3] = t.w;
        }
        wave_lds_sync();
    }
    // write-back of the wave's rows, all lanes (COOP only): a row has one writer, rewriting an unchanged row is harmless
    auto st_rows = [&]() {
        if constexpr (COOP) {
            constexpr int H = DIM / 2;
#pragma unroll
            for (int q = 0; q < G; q++) {
                f4 t; t.x = yv[4 * q]; t.y = yv[4 * q + 1]; t.z = yv[4 * q + 2]; t.w = yv[4 * q + 3];
                *reinterpret_cast<f4*>(&s_tr[(uint32_t)lane * RS + (uint32_t)q * 4u]) = t;
            }
            wave_lds_sync();
            uint64_t* dst = reinterpret_cast<uint64_t*>(c.y + (uint64_t)v0 * DIM);
#pragma unroll
            for (int i = 0; i < H; i++) {
                const uint32_t e = (uint32_t)(i * 64 + lane), row = e / (uint32_t)H, h = e % (uint32_t)H;
                const uint64_t bits = *reinterpret_cast<const uint64_t*>(&s_tr[row * RS + h * 2u]);
                if (row < (uint32_t)n_here) __hip_atomic_store(dst + e, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            wave_lds_sync();
        }
    };
    const uint32_t node_base = pcg_hash(pcg_hash((uint32_t)c.seed ^ a.round_key) + v);
    if constexpr (tile_on) tile_store();
    AE_TICK(0)
    // ---- stage B.  prepare(t0): node sets of samples t0 .. t0+S-1 and their 6 S gathers;  replay(): the
    // dependent updates.  The gathers of chunk i+1 are in flight while chunk i is replayed.
    struct Chunk {
        float rows[S][6][DIM];
        float ws[S];
        uint32_t act;
        uint32_t slot[S][6];  // tile negatives: LDS slot of negative g (rows[s][1..5] unused then)
        uint32_t bad;         // bit 8 s + g: no admissible negative found for draw (s, g) in 16 attempts -- that repulsion is skipped
    };
    struct ChunkPlan {  // node sets of S samples: idx[s][0] = sampled neighbour j, idx[s][1..5] = negatives
        uint32_t idx[S][6];
        float ws[S];
        uint32_t act;
        uint32_t bad;
    };
    // one attempt of all 5 S draws (embedder.rs:1241-1253), branch-free: `need` has bit 8 s + g set while draw (s, g) is
    // still wanted; reject k = i or k in N(i) (NodeParam::get_edge, nodeparam.rs:83-85; the sampled j is in N(i)):
    // min over xors is 0
    auto draw_pass = [&](auto hub_tag, uint32_t t0, uint32_t attempt, ChunkPlan& pl, uint32_t& need) {
        constexpr int MODE = decltype(hub_tag)::value;  // 0 uniform (gathered), 1 hubness alias table, 2 uniform from the LDS tile
        constexpr bool HUB = MODE == 1;
        uint32_t cands[S][6], slots[S][6] = {};
        if constexpr (MODE == 2) {
#pragma unroll
            for (int s = 0; s < S; s++)
#pragma unroll
                for (int g = 1; g <= 5; g++) {
                    slots[s][g] = __umulhi(pcg_hash(node_base + (t0 + (uint32_t)s) * 128u + (uint32_t)g * 16u + attempt), (uint32_t)T);
                    uint32_t node = tbase + slots[s][g];
                    node -= node >= (uint32_t)c.n ? (uint32_t)c.n : 0u;
                    cands[s][g] = hub ? s_tile_id[slots[s][g]] : node;
                }
        } else if constexpr (HUB) {  // NodeSampler::sample, embedder.rs:927-930: the 10 S table look-ups are issued together
            uint32_t xs[S][6], al[S][6];
            float od[S][6], uu[S][6];
#pragma unroll
            for (int s = 0; s < S; s++) {
#pragma unroll
                for (int g = 1; g <= 5; g++) {
                    const uint32_t w0 = pcg_hash(node_base + (t0 + (uint32_t)s) * 128u + (uint32_t)g * 16u + attempt);
                    xs[s][g] = __umulhi(w0, (uint32_t)c.n);
                    uu[s][g] = (float)(pcg_hash(w0 ^ 0x9E3779B9u) >> 8) * (1.0f / 16777216.0f);
                    const uint2 he = c.hub_tab[xs[s][g]];
                    od[s][g] = __uint_as_float(he.x);
                    al[s][g] = he.y;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < S; s++)
#pragma unroll
                for (int g = 1; g <= 5; g++) cands[s][g] = (uu[s][g] < od[s][g]) ? xs[s][g] : al[s][g];
        } else {
#pragma unroll
            for (int s = 0; s < S; s++)
#pragma unroll
                for (int g = 1; g <= 5; g++)
                    cands[s][g] = __umulhi(pcg_hash(node_base + (t0 + (uint32_t)s) * 128u + (uint32_t)g * 16u + attempt), (uint32_t)c.n);  // :1121
        }
#pragma unroll
        for (int s = 0; s < S; s++) {
#pragma unroll
            for (int g = 1; g <= 5; g++) {
                const uint32_t cand = cands[s][g];
                uint32_t acc = cand ^ v;
#pragma unroll
                for (int m = 0; m < KMAX; m++) { const uint32_t x = nbr_reg[m] ^ cand; acc = x < acc ? x : acc; }
                const uint32_t bit = 1u << (8 * s + g);
                const bool mine = (need & bit) != 0u;
                pl.idx[s][g] = (mine || attempt == 0u) ? (MODE == 2 ? slots[s][g] : cand) : pl.idx[s][g];
                need = (mine && acc != 0u) ? (need & ~bit) : need;
            }
        }
    };
    // plan of chunk t0; `overlap` (the replay of an earlier chunk) sits in the same basic block as the first attempt so
    // that the scheduler can fill the stalls of its dependent chain with the independent hash / compare streams
    auto plan_chunk = [&](auto hub_tag, uint32_t t0, ChunkPlan& pl, auto&& overlap) {
        uint32_t need = 0;
        pl.act = 0;
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
            pl.idx[s][0] = s_nbr[m_s * LS + lane];
            pl.ws[s] = s_w[m_s * LS + lane];
            pl.act |= act ? 1u << s : 0u;
            need |= act ? 0x3Eu << (8 * s) : 0u;
        }
        draw_pass(hub_tag, t0, 0u, pl, need);
        overlap();
#pragma nounroll
        for (uint32_t attempt = 1; attempt < 16u && __any(need != 0u); attempt++) draw_pass(hub_tag, t0, attempt, pl, need);  // rare
        pl.bad = need;  // (the reference loops until it finds one, embedder.rs:1241-1253: P[16 rejections] < ((k + 2) / n)^16)
    };
    auto issue = [&](auto mode_tag, const ChunkPlan& pl, Chunk& ck) {  // the 6 S gathers of a planned chunk (tile negatives: the S positive ones)
        constexpr int MODE = decltype(mode_tag)::value;
        ck.act = pl.act;
        ck.bad = pl.bad;
#pragma unroll
        for (int s = 0; s < S; s++) {
            const bool act = (pl.act >> s) & 1u;
            ck.ws[s] = pl.ws[s];
#pragma unroll
            for (int g = 0; g < 6; g++) {
                if constexpr (MODE == 2) {
                    if (g == 0) ldg(pl.idx[s][g], act, ck.rows[s][g]);
                    else ck.slot[s][g] = pl.idx[s][g];
                } else {
                    ldg(pl.idx[s][g], act, ck.rows[s][g]);
                }
            }
        }
    };
    auto replay = [&](auto mode_tag, Chunk& ck) {
        constexpr int MODE = decltype(mode_tag)::value;
#pragma unroll
        for (int s = 0; s < S; s++) {
            const bool act = (ck.act >> s) & 1u;
            float grad[DIM];
            {   // attraction, the y_i half of embedder.rs:1207-1237
                untangle(ck.rows[s][0]);
                float d = 0.f;
#pragma unroll
                for (int q = 0; q < DIM; q++) { const float df = yv[q] - ck.rows[s][0][q]; d += df * df; }
                float cij = attract_coeff<DIM, B1>(d, ck.ws[s], inv_s2, step2, a.step, a.b);
                cij = (act && d > 0.f) ? cij : 0.f;
#pragma unroll
                for (int q = 0; q < DIM; q++) { grad[q] = (ck.rows[s][0][q] - yv[q]) * cij; yv[q] -= grad[q]; }
            }
            float tneg[5][DIM];
            if constexpr (MODE == 2) {  // the five rows out of the tile, read together ahead of the dependent chain
#pragma unroll
                for (int g = 1; g <= 5; g++) {
                    const float* tp = &s_in_row[ck.slot[s][g] * (uint32_t)DIM];
                    if constexpr (DIM % 4 == 0) {
#pragma unroll
                        for (int q = 0; q < DIM / 4; q++) {
                            const f4 t = *reinterpret_cast<const f4*>(tp + 4 * q);
                            tneg[g - 1][4 * q] = t.x; tneg[g - 1][4 * q + 1] = t.y; tneg[g - 1][4 * q + 2] = t.z; tneg[g - 1][4 * q + 3] = t.w;
                        }
                    } else {
#pragma unroll
                        for (int q = 0; q < DIM; q++) tneg[g - 1][q] = tp[q];
                    }
                }
            }
#pragma unroll
            for (int g = 1; g <= 5; g++) {  // 5 repulsions, :1267-1297
                if constexpr (MODE != 2) untangle(ck.rows[s][g]);
                const float* rg = MODE == 2 ? tneg[g - 1] : ck.rows[s][g];
                float dk = 0.f;
#pragma unroll
                for (int q = 0; q < DIM; q++) { const float df = yv[q] - rg[q]; dk += df * df; }
                const float cik = repulse_coeff<DIM, B1>(dk, inv_s2, step2, a.step, a.b);
                const bool live = act && ((ck.bad >> (8 * s + g)) & 1u) == 0u;
                const bool upd = dk > 0.f && live;  // else `gradient` keeps its previous value (reference quirk B4)
#pragma unroll
                for (int q = 0; q < DIM; q++) {
                    const float gn = (rg[q] - yv[q]) * cik;
                    grad[q] = upd ? gn : grad[q];
                    yv[q] -= live ? grad[q] : 0.f;
                }
            }
        }
    };
    // Software pipeline over chunks: the rows of chunk i + 1 are gathered while chunk i + 2 is planned and chunk i replayed.
    auto stage_b = [&](auto hub_tag) {
        ChunkPlan pl;
        if constexpr (DIM <= 16) {
            Chunk cA = {}, cB = {};  // zeroed once: a row that is not requested keeps finite content
            if (nmax) {
                plan_chunk(hub_tag, 0u, pl, [] {});
                issue(hub_tag, pl, cA);
                late_recs();
                if ((uint32_t)S < nmax) plan_chunk(hub_tag, (uint32_t)S, pl, [] {});
            } else late_recs();
            AE_TICK(1)
#pragma nounroll
            for (uint32_t t0 = 0; t0 < nmax; t0 += S) {
                const bool has1 = t0 + S < nmax, has2 = t0 + 2 * S < nmax;
                if (has1) issue(hub_tag, pl, cB);
                if (has2) plan_chunk(hub_tag, t0 + 2 * S, pl, [&] { replay(hub_tag, cA); });
                else replay(hub_tag, cA);
                if (a.store_mode == 0 && valid && cA.act) st(v, yv);  // optional write-through after every chunk (AE_CE_STORE=0)
                if (has1) cA = cB;
                AE_TICK(2)
            }
        } else {  // 32 padded columns: one chunk in registers at a time
            Chunk cA;
            late_recs();
#pragma nounroll
            for (uint32_t t0 = 0; t0 < nmax; t0 += S) {
                plan_chunk(hub_tag, t0, pl, [] {});
                issue(hub_tag, pl, cA);
                replay(hub_tag, cA);
                if (a.store_mode == 0 && valid && cA.act) st(v, yv);
            }
        }
    };
    if constexpr (TILE) stage_b(std::integral_constant<int, 2>{});
    else {
        if (hub) stage_b(std::integral_constant<int, 1>{});
        else stage_b(std::integral_constant<int, 0>{});
    }
    if constexpr (COOP) { if (a.store_mode == 2) st_rows(); }
    else if (a.store_mode == 2 && valid && nv) st(v, yv);  // mode 3: one store at the very end only
    // ---- stage C: the y_j halves of :1238-1239, replayed by the target.  Per pass of CH in-edges: counts,
    // gathers of the sources' rows, an exclusive scan of the counts = position of every push in the list of
    // pushes of the pass (zero counts vanish, a count of c takes c slots, a node's pushes are contiguous since
    // the records are sorted by target); the list is parked in LDS in windows of EC pushes.
    bool any_push = false;
    if (t_begin < t_end) {
        uint32_t cn[NQ];
        float yu[NQ][DIM] = {};
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
            for (int q = 0; q < NQ; q++) ldg(recA[q].src, cn[q] != 0u, yu[q]);
        };
        load_recs(t_begin + CH, recB);  // (clamped inside when past the end)
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
            // everything the parking below needs, in registers of its own: the pipeline registers (recA, cn, yu) are
            // refilled for the NEXT pass right away, unconditionally, so that those loads are in flight during this
            // pass's parking and replay
            uint32_t pos[NQ], cnP[NQ];
            float pa[NQ], pb[NQ], pis2[NQ], yuP[NQ][DIM];
            {
                uint32_t run = incl - mine_tot;
#pragma unroll
                for (int q = 0; q < NQ; q++) {
                    pos[q] = run;
                    s_pos[lane * NQ + q] = run;
                    run += cn[q];
                    cnP[q] = cn[q];
                    pis2[q] = rcp(recA[q].s_src * recA[q].s_src);
                    pa[q] = B1 ? step2 * pis2[q] * (1.f - recA[q].w) : recA[q].w;
                    pb[q] = B1 ? step2 * pis2[q] * recA[q].w : 0.f;
#pragma unroll
                    for (int t = 0; t < DIM; t++) yuP[q][t] = yu[q][t];
                    untangle(yuP[q]);
                }
            }
            if (lane == 0) s_pos[CH] = total;
            AE_TICK(7)
#pragma unroll
            for (int q = 0; q < NQ; q++) recA[q] = recB[q];
            count_and_gather(cb + CH);
            load_recs(cb + 2 * CH, recB);
            AE_TICK(3)
            const uint64_t lo = tb_v > cb ? tb_v : cb;
            const uint64_t hi = te_v < cb + CH ? te_v : cb + CH;
            const bool has_range = hi > lo;
#pragma nounroll
            for (uint32_t w0 = 0; w0 < total; w0 += EC) {  // one window unless the pass holds more than EC pushes
#pragma unroll
                for (int q = 0; q < NQ; q++) {
                    for (uint32_t r = 0; r < cnP[q]; r++) {
                        const uint32_t e = pos[q] + r - w0;  // wraps below the window
                        if (e < (uint32_t)EC) {
#pragma unroll
                            for (int t = 0; t < DIM; t++) s_in_row[e * DIM + t] = yuP[q][t];
                            s_in_a[e] = pa[q];
                            s_in_b[e] = pb[q];
                            s_in_is2[e] = pis2[q];
                        }
                    }
                }
                wave_lds_sync();
                AE_TICK(8)
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
                if (a.store_mode == 0 && valid && len) st(v, yv);
                wave_lds_sync();
                AE_TICK(4)
            }
        }
    }
    if constexpr (COOP) { if (a.store_mode == 2 || a.store_mode == 3) st_rows(); }
    else if (valid && ((a.store_mode == 2 && any_push) || (a.store_mode == 3 && (any_push || nv)))) st(v, yv);
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
        for (int i = 7; i < 10; i++) atomicAdd(&a.prof[i], tk_acc[i]);  // finer split of the in-edge staging
    }
#undef AE_TICK
}


// rows of the LDS tile of the TILE kernels (= EC of NodeKernelCfg); the launcher needs n >= 2 x this
inline uint32_t node_kernel_tile_rows(int DIM) { return DIM <= 4 ? 1024u : (DIM <= 8 ? 256u : 128u); }

template <int DIM, bool PAD, int KMAX, bool TILE>
void launch_round_node_k(const NodeArgs& a, uint64_t nodes) {
    const unsigned grid = blocks_for(nodes, 64);
    if (a.b == 1.0f) hipLaunchKernelGGL((ce_round_node_kernel<DIM, PAD, true, KMAX, TILE>), dim3(grid), dim3(64), 0, stream(), a);
    else if constexpr (KMAX == 16 || KMAX == 32) hipLaunchKernelGGL((ce_round_node_kernel<DIM, PAD, false, KMAX, TILE>), dim3(grid), dim3(64), 0, stream(), a);
    else fail(AE_ERR_INVALID_ARG, "ce_round_node_kernel: b != 1 is instantiated for KMAX 16 / 32 only");
}
// rows of <= 8 / 12 / 16 / 24 / 32 neighbours (the exponent b != 1 only for 16 / 32: its powf code is large)
template <int DIM, bool PAD, bool TILE>
void launch_round_node_dim(ae_entropy_optim* o, const NodeArgs& a, uint64_t nodes) {
    const uint32_t k = o->g->max_nbng;
    const bool b1 = a.b == 1.0f;
    if (k <= 8 && b1) launch_round_node_k<DIM, PAD, 8, TILE>(a, nodes);
    else if (k <= 12 && b1) launch_round_node_k<DIM, PAD, 12, TILE>(a, nodes);
    else if (k <= 16) launch_round_node_k<DIM, PAD, 16, TILE>(a, nodes);
    else if (k <= 24 && b1) launch_round_node_k<DIM, PAD, 24, TILE>(a, nodes);
    else launch_round_node_k<DIM, PAD, 32, TILE>(a, nodes);
}
}  // namespace ae
