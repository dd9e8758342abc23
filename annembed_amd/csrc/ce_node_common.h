// ce_node_common.h -- device helpers shared by the node-centric (owner-computes) Hogwild kernels:
// ce_node.hip (lane-group / unfused kernels, host driver) and ce_node_round_*.hip (node-per-lane round kernel).
#pragma once
#include "ce_internal.h"

namespace ae {


constexpr uint32_t kTagEdgeCount = 0xFFFF0010u;
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
    int tile;        // 1: negatives of the uniform sampler from an LDS tile of consecutive rows (ce_node_round.h)
};

// PCG-RXS-M-XS 32 output hash: the fast mode's stream for the negative draws (the exact Philox stream
// of the oracle is used by AE_CE_SEQUENTIAL; this mode is validated statistically)
__host__ __device__ __forceinline__ uint32_t pcg_hash(uint32_t x) {
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
}  // namespace ae
