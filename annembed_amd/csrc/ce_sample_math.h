// ce_sample_math.h -- the arithmetic of one SGD sample (ce_optim_edge_shannon, src/embedder.rs:1167-1302) on rows held in
// registers, and the "published row" idiom (a row becomes visible through its own agent-scope stores over an all-ones fill).
// Shared by the sequential-equivalent dataflow kernel (ce.hip) and the event-ordered kernel (ce_event.hip): both apply
// the reference's operations one for one -- coordinates f32, scalars f64 (:1207-1229), -ffp-contract=off.
#pragma once
#include "ce_internal.h"

namespace ae {

// whole rows of the coordinate array with the widest vector access the dimension allows
template <int DIM>
__device__ __forceinline__ void load_row(const float* __restrict__ y, uint32_t node, float* out) {
    const float* p = y + (uint64_t)node * DIM;
    if constexpr (DIM == 2) {
        float2 t = *reinterpret_cast<const float2*>(p);
        out[0] = t.x; out[1] = t.y;
    } else if constexpr (DIM % 4 == 0) {
#pragma unroll
        for (int q = 0; q < DIM / 4; q++) {
            float4 t = reinterpret_cast<const float4*>(p)[q];
            out[4 * q] = t.x; out[4 * q + 1] = t.y; out[4 * q + 2] = t.z; out[4 * q + 3] = t.w;
        }
    } else {
#pragma unroll
        for (int t = 0; t < DIM; t++) out[t] = p[t];
    }
}
template <int DIM>
__device__ __forceinline__ void store_row(float* __restrict__ y, uint32_t node, const float* in) {
    float* p = y + (uint64_t)node * DIM;
    if constexpr (DIM == 2) {
        *reinterpret_cast<float2*>(p) = make_float2(in[0], in[1]);
    } else if constexpr (DIM % 4 == 0) {
#pragma unroll
        for (int q = 0; q < DIM / 4; q++)
            reinterpret_cast<float4*>(p)[q] = make_float4(in[4 * q], in[4 * q + 1], in[4 * q + 2], in[4 * q + 3]);
    } else {
#pragma unroll
        for (int t = 0; t < DIM; t++) p[t] = in[t];
    }
}

// common part of the gradient coefficient, embedder.rs:1216-1222 / :1275-1281
__device__ __forceinline__ double grad_coeff(double d_scaled, double scale, double b) {
    if (b != 1.) {
        double cw = 1. / (1. + pow(d_scaled, b));
        return 2. * b * cw * pow(d_scaled, b - 1.) / (scale * scale);
    }
    double cw = 1. / (1. + d_scaled);
    return 2. * b * cw / (scale * scale);
}

// ce_optim_edge_shannon, embedder.rs:1167-1302, one sample, on rows held in registers, in its two kinds of steps:
// the attraction along the sampled edge (updates y_i and y_j, :1207-1238) ...
template <int DIM>
__device__ __forceinline__ void sample_attract(float* yi, float* yj, float* grad, float w, double scale, double b, double grad_step) {
#pragma unroll
    for (int t = 0; t < DIM; t++) grad[t] = 0.f;  // :1199
    const double weight = (double)w;                // :1202
    float acc = 0.f;
#pragma unroll
    for (int t = 0; t < DIM; t++) {  // :1207-1211
        float df = yi[t] - yj[t];
        acc += df * df;
    }
    const double d_ij_scaled = (double)acc / (scale * scale);  // :1214
    const double coeff = grad_coeff(d_ij_scaled, scale, b);
    if (d_ij_scaled > 0.) {  // :1223-1236
        const double alfa = (double)(1.0f / kProbaMin);
        const double coeff_repulsion = 1. / fmax(d_ij_scaled * d_ij_scaled, alfa);
        const double coeff_ij = fmax(grad_step * coeff * (-weight + (1. - weight) * coeff_repulsion), -0.49);
        const float cf = (float)coeff_ij;
#pragma unroll
        for (int t = 0; t < DIM; t++) grad[t] = (yj[t] - yi[t]) * cf;
    }
#pragma unroll
    for (int t = 0; t < DIM; t++) {  // :1237-1238
        yi[t] -= grad[t];
        yj[t] += grad[t];
    }
}
// ... and one repulsion from a negative sample (updates y_i only, :1267-1297; `grad` carries over, see the quirk below)
template <int DIM>
__device__ __forceinline__ void sample_repulse(float* yi, const float* yk, float* grad, double scale, double b, double grad_step) {
    float ak = 0.f;
#pragma unroll
    for (int t = 0; t < DIM; t++) {  // :1267-1271
        float df = yi[t] - yk[t];
        ak += df * df;
    }
    const double d_ik = (double)ak;
    const double d_ik_scaled = d_ik / (scale * scale);  // :1274
    const double cf2 = grad_coeff(d_ik_scaled, scale, b);
    if (d_ik > 0.) {  // :1286-1295
        const double coeff_repulsion = 1. / fmax(d_ik_scaled * d_ik_scaled, 1. / 16.);
        const double coeff_ik = fmin(grad_step * cf2 * coeff_repulsion, 2.);
        const float cf = (float)coeff_ik;
#pragma unroll
        for (int t = 0; t < DIM; t++) grad[t] = (yk[t] - yi[t]) * cf;
    }  // else: `gradient` keeps its previous value, as in the reference
#pragma unroll
    for (int t = 0; t < DIM; t++) yi[t] -= grad[t];  // :1297
}
// The sample's arithmetic in f64 (the reference's scalar type, embedder.rs:1207-1229) with ONE division per interaction -- for the modes
// that are validated statistically (AE_CE_SLICED, AE_CE_ORDERED); the bit-exact mode keeps the functions above.  The
// reference's formula spends four dependent f64 divisions on an attraction or a repulsion -- d / s^2, 1 / (1 + d), . / s^2,
// 1 / max(d^2, .) -- and a sample is six interactions: 24 IEEE division sequences (a dozen dependent f64 instructions each) in a
// row.  Measured on a rank's share of a configs[3] batch (31 k events per step: one wave per SIMD, nothing hides a dependent chain):
// 9.4 of a step's 23.5 us.  Algebraically the coefficient is num / ((1 + d) max(d^2, .)) with 1 / s^2 taken once per sample:
// one reciprocal per interaction, v_rcp_f64 refined by two Newton steps (what the division sequence itself does).  Still f64
// throughout; it agrees with the four-division form to a few ulps of f64 (~1e-15 relative) before the coefficient is rounded to
// the f32 it multiplies the coordinates with -- eight orders below that rounding.  In the ordered dataflow the divisions sat on the
// critical path of every dependency level (a successor of y_i waits for all six interactions of its predecessor).
__device__ __forceinline__ double rcp_f64(double x) {
    double r = __builtin_amdgcn_rcp(x);
    double e = __builtin_fma(-x, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-x, r, 1.0);
    return __builtin_fma(r, e, r);
}
template <int DIM>
__device__ __forceinline__ void attract_f64(float* yi, float* yj, float* grad, float w, double inv_s2, double b, double step) {
    float acc = 0.f;
#pragma unroll
    for (int t = 0; t < DIM; t++) { grad[t] = 0.f; const float df = yi[t] - yj[t]; acc += df * df; }
    const double d = (double)acc * inv_s2;
    if (d > 0.) {
        const double weight = (double)w;
        const double m = fmax(d * d, (double)(1.0f / kProbaMin));
        double coeff_ij;
        if (b == 1.) {   // 2 step / s^2 (-w + (1 - w) / m) / (1 + d)  =  2 step / s^2 ((1 - w) - w m) / ((1 + d) m)
            coeff_ij = 2. * step * inv_s2 * ((1. - weight) - weight * m) * rcp_f64((1. + d) * m);
        } else {
            const double pb = pow(d, b);
            coeff_ij = 2. * b * step * inv_s2 * (pb / d) * ((1. - weight) - weight * m) * rcp_f64((1. + pb) * m);
        }
        const float cf = (float)fmax(coeff_ij, -0.49);
#pragma unroll
        for (int t = 0; t < DIM; t++) grad[t] = (yj[t] - yi[t]) * cf;
    }
#pragma unroll
    for (int t = 0; t < DIM; t++) { yi[t] -= grad[t]; yj[t] += grad[t]; }
}
template <int DIM>
__device__ __forceinline__ void repulse_f64(float* yi, const float* yk, float* grad, double inv_s2, double b, double step) {
    float ak = 0.f;
#pragma unroll
    for (int t = 0; t < DIM; t++) { const float df = yi[t] - yk[t]; ak += df * df; }
    if (ak > 0.f) {
        const double d = (double)ak * inv_s2;
        const double m = fmax(d * d, 1. / 16.);
        double coeff_ik;
        if (b == 1.) {
            coeff_ik = 2. * step * inv_s2 * rcp_f64((1. + d) * m);
        } else {
            const double pb = pow(d, b);
            coeff_ik = 2. * b * step * inv_s2 * (pb / d) * rcp_f64((1. + pb) * m);
        }
        const float cf = (float)fmin(coeff_ik, 2.);
#pragma unroll
        for (int t = 0; t < DIM; t++) grad[t] = (yk[t] - yi[t]) * cf;
    }  // else: `gradient` keeps its previous value, as in the reference
#pragma unroll
    for (int t = 0; t < DIM; t++) yi[t] -= grad[t];
}

// The repulsion with its scalar coefficient in f32 (hardware reciprocals): for the modes whose negatives are read unsynchronised
// anyway (AE_CE_ORDERED, AE_CE_SLICED) -- the rounding of the coefficient (1e-7) is far below the difference between two
// admissible readings of the negative's row; the five dependent repulsions of a source event are on the critical path of its
// node's chain (4 dependent f64 divisions each).
template <int DIM>
__device__ __forceinline__ void sample_repulse_f32(float* yi, const float* yk, float* grad, float inv_s2, float b, float step) {
    float ak = 0.f;
#pragma unroll
    for (int t = 0; t < DIM; t++) { const float df = yi[t] - yk[t]; ak += df * df; }
    const float d = ak * inv_s2;
    if (ak > 0.f) {
        const float coeff = b == 1.f ? 2.0f * inv_s2 * __builtin_amdgcn_rcpf(1.0f + d)
                                     : 2.0f * b * __builtin_amdgcn_rcpf(1.0f + __powf(d, b)) * __powf(d, b - 1.0f) * inv_s2;
        const float cf = fminf(step * coeff * __builtin_amdgcn_rcpf(fmaxf(d * d, 1.0f / 16.0f)), 2.0f);
#pragma unroll
        for (int t = 0; t < DIM; t++) grad[t] = (yk[t] - yi[t]) * cf;
    }  // else: `gradient` keeps its previous value, as in the reference
#pragma unroll
    for (int t = 0; t < DIM; t++) yi[t] -= grad[t];
}
constexpr uint64_t kUnpublished64 = ~0ull;
constexpr uint32_t kUnpublished32 = ~0u;
template <int DIM>
__device__ __forceinline__ bool df_try_load_version(const float* __restrict__ ver, uint32_t pv, float* out) {
    const float* p = ver + (uint64_t)pv * DIM;  // version index = sample * 2 + slot
    bool ok = true;
    if constexpr (DIM % 2 == 0) {
#pragma unroll
        for (int q = 0; q < DIM / 2; q++) {
            const uint64_t bits = __hip_atomic_load(reinterpret_cast<const uint64_t*>(p) + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ok &= bits != kUnpublished64;
            out[2 * q] = __uint_as_float((uint32_t)bits);
            out[2 * q + 1] = __uint_as_float((uint32_t)(bits >> 32));
        }
    } else {
#pragma unroll
        for (int t = 0; t < DIM; t++) {
            const uint32_t bits = __hip_atomic_load(reinterpret_cast<const uint32_t*>(p) + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ok &= bits != kUnpublished32;
            out[t] = __uint_as_float(bits);
        }
    }
    return ok;
}
template <int DIM>
__device__ __forceinline__ void df_store_version(float* __restrict__ ver, uint64_t v, const float* in) {
    float* p = ver + v * DIM;
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

// A row of the coordinate array as the memory side has it NOW: agent-scope loads.  A plain or non-temporal load may be served
// by this XCD's L2, whose copy of a line another XCD's owner keeps rewriting is not refreshed before the launch ends (the
// per-XCD L2s are not coherent): negatives read that way were up to a whole window old -- measured as final CE +1 ... +3 %
// and the shortest edge-length quantiles -3 ... -12 % against the sequential loop.
template <int DIM>
__device__ __forceinline__ void load_row_coherent(const float* __restrict__ y, uint32_t node, float* out) {
    float tmp[DIM];
    (void)df_try_load_version<DIM>(y + (uint64_t)node * DIM, 0, tmp);
#pragma unroll
    for (int t = 0; t < DIM; t++) out[t] = tmp[t];
}

}  // namespace ae
