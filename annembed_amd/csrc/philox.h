// philox.h -- Philox4x32-10 counter RNG and the library's stream convention (host + device).
//
// The reference draws from unseeded thread RNGs (src/embedder.rs:1121,1182) and from
// Xoshiro256++ / Ziggurat (src/tools/svdapprox.rs:70-73); neither is reproducible nor vendored, so
// the build defines its own reproducible stream:
//   key     = (seed lo, seed hi)
//   counter = (c0, c1, c2, block)   block = 0,1,2,... as 4-word blocks are consumed
//   CE sample s of batch `iter`: c0,c1 = s (u64), c2 = iter.   Other uses: c2 = tag (common.h).
// Words are consumed in order; a u64 is (word << 32) | next word; an index in [0,n) is the high
// 64 bits of u64 * n; an f32 in [0,1) is (word >> 8) * 2^-24.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace ae {

__host__ __device__ inline void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

struct PhiloxStream {
    uint32_t k0, k1, c0, c1, c2, blk;
    uint32_t buf[4];
    int pos;
    __host__ __device__ PhiloxStream(uint64_t seed, uint64_t c01, uint32_t c2_)
        : k0((uint32_t)seed), k1((uint32_t)(seed >> 32)), c0((uint32_t)c01), c1((uint32_t)(c01 >> 32)), c2(c2_),
          blk(0), pos(4) {}
    __host__ __device__ inline uint32_t u32() {
        if (pos == 4) {
            philox4x32_10(c0, c1, c2, blk++, k0, k1, buf);
            pos = 0;
        }
        // static indexing keeps buf in registers
        uint32_t v = pos == 0 ? buf[0] : pos == 1 ? buf[1] : pos == 2 ? buf[2] : buf[3];
        pos++;
        return v;
    }
    __host__ __device__ inline uint64_t u64() {
        uint64_t hi = u32();
        uint64_t lo = u32();
        return (hi << 32) | lo;
    }
    __host__ __device__ inline uint64_t index(uint64_t n) {
#if defined(__HIP_DEVICE_COMPILE__)
        return __umul64hi(u64(), n);
#else
        return (uint64_t)(((unsigned __int128)u64() * n) >> 64);
#endif
    }
    __host__ __device__ inline float f32() { return (float)(u32() >> 8) * (1.0f / 16777216.0f); }
};

// Box-Muller on a word pair (N(0,1) f32), used for Omega and the projection noise
__host__ __device__ inline void box_muller(uint32_t w0, uint32_t w1, float& z0, float& z1) {
    float u1 = (float)((w0 >> 8) + 1u) * (1.0f / 16777216.0f);
    float u2 = (float)(w1 >> 8) * (1.0f / 16777216.0f);
    float r = sqrtf(-2.0f * logf(u1));
    float a = 6.28318530717958647692f * u2;
    z0 = r * cosf(a);
    z1 = r * sinf(a);
}

}  // namespace ae
