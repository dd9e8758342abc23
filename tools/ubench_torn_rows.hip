// Microbenchmark: can a coordinate row be read TORN while its owner rewrites it?  (verdict r3, item 7)
// The faithful CE modes read a negative sample's row "as the memory system has it" while the node's owner may be rewriting it in the
// same launch; the reference never sees half a row (negatives go through `try_read`, embedder.rs:1257-1265).  A row of D floats is
// written with all D columns equal to one counter value; a reader that finds two different values in a row has read it torn.
// Access patterns = the library's own:
//   writers: W0 lane group, one 16-byte store per lane (coop_store, ce_slice_kernels.h: a row is ONE request of G adjacent lanes)
//            W1 one lane, D/4 16-byte stores (store_row: chain kernels, sequential / ordered modes' commit)
//            W2 one lane, D/2 8-byte agent-scope stores (df_store_version / store_row_agent)
//   readers: R0 lane group, one 16-byte load per lane (coop_issue: rows and tile of the time-sliced mode)
//            R1 one lane, D/4 16-byte loads (load_row)
//            R2 one lane, D/2 8-byte agent-scope loads (load_row_coherent: the negatives of the ordered mode)
// Rows are hot (4096 rows: everything in L2, the worst case for tearing: no DRAM latency hides the window).
// hipcc --offload-arch=gfx950 -O3 tools/ubench_torn_rows.hip -o tools/ubench_torn_rows && tools/ubench_torn_rows
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ inline uint32_t pcg(uint32_t x) { uint32_t s = x * 747796405u + 2891336453u; uint32_t w = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u; return (w >> 22u) ^ w; }
using f4 = __attribute__((ext_vector_type(4))) float;

// blocks [0, writers) write, the rest read; `stop` is raised by the host
template <int D, int WMODE, int RMODE>
__global__ void __launch_bounds__(64) torn_kernel(float* __restrict__ y, uint32_t nrows, unsigned writers, int iters, unsigned long long* __restrict__ stats) {
    constexpr int G = D / 4;
    const uint32_t lane = threadIdx.x;
    if (blockIdx.x < writers) {
        for (int it = 0; it < iters; it++) {
            const float val = (float)(it + 1);
            if constexpr (WMODE == 0) {
                const uint32_t grp = (blockIdx.x * 64u + lane) / G, sub = lane % G;
                const uint32_t row = __umulhi(pcg(grp * 977u + (uint32_t)it * 0x9E3779B9u), nrows);
                f4 v = {val, val, val, val};
                *reinterpret_cast<f4*>(y + (uint64_t)row * D + sub * 4) = v;
            } else if constexpr (WMODE == 1) {
                const uint32_t row = __umulhi(pcg((blockIdx.x * 64u + lane) * 977u + (uint32_t)it * 0x9E3779B9u), nrows);
                f4 v = {val, val, val, val};
#pragma unroll
                for (int q = 0; q < G; q++) *reinterpret_cast<f4*>(y + (uint64_t)row * D + q * 4) = v;
            } else {
                const uint32_t row = __umulhi(pcg((blockIdx.x * 64u + lane) * 977u + (uint32_t)it * 0x9E3779B9u), nrows);
                const uint32_t b = __float_as_uint(val);
                const uint64_t bits = ((uint64_t)b << 32) | b;
#pragma unroll
                for (int q = 0; q < D / 2; q++)
                    __hip_atomic_store(reinterpret_cast<uint64_t*>(y + (uint64_t)row * D) + q, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        return;
    }
    unsigned long long torn = 0, reads = 0;
    for (int it = 0; it < iters; it++) {
        if constexpr (RMODE == 0) {
            const uint32_t grp = (blockIdx.x * 64u + lane) / G, sub = lane % G;
            const uint32_t row = __umulhi(pcg(grp * 31337u + (uint32_t)it * 0x85EBCA6Bu), nrows);
            const f4 v = *reinterpret_cast<const f4*>(y + (uint64_t)row * D + sub * 4);
            bool bad = v.x != v.y || v.x != v.z || v.x != v.w;
            // the group's lanes compare their pieces
            float first = __shfl(v.x, (int)(lane & ~(uint32_t)(G - 1)));
            bad = bad || v.x != first;
            const unsigned long long m = __ballot(bad);
            if (lane == 0) {  // rows with any bad lane
                unsigned long long rows_bad = 0;
                for (int g = 0; g < 64 / G; g++) rows_bad += ((m >> (g * G)) & ((1ull << G) - 1ull)) ? 1 : 0;
                torn += rows_bad;
                reads += 64 / G;
            }
        } else if constexpr (RMODE == 1) {
            const uint32_t row = __umulhi(pcg((blockIdx.x * 64u + lane) * 31337u + (uint32_t)it * 0x85EBCA6Bu), nrows);
            f4 v[G];
#pragma unroll
            for (int q = 0; q < G; q++) v[q] = *reinterpret_cast<const f4*>(y + (uint64_t)row * D + q * 4);
            bool bad = false;
#pragma unroll
            for (int q = 0; q < G; q++) bad = bad || v[q].x != v[0].x || v[q].y != v[0].x || v[q].z != v[0].x || v[q].w != v[0].x;
            torn += bad ? 1 : 0;
            reads += 1;
        } else {
            const uint32_t row = __umulhi(pcg((blockIdx.x * 64u + lane) * 31337u + (uint32_t)it * 0x85EBCA6Bu), nrows);
            uint64_t w[D / 2];
#pragma unroll
            for (int q = 0; q < D / 2; q++) w[q] = __hip_atomic_load(reinterpret_cast<const uint64_t*>(y + (uint64_t)row * D) + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            bool bad = false;
#pragma unroll
            for (int q = 0; q < D / 2; q++) bad = bad || w[q] != w[0] || (uint32_t)w[q] != (uint32_t)(w[q] >> 32);
            torn += bad ? 1 : 0;
            reads += 1;
        }
    }
    for (int off = 32; off > 0; off >>= 1) { torn += __shfl_xor(torn, off); reads += __shfl_xor(reads, off); }
    if (lane == 0) { atomicAdd(&stats[0], torn); atomicAdd(&stats[1], reads); }
}

template <int D, int WMODE, int RMODE>
void run(const char* name) {
    const uint32_t nrows = 4096;
    float* y;
    unsigned long long* stats;
    CK(hipMalloc(&y, (size_t)nrows * D * 4));
    CK(hipMemset(y, 0, (size_t)nrows * D * 4));
    CK(hipMalloc(&stats, 16));
    CK(hipMemset(stats, 0, 16));
    const unsigned writers = 512, readers = 1536;
    const int iters = 20000;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((torn_kernel<D, WMODE, RMODE>), dim3(writers + readers), dim3(64), 0, 0, y, nrows, writers, iters, stats);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[2];
    CK(hipMemcpy(h, stats, 16, hipMemcpyDeviceToHost));
    const double writes = (double)writers * 64 * iters / (WMODE == 0 ? D / 4 : 1);
    printf("D=%2d  %-44s torn %10llu of %.3e row reads (%.2e per read), %.3e row writes into %u rows in %.1f ms\n", D, name, h[0], (double)h[1],
           h[1] ? (double)h[0] / (double)h[1] : 0., writes, nrows, ms);
    CK(hipFree(y)); CK(hipFree(stats));
}

int main() {
    run<8, 0, 0>("lane-group store / lane-group load (sliced)");
    run<16, 0, 0>("lane-group store / lane-group load (sliced)");
    run<8, 1, 1>("row-per-lane 16-B stores / 16-B loads");
    run<16, 1, 1>("row-per-lane 16-B stores / 16-B loads");
    run<8, 0, 1>("lane-group store / row-per-lane loads");
    run<8, 2, 2>("8-B agent stores / 8-B agent loads (ordered)");
    run<16, 2, 2>("8-B agent stores / 8-B agent loads (ordered)");
    run<2, 2, 2>("8-B agent store / load, d = 2: one granule");
    return 0;
}
