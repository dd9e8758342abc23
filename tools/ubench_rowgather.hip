// Microbenchmark: random gathers of WIDE coordinate rows (d = 8 / 16: 32 / 64 bytes) out of a footprint larger than
// the Infinity Cache -- the access pattern of the CE loop at the C4 / C5 shapes -- with the row held by one lane
// (the node-per-lane kernel: 2 / 4 dwordx4 loads per lane, every lane on its own cache line) against the row
// spread over a lane group (G lanes x 16 bytes: one request per row).
// hipcc --offload-arch=gfx950 -O3 tools/ubench_rowgather.hip -o /tmp/ubr && /tmp/ubr
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)

__device__ inline uint32_t pcg(uint32_t x) { uint32_t s = x * 747796405u + 2891336453u; uint32_t w = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u; return (w >> 22u) ^ w; }
using f4 = __attribute__((ext_vector_type(4))) float;

// MODE 0: row per lane (Q = ROWB/16 dwordx4 loads per lane);  MODE 1: G = ROWB/16 lanes per row, one dwordx4 each;
// R = independent rows in flight per lane (MODE 0) / per lane group (MODE 1) and iteration
template <int MODE, int ROWB, int R>
__global__ void __launch_bounds__(64) k(const f4* __restrict__ y, uint32_t nrows, int iters, float* sink) {
    constexpr int Q = ROWB / 16;
    const uint32_t lane = threadIdx.x, gid = blockIdx.x * 64u + lane;
    float acc = 0.f;
    for (int it = 0; it < iters; it++) {
        if constexpr (MODE == 0) {
            f4 v[R][Q];
#pragma unroll
            for (int r = 0; r < R; r++) {
                const uint32_t row = __umulhi(pcg(gid * 977u + (uint32_t)(it * R + r) * 0x9E3779B9u), nrows);
#pragma unroll
                for (int q = 0; q < Q; q++) v[r][q] = __builtin_nontemporal_load(y + (uint64_t)row * Q + q);
            }
#pragma unroll
            for (int r = 0; r < R; r++)
#pragma unroll
                for (int q = 0; q < Q; q++) acc += v[r][q].x + v[r][q].w;
        } else {
            f4 v[R];
            const uint32_t grp = gid / Q, sub = gid % Q;
#pragma unroll
            for (int r = 0; r < R; r++) {
                const uint32_t row = __umulhi(pcg(grp * 977u + (uint32_t)(it * R + r) * 0x9E3779B9u), nrows);
                v[r] = __builtin_nontemporal_load(y + (uint64_t)row * Q + sub);
            }
#pragma unroll
            for (int r = 0; r < R; r++) acc += v[r].x + v[r].w;
        }
    }
    if (acc == 123.456f) *sink = acc;
}

template <int MODE, int ROWB, int R>
void run(const f4* y, uint32_t nrows, float* sink, const char* name) {
    const int blocks = 256 * 16, iters = 64;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((k<MODE, ROWB, R>), dim3(blocks), dim3(64), 0, 0, y, nrows, 2, sink); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); hipLaunchKernelGGL((k<MODE, ROWB, R>), dim3(blocks), dim3(64), 0, 0, y, nrows, iters, sink); CK(hipEventRecord(b)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    const double lanes = (double)blocks * 64, rows = (MODE == 0 ? lanes : lanes / (ROWB / 16)) * iters * R;
    printf("  %-44s %7.2f G rows/s  %7.1f GB/s\n", name, rows / (ms * 1e-3) / 1e9, rows * ROWB / (ms * 1e-3) / 1e9);
}

int main() {
    float* sink; CK(hipMalloc(&sink, 4));
    for (uint64_t mb : {4ull, 64ull, 400ull, 3200ull}) {
        const uint64_t bytes = mb << 20;
        f4* y; CK(hipMalloc(&y, bytes)); CK(hipMemset(y, 0, bytes));
        printf("footprint %llu MiB\n", (unsigned long long)mb);
        run<0, 32, 6>(y, (uint32_t)(bytes / 32), sink, "32-B rows, row per lane (2 x 16 B), 6 rows");
        run<1, 32, 6>(y, (uint32_t)(bytes / 32), sink, "32-B rows, 2 lanes per row, 6 rows");
        run<1, 32, 12>(y, (uint32_t)(bytes / 32), sink, "32-B rows, 2 lanes per row, 12 rows");
        run<0, 64, 6>(y, (uint32_t)(bytes / 64), sink, "64-B rows, row per lane (4 x 16 B), 6 rows");
        run<1, 64, 6>(y, (uint32_t)(bytes / 64), sink, "64-B rows, 4 lanes per row, 6 rows");
        run<1, 64, 12>(y, (uint32_t)(bytes / 64), sink, "64-B rows, 4 lanes per row, 12 rows");
        run<1, 64, 24>(y, (uint32_t)(bytes / 64), sink, "64-B rows, 4 lanes per row, 24 rows");
        run<1, 128, 6>(y, (uint32_t)(bytes / 128), sink, "128-B rows, 8 lanes per row, 6 rows");
        run<1, 128, 12>(y, (uint32_t)(bytes / 128), sink, "128-B rows, 8 lanes per row, 12 rows");
        run<1, 128, 24>(y, (uint32_t)(bytes / 128), sink, "128-B rows, 8 lanes per row, 24 rows");
        CK(hipFree(y));
    }
    return 0;
}
