// Probe: does hipExtStreamCreateWithCUMask confine a stream's kernels to a CU subset on this box, and which (XCC, SE, CU)
// does a mask bit select?  Build: hipcc --offload-arch=gfx950 -O2 -o tools/probe_cumask tools/probe_cumask.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <set>
#include <vector>
#include <chrono>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void census_kernel(uint32_t* out, int spin) {
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = hw; out[blockIdx.x * 2 + 1] = xcc; }
    for (int i = 0; i < spin; i++) __builtin_amdgcn_s_sleep(64);  // keep the block alive so that the grid spreads over every CU it may use
}

__global__ void stream_kernel(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}

static int census(hipStream_t s, const char* name, uint32_t* d_out, int blocks) {
    CK(hipMemsetAsync(d_out, 0xff, blocks * 8, s));
    hipLaunchKernelGGL(census_kernel, dim3(blocks), dim3(256), 0, s, d_out, 200);
    CK(hipStreamSynchronize(s));
    std::vector<uint32_t> h(blocks * 2);
    CK(hipMemcpy(h.data(), d_out, blocks * 8, hipMemcpyDeviceToHost));
    std::set<uint32_t> cus, xccs;
    for (int b = 0; b < blocks; b++) {
        const uint32_t hw = h[b * 2], xcc = h[b * 2 + 1] & 0xf;
        const uint32_t cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        cus.insert(xcc << 16 | se << 8 | sh << 4 | cu);
        xccs.insert(xcc);
    }
    printf("%-28s distinct (xcc,se,sh,cu) %zu  xccs %zu :", name, cus.size(), xccs.size());
    int shown = 0;
    for (uint32_t c : cus) { if (shown++ < 12) printf(" x%u.se%u.%u.cu%u", c >> 16, (c >> 8) & 0xff, (c >> 4) & 0xf, c & 0xf); }
    printf("\n");
    return 0;
}

static double time_stream(hipStream_t s, const float4* a, float4* b, size_t n) {
    hipLaunchKernelGGL(stream_kernel, dim3(2048), dim3(256), 0, s, a, b, n);
    hipStreamSynchronize(s);
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < 5; r++) hipLaunchKernelGGL(stream_kernel, dim3(2048), dim3(256), 0, s, a, b, n);
    hipStreamSynchronize(s);
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / 5;
}

int main() {
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    printf("device %s CUs %d\n", p.name, p.multiProcessorCount);
    uint32_t* d_out;
    const int blocks = 4096;
    CK(hipMalloc(&d_out, blocks * 8));
    hipStream_t full;
    CK(hipStreamCreateWithFlags(&full, hipStreamNonBlocking));
    if (census(full, "unmasked stream", d_out, blocks)) return 1;
    const size_t n = (size_t)1 << 26;  // 1 GiB of float4
    float4 *a, *b;
    CK(hipMalloc(&a, n * 16));
    CK(hipMalloc(&b, n * 16));
    CK(hipMemset(a, 0, n * 16));
    printf("copy 1 GiB unmasked: %.3f ms\n", time_stream(full, a, b, n) * 1e3);
    struct { const char* name; uint32_t m[8]; } masks[] = {
        {"bits 0..31", {0xffffffffu, 0, 0, 0, 0, 0, 0, 0}},
        {"bits 32..255", {0, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu}},
        {"bits 0..7", {0xffu, 0, 0, 0, 0, 0, 0, 0}},
        {"every 8th bit", {0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u}},
        {"bits 224..255", {0, 0, 0, 0, 0, 0, 0, 0xffffffffu}},
    };
    for (auto& mk : masks) {
        hipStream_t s;
        hipError_t e = hipExtStreamCreateWithCUMask(&s, 8, mk.m);
        if (e != hipSuccess) { printf("%s: hipExtStreamCreateWithCUMask -> %s\n", mk.name, hipGetErrorString(e)); continue; }
        if (census(s, mk.name, d_out, blocks)) return 1;
        printf("    copy 1 GiB: %.3f ms\n", time_stream(s, a, b, n) * 1e3);
        hipStreamDestroy(s);
    }
    return 0;
}
