// streaming-read calibration: sum a 188 MB float array with float4 loads (what a tall-skinny product must at least do)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(256) sum4(const float4* __restrict__ x, size_t n4, float* out) {
    float s = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 v = x[i];
        s += v.x + v.y + v.z + v.w;
    }
    if (s == 123.456f) out[0] = s;
}
int main() {
    const size_t n = 60000ull * 784;
    float* d; float* o;
    hipMalloc(&d, n * 4); hipMalloc(&o, 4);
    hipMemset(d, 0, n * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int grid : {512, 1024, 2048, 4096, 8192}) {
        sum4<<<grid, 256>>>((const float4*)d, n / 4, o);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < 20; r++) sum4<<<grid, 256>>>((const float4*)d, n / 4, o);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("grid %5d: %.1f us per pass, %.2f TB/s\n", grid, ms / 20 * 1e3, n * 4.0 / (ms / 20 * 1e-3) / 1e12);
    }
    return 0;
}
