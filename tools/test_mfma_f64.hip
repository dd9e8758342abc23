// probes the output layout of v_mfma_f64_16x16x4_f64: D = A(16x4) * B(4x16) with A[i][k] = 100 i + k, B[k][j] = (k == 0) * j + (k == 1)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4_t __attribute__((ext_vector_type(4)));
__global__ void probe(double* out) {
    const int lane = threadIdx.x;
    const int c = lane & 15, k = lane >> 4;
    double a = 100.0 * c + k;                 // A[i = c][k]
    double b = (k == 0) ? (double)c : (k == 1 ? 1.0 : 0.0);  // B[k][j = c]
    d4_t acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    for (int q = 0; q < 4; q++) out[lane * 4 + q] = acc[q];
}
int main() {
    double* d; hipMalloc(&d, 256 * 8);
    probe<<<1, 64>>>(d);
    double h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    // expected D[i][j] = A[i][0]*j + A[i][1] = 100 i * j + (100 i + 1)
    int ok0 = 1, ok1 = 1;
    for (int lane = 0; lane < 64; lane++) for (int q = 0; q < 4; q++) {
        int j = lane % 16;
        int i0 = 4 * (lane / 16) + q, i1 = (lane / 16) + 4 * q;
        double e0 = 100.0 * i0 * j + 100.0 * i0 + 1, e1 = 100.0 * i1 * j + 100.0 * i1 + 1;
        if (h[lane * 4 + q] != e0) ok0 = 0;
        if (h[lane * 4 + q] != e1) ok1 = 0;
    }
    printf("layout i=4*(lane/16)+q: %d   layout i=(lane/16)+4q: %d\n", ok0, ok1);
    printf("lane0: %g %g %g %g  lane16: %g %g %g %g lane1: %g %g\n", h[0], h[1], h[2], h[3], h[64], h[65], h[66], h[67], h[4], h[5]);
    return 0;
}
