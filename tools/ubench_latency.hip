// dependent-chain latency of the access flavours used by the CE kernels (one wave per CU, 64 lanes chasing)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)
template<int MODE>
__global__ void __launch_bounds__(64) chase(uint64_t* a, uint64_t n, int iters, uint64_t* out){
  uint64_t i = (blockIdx.x*64 + threadIdx.x) % n;
  uint64_t* own = a + n + (blockIdx.x*64 + threadIdx.x);   // private store target
  for(int it=0; it<iters; it++){
    uint64_t v;
    if constexpr (MODE==0) v = a[i];
    else if constexpr (MODE==1) v = __builtin_nontemporal_load(&a[i]);
    else if constexpr (MODE==2) v = __hip_atomic_load(&a[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if constexpr (MODE==3) { __hip_atomic_store(own, (uint64_t)it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); v = __builtin_nontemporal_load(&a[i]); }
    else if constexpr (MODE==4) { *own = (uint64_t)it; v = __builtin_nontemporal_load(&a[i]); }
    else if constexpr (MODE==5) { __builtin_nontemporal_store((uint64_t)it, own); v = __builtin_nontemporal_load(&a[i]); }
    i = v;
  }
  out[blockIdx.x*64+threadIdx.x] = i;
}
template<int MODE> void run(const char* name, uint64_t* a, uint64_t n, uint64_t* out, int blocks){
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  int iters = 2000;
  hipLaunchKernelGGL(chase<MODE>, dim3(blocks), dim3(64), 0, 0, a, n, 10, out); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); hipLaunchKernelGGL(chase<MODE>, dim3(blocks), dim3(64), 0, 0, a, n, iters, out); CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms,e0,e1));
  printf("  %-36s blocks=%5d  %7.3f us per dependent step\n", name, blocks, ms*1e3/iters);
}
int main(){
  for (uint64_t n : {60000ull, 11000000ull}) {
    std::vector<uint64_t> h(n); uint64_t x = 12345;
    for (uint64_t i=0;i<n;i++){ x = x*6364136223846793005ull + 1442695040888963407ull; h[i] = (x>>20) % n; }
    uint64_t *a, *out; CK(hipMalloc(&a, (n + 4096*64)*8)); CK(hipMalloc(&out, 4096*64*8));
    CK(hipMemcpy(a, h.data(), n*8, hipMemcpyHostToDevice));
    printf("n=%llu\n", (unsigned long long)n);
    for (int blocks : {256, 940, 4096}) {
      run<0>("plain load", a, n, out, blocks);
      run<1>("nt load", a, n, out, blocks);
      run<2>("sc1 (agent) load", a, n, out, blocks);
      run<3>("sc1 store + nt load", a, n, out, blocks);
      run<4>("plain store + nt load", a, n, out, blocks);
      run<5>("nt store + nt load", a, n, out, blocks);
    }
    CK(hipFree(a)); CK(hipFree(out));
  }
}
