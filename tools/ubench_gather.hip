// Microbenchmark: random small accesses to a coordinate array (CE-loop access pattern) under
// different cache policies / atomics.  hipcc --offload-arch=gfx950 -O3 tools/ubench_gather.hip -o /tmp/ub
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)

__device__ inline uint32_t lcg(uint32_t& s){ s = s*1664525u + 1013904223u; return s; }
__device__ inline uint64_t ridx(uint32_t& s, uint64_t n){ uint64_t r = ((uint64_t)lcg(s)<<32)|lcg(s); return __umul64hi(r, n); }

template<int MODE>
__global__ void __launch_bounds__(256) k(float2* y, uint64_t n, int iters, float* sink){
  uint32_t s = (blockIdx.x*blockDim.x + threadIdx.x)*2654435761u + 12345u;
  float acc = 0.f;
  for(int it=0; it<iters; it++){
    uint64_t i = ridx(s, n);
    if constexpr (MODE==0){ float2 v = y[i]; acc += v.x + v.y; }                                    // plain load 8B
    else if constexpr (MODE==1){ uint64_t v = __hip_atomic_load((uint64_t*)&y[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); acc += (float)(v&0xff); } // sc1 load
    else if constexpr (MODE==2){ unsafeAtomicAdd(&y[i].x, 1.0f); unsafeAtomicAdd(&y[i].y, 1.0f); }   // 2 hw float atomics, no return
    else if constexpr (MODE==3){ float a = unsafeAtomicAdd(&y[i].x, 1.0f); float b = unsafeAtomicAdd(&y[i].y, 1.0f); acc += a+b; } // returning
    else if constexpr (MODE==4){ y[i] = make_float2(acc, (float)it); }                               // plain store 8B
    else if constexpr (MODE==5){ uint64_t old = *(volatile uint64_t*)&y[i]; atomicCAS((unsigned long long*)&y[i], old, old+1); } // load + CAS64
    else if constexpr (MODE==6){ __hip_atomic_fetch_add(&y[i].x, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); __hip_atomic_fetch_add(&y[i].y, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);} // wg-scope atomics
    else if constexpr (MODE==7){ uint64_t v = __builtin_nontemporal_load((uint64_t*)&y[i]); acc += (float)(v&0xff); }     // nt load
    else if constexpr (MODE==8){ float4 v = ((float4*)y)[i>>1]; acc += v.x+v.w; }                    // plain 16B
    else if constexpr (MODE==10){ atomicAdd((unsigned int*)&y[i].x, 1u); atomicAdd((unsigned int*)&y[i].y, 1u); }   // 2x u32 atomics
    else if constexpr (MODE==11){ atomicAdd((unsigned long long*)&y[i], 0x100000001ull); }        // 1x u64 atomic (packed 2x32)
    else if constexpr (MODE==12){ __hip_atomic_store((uint64_t*)&y[i], (uint64_t)it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } // sc1 store 8B
    else if constexpr (MODE==13){ unsigned long long o = atomicAdd((unsigned long long*)&y[i], 0x100000001ull); acc += (float)(o&0xff); } // u64 atomic returning
    else if constexpr (MODE==14){ unsafeAtomicAdd((double*)&y[i], 1.0); }   // f64 atomic add
    else if constexpr (MODE==9){ float2 v = y[i]; v.x += 1.f; v.y += 1.f; y[i] = v; }               // racy RMW
  }
  if(acc == 123.456f) *sink = acc;
}

template<int MODE> double run(float2* y, uint64_t n, int iters, float* sink, int blocks){
  hipEvent_t a,b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, y, n, 4, sink); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a)); hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, y, n, iters, sink); CK(hipEventRecord(b)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms,a,b));
  return (double)blocks*256*iters/ (ms*1e-3) / 1e9;
}

int main(){
  float* sink; CK(hipMalloc(&sink, 4));
  const char* names2[] = {"2x u32 atomic add","1x u64 atomic add","sc1 store 8B","1x u64 atomic ret","1x f64 atomic add"};
  const char* names[] = {"plain load 8B","sc1(agent) load 8B","2x f32 atomic add noret","2x f32 atomic add ret","plain store 8B","load+CAS64","2x f32 atomic wg-scope","nt load 8B","plain load 16B","racy RMW 8B"};
  for (uint64_t n : {60000ull, 1650000ull, 44000000ull}) {
    float2* y; CK(hipMalloc(&y, n*sizeof(float2))); CK(hipMemset(y, 0, n*sizeof(float2)));
    printf("n=%llu (%.1f MB)\n", (unsigned long long)n, n*8/1e6);
    int blocks = 256*8, iters = 256;
    printf("  %-28s %8.2f Gop/s\n", names[0], run<0>(y,n,iters,sink,blocks));
    printf("  %-28s %8.2f Gop/s\n", names[8], run<8>(y,n,iters,sink,blocks));
    printf("  %-28s %8.2f Gop/s\n", names[1], run<1>(y,n,iters,sink,blocks));
    printf("  %-28s %8.2f Gop/s\n", names[7], run<7>(y,n,iters,sink,blocks));
    printf("  %-28s %8.2f Gop/s (pairs)\n", names[2], run<2>(y,n,iters,sink,blocks));
    printf("  %-28s %8.2f Gop/s (pairs)\n", names[3], run<3>(y,n,iters,sink,blocks));
    printf("  %-28s %8.2f Gop/s (pairs)\n", names[6], run<6>(y,n,iters,sink,blocks));
    printf("  %-28s %8.2f Gop/s\n", names[4], run<4>(y,n,iters,sink,blocks));
    printf("  %-28s %8.2f Gop/s\n", names[5], run<5>(y,n,iters,sink,blocks));
    printf("  %-28s %8.2f Gop/s\n", names[9], run<9>(y,n,iters,sink,blocks));
    printf("  %-28s %8.2f Gop/s (pairs)\n", names2[0], run<10>(y,n,iters,sink,blocks));
    printf("  %-28s %8.2f Gop/s\n", names2[1], run<11>(y,n,iters,sink,blocks));
    printf("  %-28s %8.2f Gop/s\n", names2[2], run<12>(y,n,iters,sink,blocks));
    printf("  %-28s %8.2f Gop/s\n", names2[3], run<13>(y,n,iters,sink,blocks));
    printf("  %-28s %8.2f Gop/s\n", names2[4], run<14>(y,n,iters,sink,blocks));
    // lost-update check: total of y after N increments
    CK(hipMemset(y, 0, n*sizeof(float2)));
    hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, y, n, 64, sink); CK(hipDeviceSynchronize());
    std::vector<float2> h(n); CK(hipMemcpy(h.data(), y, n*sizeof(float2), hipMemcpyDeviceToHost));
    double tot=0; for(auto&v:h) tot+=v.x; printf("  atomic agent: applied %.0f of %.0f increments\n", tot, (double)blocks*256*64);
    CK(hipMemset(y, 0, n*sizeof(float2)));
    hipLaunchKernelGGL(k<6>, dim3(blocks), dim3(256), 0, 0, y, n, 64, sink); CK(hipDeviceSynchronize());
    CK(hipMemcpy(h.data(), y, n*sizeof(float2), hipMemcpyDeviceToHost));
    tot=0; for(auto&v:h) tot+=v.x; printf("  atomic wg-scope: applied %.0f of %.0f increments\n", tot, (double)blocks*256*64);
    CK(hipMemset(y, 0, n*sizeof(float2)));
    hipLaunchKernelGGL(k<9>, dim3(blocks), dim3(256), 0, 0, y, n, 64, sink); CK(hipDeviceSynchronize());
    CK(hipMemcpy(h.data(), y, n*sizeof(float2), hipMemcpyDeviceToHost));
    tot=0; for(auto&v:h) tot+=v.x; printf("  racy RMW: applied %.0f of %.0f increments\n", tot, (double)blocks*256*64);
    CK(hipFree(y));
  }
  return 0;
}
