// Hand-off latency between lanes of different workgroups, (a) across the whole chip with agent-scope accesses (the primitive of
// ce_dataflow_kernel today) and (b) with every participating workgroup on ONE XCD -- workgroups are dealt to the 8 XCDs round
// robin, so a grid of 8 x as many workgroups in which only blockIdx % 8 == 0 take part sits on XCD 0 (checked through
// HW_REG_XCC_ID) -- where the XCD's own L2 is the coherence point and the poll can be an L1-bypassing load that hits L2.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_handoff_xcd.hip -o /tmp/ubench_handoff_xcd && /tmp/ubench_handoff_xcd
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// MODE 0: agent-scope load / store.  1: non-temporal load (L1 bypass, L2 hit), agent-scope store.  2: nt load, nt store.
// 3: workgroup-scope atomic load after an L1 invalidate (buffer_inv sc0), agent-scope store
template <int MODE>
__device__ __forceinline__ uint64_t poll(const uint64_t* p) {
    if constexpr (MODE == 0) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if constexpr (MODE == 3) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
    else return __builtin_nontemporal_load(p);
}
template <int MODE>
__device__ __forceinline__ void publish(uint64_t* p, uint64_t v) {
    if constexpr (MODE == 2) __builtin_nontemporal_store(v, p);
    else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int MODE>
__global__ void __launch_bounds__(64) pingpong(uint64_t* slots, int rounds, int active_lanes, int xcd_stride, unsigned int* xcc, unsigned int* fails) {
    if (blockIdx.x % xcd_stride != 0) return;
    const int blk = blockIdx.x / xcd_stride;
    const int pair = blk >> 1, side = blk & 1, lane = threadIdx.x;
    if (lane == 0) xcc[blk] = __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20) & 0xF;  // HW_REG_XCC_ID
    uint64_t* mine = slots + ((uint64_t)pair * 64 + lane) * 16 + (side ? 8 : 0);
    uint64_t* other = slots + ((uint64_t)pair * 64 + lane) * 16 + (side ? 0 : 8);
    const bool act = lane < active_lanes;
    uint64_t seq = 1;
    bool done = !act;
    uint64_t budget = 0;
    if (act && side == 0) publish<MODE>(mine, seq);
    while (!__all(done)) {
        if (!done) {
            const uint64_t v = poll<MODE>(other);
            if (v == seq) {
                if (side == 1) { publish<MODE>(mine, seq); seq++; }
                else { seq++; if (seq <= (uint64_t)rounds) publish<MODE>(mine, seq); }
                if (seq > (uint64_t)rounds) done = true;
            } else if (++budget > (1ull << 22)) { done = true; atomicAdd(fails, 1u); }
        }
    }
}

template <int MODE>
void run(const char* name, uint64_t* slots, unsigned int* xcc, unsigned int* fails, int blocks, int lanes, int xcd_stride) {
    const int rounds = 2000;
    CK(hipMemset(slots, 0, (size_t)4096 / 2 * 64 * 16 * 8));
    CK(hipMemset(fails, 0, 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(pingpong<MODE>, dim3(blocks * xcd_stride), dim3(64), 0, 0, slots, rounds, lanes, xcd_stride, xcc, fails);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned int hx[4096], hf;
    CK(hipMemcpy(hx, xcc, blocks * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(&hf, fails, 4, hipMemcpyDeviceToHost));
    unsigned mask = 0;
    for (int b = 0; b < blocks; b++) mask |= 1u << hx[b];
    printf("%-34s stride %d blocks %5d x %2d lanes: %7.3f us per hop   xcc mask 0x%02x  timed-out lanes %u\n", name, xcd_stride, blocks, lanes,
           ms * 1e3 / (2.0 * rounds), mask, hf);
}

int main() {
    uint64_t* slots;
    unsigned int *xcc, *fails;
    CK(hipMalloc(&slots, (size_t)4096 / 2 * 64 * 16 * 8));
    CK(hipMalloc(&xcc, 4096 * 4));
    CK(hipMalloc(&fails, 4));
    for (int stride : {1, 8}) {
        for (int blocks : {2, 128, 512, 940}) {
            if (stride == 8 && blocks > 512) continue;  // one XCD: 32 CUs x 16 one-wave workgroups resident
            for (int lanes : {1, 8, 64}) {
                run<0>("agent load / agent store", slots, xcc, fails, blocks, lanes, stride);
                run<1>("nt load / agent store", slots, xcc, fails, blocks, lanes, stride);
                run<2>("nt load / nt store", slots, xcc, fails, blocks, lanes, stride);
                run<3>("wg-scope load + L1 inv / agent store", slots, xcc, fails, blocks, lanes, stride);
            }
        }
    }
    return 0;
}
