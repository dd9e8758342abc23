// Hand-off latency between lanes of DIFFERENT workgroups through agent-scope 8-byte atomics -- the primitive under the
// sequentially consistent CE modes (a sample's result travels from the owner of one row to the next sample that needs it).
// Lane l of block 2p and lane l of block 2p+1 play ping-pong: A stores a sequence number, B polls it and answers, A polls the
// answer.  Reported: microseconds per HOP (half a round trip), for a few pairs on an idle chip and for every lane of a
// C2-sized grid (940 one-wave blocks) polling at once -- the regime of ce_event_window_kernel / ce_dataflow_kernel.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_handoff.hip -o /tmp/ubench_handoff && /tmp/ubench_handoff
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void __launch_bounds__(64) pingpong(uint64_t* slots, int rounds, int active_lanes, unsigned long long* cycles) {
    const int pair = blockIdx.x >> 1, side = blockIdx.x & 1, lane = threadIdx.x;
    uint64_t* mine = slots + ((uint64_t)pair * 64 + lane) * 16 + (side ? 8 : 0);   // 64-byte apart: one line per direction
    uint64_t* other = slots + ((uint64_t)pair * 64 + lane) * 16 + (side ? 0 : 8);
    const bool act = lane < active_lanes;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    uint64_t seq = 1;
    bool done = !act;
    uint64_t budget = 0;
    if (act && side == 0) __hip_atomic_store(mine, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (!__all(done)) {
        if (!done) {
            const uint64_t v = __hip_atomic_load(other, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint64_t want = side == 0 ? seq : seq;  // A waits for B's echo of seq; B waits for A's seq
            if (v == want) {
                if (side == 1) { __hip_atomic_store(mine, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); seq++; }
                else { seq++; if (seq <= (uint64_t)rounds) __hip_atomic_store(mine, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
                if (seq > (uint64_t)rounds) done = true;
            } else if (++budget > (1ull << 26)) done = true;
        }
    }
    if (lane == 0) cycles[blockIdx.x] = __builtin_amdgcn_s_memtime() - t0;
}

int main() {
    const int max_blocks = 4096;
    uint64_t* slots;
    unsigned long long* cyc;
    CK(hipMalloc(&slots, (size_t)max_blocks / 2 * 64 * 16 * 8));
    CK(hipMalloc(&cyc, max_blocks * 8));
    const int rounds = 2000;
    for (int blocks : {2, 16, 256, 940, 1880}) {
        for (int lanes : {1, 8, 64}) {
            CK(hipMemset(slots, 0, (size_t)max_blocks / 2 * 64 * 16 * 8));
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0));
            CK(hipEventCreate(&e1));
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(pingpong, dim3(blocks), dim3(64), 0, 0, slots, rounds, lanes, cyc);
            CK(hipEventRecord(e1));
            CK(hipDeviceSynchronize());
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf("blocks %5d (pairs %4d) x %2d lanes: %7.3f us per hop\n", blocks, blocks / 2, lanes, ms * 1e3 / (2.0 * rounds));
        }
    }
    return 0;
}
