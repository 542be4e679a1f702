// What does the "which counters are live?" look-up cost when every wave of a resident grid does it at once?
// mode 0: every lane of a wave reads its own 128-byte line (64 lines, the same 64 for all waves), agent-scope load
// mode 1: all lanes read one word of line (wave % 64)
// mode 2: like 0 with plain (cacheable) loads
// mode 3: one atomic on line (wave % 64); mode 4: that atomic, then the look-up of mode 0; mode 5: that atomic, then one word of the same line
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k(unsigned* ctr, int mode, unsigned k_iter, unsigned* sink)
{
    const unsigned wave = blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    unsigned acc = 0;
    for (unsigned i = 0; i < k_iter; i++) {
        unsigned v;
        if (mode == 0) v = __hip_atomic_load(ctr + lane * 32 + (acc & 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (mode == 1) v = __hip_atomic_load(ctr + (wave % 64) * 32 + (acc & 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (mode == 2) v = ((volatile unsigned*)ctr)[lane * 32 + (acc & 1)];
        else if (mode == 3) { unsigned j = 0; if (lane == 0) j = atomicAdd(ctr + (wave % 64) * 32, 1u); j = __builtin_amdgcn_readfirstlane(j); v = j; }
        else if (mode == 4) { unsigned j = 0; if (lane == 0) j = atomicAdd(ctr + (wave % 64) * 32, 1u); j = __builtin_amdgcn_readfirstlane(j); v = j + __hip_atomic_load(ctr + lane * 32 + (j & 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        else { unsigned j = 0; if (lane == 0) j = atomicAdd(ctr + (wave % 64) * 32, 1u); j = __builtin_amdgcn_readfirstlane(j); v = j + __hip_atomic_load(ctr + (wave % 64) * 32 + 2 + (j & 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        acc += v + 1;
    }
    if (acc == 0xdeadbeef) sink[wave] = acc;
}

int main()
{
    unsigned *ctr, *sink;
    const unsigned blocks = 1280, K = 16;
    CHECK(hipMalloc(&ctr, 1 << 20)); CHECK(hipMalloc(&sink, blocks * 4 * 4)); CHECK(hipMemset(ctr, 0, 1 << 20));
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int mode = 0; mode < 6; mode++) {
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, ctr, mode, K, sink); CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(a));
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, ctr, mode, K, sink);
        CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
        float ms; CHECK(hipEventElapsedTime(&ms, a, b));
        printf("mode %d: %.3f ms for %u dependent look-ups by each of %u waves: %.2f us per look-up\n", mode, ms, K, blocks * 4, ms * 1e3 / K);
    }
    return 0;
}
