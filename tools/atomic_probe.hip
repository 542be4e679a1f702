// atomic_probe.hip — what does a returning global atomicAdd cost when every wave of a full
// grid uses it to fetch work? One lane per wave adds to one of S counters (128 bytes apart),
// K times in a row, each add dependent on the previous result (like a work-fetch loop).
// Prints the wall time per launch, the aggregate rate and the per-wave latency per add.
//   hipcc --offload-arch=gfx950 -O3 -o atomic_probe atomic_probe.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void probe(uint32_t* counters, int S, int K, uint32_t* out)
{
    const uint32_t lane = threadIdx.x & 63;
    uint32_t seg = blockIdx.x % (uint32_t)S, acc = 0;
    for (int i = 0; i < K; i++) {
        uint32_t v = 0;
        if (lane == 0) v = atomicAdd(counters + seg * 32, 64u);
        v = (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
        acc += v;
        seg = (seg + (v & 0u)) % (uint32_t)S;      // keeps the chain dependent
    }
    if (lane == 0) out[blockIdx.x * 4 + threadIdx.x / 64] = acc;
}

int main()
{
    const int blocks = 1536, K = 64;
    uint32_t *counters, *out;
    CHECK(hipMalloc(&counters, 4096 * 128)); CHECK(hipMalloc(&out, blocks * 4 * 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int Ss[] = {1, 8, 64, 512, 4096};
    for (int S : Ss) {
        float ms = 0;
        for (int rep = 0; rep < 3; rep++) {
            CHECK(hipMemset(counters, 0, 4096 * 128));
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(probe, dim3(blocks), dim3(256), 0, 0, counters, S, K, out);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            CHECK(hipEventElapsedTime(&ms, e0, e1));
        }
        const double adds = (double)blocks * 4 * K;
        printf("S=%5d counters: %8.3f ms for %.0f adds  -> %7.1f Madds/s aggregate, %7.1f ns per add per counter, %7.2f us per dependent add per wave\n",
               S, ms, adds, adds / ms / 1e3, ms * 1e6 / (adds / S), ms * 1e3 / K);
    }
    return 0;
}
