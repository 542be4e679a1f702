// Throughput of returning device-scope atomics as a function of the number of distinct addresses and their spacing:
// every wave of a resident grid does K atomicAdd (lane 0, result fed back so they are dependent) on counter (wave % C).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_atomics(unsigned* ctr, unsigned n_ctr, unsigned stride_words, unsigned k, unsigned* sink)
{
    const unsigned wave = blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    unsigned acc = 0;
    if ((threadIdx.x & 63) == 0) {
        unsigned* p = ctr + (size_t)(wave % n_ctr) * stride_words;
        for (unsigned i = 0; i < k; i++) acc += atomicAdd(p, 1u + (acc & 1u));
        sink[wave] = acc;
    }
}

int main()
{
    unsigned *ctr, *sink;
    const unsigned blocks = 1280, waves = blocks * 4, K = 64;
    CHECK(hipMalloc(&ctr, 64u << 20)); CHECK(hipMalloc(&sink, waves * 4));
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (unsigned stride : {1u, 32u, 64u, 1024u}) {
        for (unsigned c : {1u, 8u, 64u, 512u, 4096u, 5120u}) {
            if ((size_t)c * stride * 4 > (64u << 20)) continue;
            CHECK(hipMemset(ctr, 0, 64u << 20));
            hipLaunchKernelGGL(k_atomics, dim3(blocks), dim3(256), 0, 0, ctr, c, stride, K, sink);
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(a));
            hipLaunchKernelGGL(k_atomics, dim3(blocks), dim3(256), 0, 0, ctr, c, stride, K, sink);
            CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
            float ms; CHECK(hipEventElapsedTime(&ms, a, b));
            const double total = (double)waves * K;
            printf("stride %5u B  counters %5u : %8.3f ms  %8.1f M atomics/s  %7.3f us per atomic per address  (%.2f us per dependent atomic of a wave)\n",
                   stride * 4, c, ms, total / ms / 1e3, ms * 1e3 / (total / c), ms * 1e3 / K);
        }
    }
    return 0;
}
