// gather_probe.hip — how fast does a CU read 128-byte records that every lane picks for itself?
//
// Three access shapes over the same random record indices (table in L2 / Infinity Cache):
//   A  per-lane record: lane l issues 7 x 16-byte loads of ITS record (what a one-ray-per-lane
//      BVH walk does: 64 different cache lines per load instruction)
//   B  8-lane cooperative: the 8 lanes of a group read the 8 x 16-byte pieces of ONE record per
//      instruction, 8 rounds for the 8 records of the group (one cache line per 8 lanes)
//   C  B + LDS transpose so that every lane ends up with its whole record in registers
// Prints records/s and the implied bytes/s.   hipcc --offload-arch=gfx950 -O3 -o gather_probe gather_probe.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t hash(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

template <int MODE>
__global__ __launch_bounds__(256) void probe(const float4* __restrict__ table, uint32_t n_rec, int iters, float* out)
{
    __shared__ float4 lds[256 * 8];
    const uint32_t tid = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63, g = lane & 7;
    float acc = 0;
    uint32_t idx = hash(tid) % n_rec;
    for (int it = 0; it < iters; it++) {
        float4 r[8];
        if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 7; j++) r[j] = table[(size_t)idx * 8 + j];
            r[7] = make_float4(0, 0, 0, 0);
        } else {
#pragma unroll
            for (int k = 0; k < 8; k++) {
                uint32_t other = __shfl(idx, (int)((lane & ~7u) + k), 64);
                r[k] = table[(size_t)other * 8 + g];     // piece g of the record of group lane k
            }
            if (MODE == 2) {
                // transpose through LDS: lane g holds piece g of records 0..7 -> lane k wants pieces 0..7 of record k
                float4* base = lds + (threadIdx.x & ~7u) * 8;
#pragma unroll
                for (int k = 0; k < 8; k++) base[k * 8 + g] = r[k];
                __builtin_amdgcn_s_waitcnt(0xC07F);
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int j = 0; j < 8; j++) r[j] = base[g * 8 + j];
            }
        }
        float s = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) s += r[j].x + r[j].y + r[j].z + r[j].w;
        acc += s;
        idx = hash(idx + __float_as_uint(s) * 0u + it * 977u + tid) % n_rec;   // next record depends on the loaded data's arrival
        if (s == 12345.678f) idx = 0;
    }
    out[tid] = acc;
}

int main(int argc, char** argv)
{
    const uint32_t n_rec = argc > 1 ? atoi(argv[1]) : 90000;   // 90k x 128 B = 11.5 MB (the 1M-triangle soup's node array)
    const int iters = 256, blocks = 2048;
    std::vector<float> h((size_t)n_rec * 32);
    for (size_t i = 0; i < h.size(); i++) h[i] = (float)(i % 97) * 0.01f;
    float4* table; float* out;
    CHECK(hipMalloc(&table, h.size() * 4)); CHECK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    CHECK(hipMemcpy(table, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const char* names[3] = {"A per-lane record (7 loads/lane)", "B 8-lane cooperative (8 loads/lane, 8 records/group)", "C cooperative + LDS transpose"};
    for (int mode = 0; mode < 3; mode++) {
        for (int rep = 0; rep < 2; rep++) {
            CHECK(hipEventRecord(e0));
            if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(blocks), dim3(256), 0, 0, table, n_rec, iters, out);
            if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(256), 0, 0, table, n_rec, iters, out);
            if (mode == 2) hipLaunchKernelGGL(probe<2>, dim3(blocks), dim3(256), 0, 0, table, n_rec, iters, out);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep == 1) {
                double recs = (double)blocks * 256 * iters;
                printf("%-58s %8.3f ms  %7.2f Grecords/s  %7.2f TB/s (128 B/record)\n", names[mode], ms, recs / ms / 1e6, recs * 128 / ms / 1e9);
            }
        }
    }
    return 0;
}
