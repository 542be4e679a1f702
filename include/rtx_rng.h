/*
 * rtx_rng.h — definition of the counter-based sampler of the MI355X renderer core.
 *
 * The reference draws from one std::mt19937 per worker thread (RTBase/Sampling.h:13-29,
 * seeded at RTBase/Renderer.h:102-104); tiles are handed to threads dynamically, so its
 * streams are not reproducible run to run. The core replaces that generator by a stateless
 * function of (seed, stream, item, sample, dimension): the d-th call of Sampler::next()
 * (RTBase/Sampling.h:10) inside one pixel sample / light path returns rtx_rng_float of
 * the same five numbers on every device and in every launch order.
 *
 * Plain C, usable from host C/C++, from HIP device code and from test code.
 */
#ifndef RTX_RNG_H
#define RTX_RNG_H

#include <stdint.h>

#if defined(__HIPCC__)
#define RTX_RNG_FN __host__ __device__ static inline
#else
#define RTX_RNG_FN static inline
#endif

/* streams: which family of paths a key belongs to */
#define RTX_STREAM_CAMERA 0u /* pixel samples: item = y*W+x, sample = SPP index (1-based)  */
#define RTX_STREAM_LIGHT  1u /* light-tracing paths: item = path index, sample = pass      */
#define RTX_STREAM_VPL    2u /* instant-radiosity light paths: item = path index, sample = pass */

/* PCG output permutation (RXS-M-XS 32) used as an integer hash */
RTX_RNG_FN uint32_t rtx_rng_hash(uint32_t x)
{
    uint32_t state = x * 747796405u + 2891336453u;
    uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
    return (word >> 22u) ^ word;
}

/* key of one path */
RTX_RNG_FN uint32_t rtx_rng_key(uint32_t seed, uint32_t stream, uint32_t item, uint32_t sample)
{
    uint32_t h = rtx_rng_hash(seed ^ (stream * 0x9E3779B9u));
    h = rtx_rng_hash(h + item);
    h = rtx_rng_hash(h ^ (sample * 0x85EBCA6Bu));
    return h;
}

/* dimension d of the path: a float in [1e-8, 1), clamped from below like MTRandom::next
 * (RTBase/Sampling.h:24-28) */
RTX_RNG_FN float rtx_rng_float(uint32_t key, uint32_t dim)
{
    uint32_t bits = rtx_rng_hash(key + dim * 0xC2B2AE35u);
    float r = (float)(bits >> 8) * (1.0f / 16777216.0f);
    return r < 1e-8f ? 1e-8f : r;
}

#endif /* RTX_RNG_H */
