// Counter-based RNG shared by the rollout and the stand-alone categorical draw.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// ---------------------------------------------------------------------------------------------
// Philox4x32 counter RNG (throughput mode; parity mode injects eps / q instead)
// ---------------------------------------------------------------------------------------------
// Philox4x32 with 7 rounds (Salmon et al., SC'11: the 7-round variant is the fastest one that passes
// BigCrush; 10 is the conservative default).  The stream is ours, not torch's: parity is defined on
// injected noise, and both rollout kernels draw the same numbers for the same (seed, row, t, j).
#define DN_PHILOX_ROUNDS 7
__device__ __forceinline__ uint4 philox4x32(uint4 c, uint2 k) {
#pragma unroll
    for (int i = 0; i < DN_PHILOX_ROUNDS; ++i) {
        // one 32x32->64 product each (v_mad_u64_u32) instead of a v_mul_hi_u32 / v_mul_lo_u32 pair: integer
        // multiplies are quarter rate and sit on the critical VALU path of the goal draw
        const uint64_t p0 = (uint64_t)c.x * 0xD2511F53ull, p1 = (uint64_t)c.z * 0xCD9E8D57ull;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        c = make_uint4(hi1 ^ c.y ^ k.x, lo1, hi0 ^ c.w ^ k.y, lo0);
        k.x += 0x9E3779B9u;
        k.y += 0xBB67AE85u;
    }
    return c;
}
// stream 0: eps (uniform [0,1)), stream 1: q (Exp(1)); element j of (row, t)
__device__ __forceinline__ uint32_t rng_bits(uint64_t seed, uint32_t stream, uint32_t row, uint32_t t, uint32_t j) {
    uint4 c = make_uint4(row, t, j >> 2, stream);
    uint4 r = philox4x32(c, make_uint2((uint32_t)seed, (uint32_t)(seed >> 32)));
    const uint32_t s = j & 3u;
    return s == 0 ? r.x : s == 1 ? r.y : s == 2 ? r.z : r.w;
}
__device__ __forceinline__ float rng_uniform(uint64_t seed, uint32_t row, uint32_t t, uint32_t j) {
    return (float)(rng_bits(seed, 0, row, t, j) >> 8) * (1.0f / 16777216.0f);
}
// The goal draw of the in-kernel stream (round 3): ONE uniform per (row, t) - word 0 of block (row, t, 0, stream 1) - and the
// inverse of the cumulative weights: goal = min { j : w_0 + .. + w_j > u * (w_0 + .. + w_{G-1}) }.  An exact categorical sample
// like the exponential race argmax_j w_j / q_j that torch.multinomial runs (and that the kernels keep for INJECTED q, which is
// what parity with the reference is defined on), for 1 Philox block and G additions instead of G / 4 blocks and 2 G logarithms.
// Every kernel adds the weights in the order its layout gives; two kernels then differ only where u * total falls within
// rounding of a boundary (~1e-5 of the draws; the kernels' softmax values differ at that level anyway).
// 23 random bits + 1/2: exact in fp32 and strictly inside (0, 1).
__device__ __forceinline__ float rng_draw_u(uint64_t seed, uint32_t row, uint32_t t) {
    const uint4 r = philox4x32(make_uint4(row, t, 0u, 1u), make_uint2((uint32_t)seed, (uint32_t)(seed >> 32)));
    return ((float)(r.x >> 9) + 0.5f) * (1.0f / 8388608.0f);
}
__device__ __forceinline__ float rng_exp1(uint64_t seed, uint32_t row, uint32_t t, uint32_t j) {
    const float u = ((float)(rng_bits(seed, 1, row, t, j) >> 8) + 1.0f) * (1.0f / 16777216.0f);   // (0,1]
    return -logf(u) + 1e-30f;
}

