// Issue behaviour of v_mfma_f32_32x32x16_bf16 on one SIMD (round 3):
//  1. clocks per MFMA for a chain on ONE accumulator (each MFMA's SrcC is the previous result) against two and three interleaved
//     accumulators, at 1 and 2 waves per SIMD;
//  2. two waves on a SIMD, one issuing only MFMAs (one accumulator), the other only VALU (independent v_fma): each alone and
//     both together - do VALU instructions of ANOTHER wave fill the matrix pipe's shadow?
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/bf16_issue tools/ubench/bf16_issue.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

template <int NACC>
__global__ __launch_bounds__(512) void chain(float *out, int iters, unsigned long long *ticks) {
    f16v acc[3] = {{0}, {0}, {0}};
    bf8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(0.5f + i); }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 12; ++j) acc[j % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j % NACC], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int k = 0; k < 3; ++k) for (int r = 0; r < 16; ++r) s += acc[k][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) { ticks[2 * (threadIdx.x >> 6)] = t0; ticks[2 * (threadIdx.x >> 6) + 1] = t1; }
}

// waves 0..3 of the block (one per SIMD): MFMA only; waves 4..7 (the second wave of each SIMD): VALU only.  mode bit 0: MFMA
// waves work, bit 1: VALU waves work.
__global__ __launch_bounds__(512) void mixed(float *out, int iters, int mode, unsigned long long *ticks) {
    const int wave = threadIdx.x >> 6;
    const bool mf = wave < 4;
    f16v acc = {0};
    bf8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(0.5f + i); }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.25f + i;
    const float c1 = 1.0001f, c2 = 0.5f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (mf) {
        if (mode & 1)
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int j = 0; j < 12; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
            }
    } else {
        if (mode & 2)
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int j = 0; j < 12; ++j)
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = __builtin_fmaf(v[i], c1, c2);      // 96 independent-ish VALU per iteration
            }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int r = 0; r < 16; ++r) s += acc[r];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) { ticks[2 * wave] = t0; ticks[2 * wave + 1] = t1; }
}

int main() {
    float *out; unsigned long long *ticks, tk[16];
    (void)hipMalloc(&out, 1 << 22); (void)hipMalloc(&ticks, 16 * 8);
    const int iters = 20000;
    for (int nacc = 1; nacc <= 3; ++nacc)
        for (int wps = 1; wps <= 2; ++wps) {
            for (int rep = 0; rep < 2; ++rep) {
                if (nacc == 1) hipLaunchKernelGGL(chain<1>, dim3(256), dim3(256 * wps), 0, 0, out, iters, ticks);
                if (nacc == 2) hipLaunchKernelGGL(chain<2>, dim3(256), dim3(256 * wps), 0, 0, out, iters, ticks);
                if (nacc == 3) hipLaunchKernelGGL(chain<3>, dim3(256), dim3(256 * wps), 0, 0, out, iters, ticks);
                (void)hipMemcpy(tk, ticks, sizeof(tk), hipMemcpyDeviceToHost);
            }
            unsigned long long lo = ~0ull, hi = 0;
            for (int w = 0; w < 4 * wps; ++w) { if (tk[2 * w] < lo) lo = tk[2 * w]; if (tk[2 * w + 1] > hi) hi = tk[2 * w + 1]; }
            printf("chain on %d accumulator(s), %d wave(s) per SIMD: %.1f clocks per MFMA per SIMD\n", nacc, wps,
                   (double)(hi - lo) / iters / 12 / wps);
        }
    for (int mode = 1; mode <= 3; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(mixed, dim3(256), dim3(512), 0, 0, out, iters, mode, ticks);
            (void)hipMemcpy(tk, ticks, sizeof(tk), hipMemcpyDeviceToHost);
        }
        const double mf = (double)(tk[1] - tk[0]) / iters, va = (double)(tk[9] - tk[8]) / iters;
        printf("mode %d (%s): MFMA wave %.0f clocks per 12 MFMAs (%.1f each), VALU wave %.0f clocks per 96 v_fma (%.2f each)\n", mode,
               mode == 1 ? "MFMA wave alone" : mode == 2 ? "VALU wave alone" : "both on one SIMD", mf, mf / 12, va, va / 96);
    }
    return 0;
}
