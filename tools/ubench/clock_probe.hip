// What clock does the chip sustain under fp32-MFMA load?  (the roofline peak of 157.3 TFLOP/s assumes 2.4 GHz.)
//
// Every wave issues a fixed number of v_mfma_f32_32x32x2_f32 on four independent accumulators.  One such instruction occupies
// the SIMD's MFMA pipe for 16 passes of 4 clocks = 64 shader clocks, so a wave alone on its SIMD needs n_mfma * 64 clocks:
// clocks / wall time = the shader clock actually sustained.  s_memtime and s_memrealtime are read on the same launch.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/clock_probe tools/ubench/clock_probe.hip && tools/ubench/clock_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void mfma_chain(int iters, float a, float b, float *sink, unsigned long long *ticks) {
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c3, 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
    if (s == 12345.678f) sink[0] = s;          // keeps the chain alive, never true
    if (threadIdx.x == 0) {
        ticks[2 * blockIdx.x] = t1 - t0;
        ticks[2 * blockIdx.x + 1] = r1 - r0;
    }
}

// the same with a dependent chain of VALU FMAs (4 clocks per wave64 instruction on a SIMD16)
__global__ __launch_bounds__(256) void valu_chain(int iters, float a, float b, float *sink, unsigned long long *ticks) {
    float x = a;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) x = __builtin_fmaf(x, b, a);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (x == 12345.678f) sink[0] = x;
    if (threadIdx.x == 0) {
        ticks[2 * blockIdx.x] = t1 - t0;
        ticks[2 * blockIdx.x + 1] = r1 - r0;
    }
}

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                \
            return 1;                                                              \
        }                                                                          \
    } while (0)

int main() {
    float *sink;
    unsigned long long *ticks, h[2];
    CK(hipMalloc(&sink, 4));
    CK(hipMalloc(&ticks, 16 * 4096));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    int clk_khz = 0;
    CK(hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0));
    printf("hipDeviceAttributeClockRate: %d MHz\n", clk_khz / 1000);
    const int iters = 400000;                    // 1.6 M MFMAs = 102.4 M clocks = 42.7 ms at 2.4 GHz
    const int grids[] = {1, 64, 256, 512, 1024};  // 256 threads each: 1 wave per SIMD of a CU; 1024 blocks = 4 waves per SIMD
    for (int kind = 0; kind < 2; ++kind)
        for (int gi = 0; gi < 5; ++gi) {
            const int g = grids[gi];
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipEventRecord(e0, 0));
                if (kind == 0) hipLaunchKernelGGL(mfma_chain, dim3(g), dim3(256), 0, 0, iters, 1.0f, 0.5f, sink, ticks);
                else hipLaunchKernelGGL(valu_chain, dim3(g), dim3(256), 0, 0, iters, 1.0f, 0.5f, sink, ticks);
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float ms = 0;
                CK(hipEventElapsedTime(&ms, e0, e1));
                CK(hipMemcpy(h, ticks, 16, hipMemcpyDeviceToHost));
                if (rep == 0) continue;
                // clocks a wave needs when it has the pipe to itself, times the waves that share a SIMD
                const double per_simd = (g + 255) / 256;
                const double clocks = kind == 0 ? 4.0 * iters * 64 * per_simd : 16.0 * iters * 4 * per_simd;
                printf("%s blocks=%4d (%g wave(s)/SIMD): wall %8.3f ms -> %7.1f MHz sustained;  s_memtime %llu ticks = %7.1f MHz, "
                       "s_memrealtime %llu ticks = %6.1f MHz  (over wave 0's own span)\n",
                       kind == 0 ? "mfma" : "valu", g, per_simd, ms, clocks / ms / 1e3, h[0], h[0] / ms / 1e3, h[1], h[1] / ms / 1e3);
            }
        }
    return 0;
}
