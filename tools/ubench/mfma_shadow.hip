// Can VALU instructions issue in the shadow of a v_mfma_f32_32x32x2_f32 (64 clocks on the SIMD's matrix pipe)?
// One wave per SIMD; every MFMA is followed by NV independent v_fma_f32 on registers the MFMAs do not touch.
// If the fillers hide, clocks per MFMA stay at 64 until 4 * NV exceeds the shadow; if they serialise, 64 + 4 * NV.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/mfma_shadow tools/ubench/mfma_shadow.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NV>
__global__ __launch_bounds__(256) void k(int iters, float a, float b, float *sink, unsigned long long *ticks) {
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = a + i;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#define STEP(C)                                                           \
        C = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, C, 0, 0, 0);       \
        _Pragma("unroll") for (int i = 0; i < NV; ++i) v[i % 16] = __builtin_fmaf(v[i % 16], b, a);
        STEP(c0) STEP(c1) STEP(c2) STEP(c3)
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r] + v[r];
    if (s == 12345.678f) sink[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}

// the same for v_mfma_f32_32x32x16_bf16 (32 clocks): the question for the split-bf16 GEMM's operand split
template <int NV>
__global__ __launch_bounds__(256) void kb(int iters, float a, float b, float *sink, unsigned long long *ticks) {
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    bf16x8 x, y;
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = (__bf16)(a + i), y[i] = (__bf16)(b + i);
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = a + i;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#define STEPB(C)                                                          \
        C = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, C, 0, 0, 0);    \
        _Pragma("unroll") for (int i = 0; i < NV; ++i) v[i % 16] = __builtin_fmaf(v[i % 16], b, a);
        STEPB(c0) STEPB(c1) STEPB(c2) STEPB(c3)
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r] + v[r];
    if (s == 12345.678f) sink[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}

template <int NV>
void runb(float *sink, unsigned long long *ticks) {
    const int iters = 20000;
    hipLaunchKernelGGL(kb<NV>, dim3(256), dim3(256), 0, 0, iters, 1.0f, 0.5f, sink, ticks);
    hipLaunchKernelGGL(kb<NV>, dim3(256), dim3(256), 0, 0, iters, 1.0f, 0.5f, sink, ticks);
    unsigned long long h = 0;
    (void)hipMemcpy(&h, ticks, 8, hipMemcpyDeviceToHost);
    printf("bf16 32x32x16, NV=%2d independent v_fma per MFMA: %6.1f clocks per MFMA (serial would be %d)\n", NV, (double)h / (4.0 * iters),
           32 + 4 * NV);
}

template <int NV>
void run(float *sink, unsigned long long *ticks) {
    const int iters = 20000;
    hipLaunchKernelGGL(k<NV>, dim3(256), dim3(256), 0, 0, iters, 1.0f, 0.5f, sink, ticks);
    hipLaunchKernelGGL(k<NV>, dim3(256), dim3(256), 0, 0, iters, 1.0f, 0.5f, sink, ticks);
    unsigned long long h = 0;
    hipMemcpy(&h, ticks, 8, hipMemcpyDeviceToHost);
    printf("NV=%2d independent v_fma per MFMA: %6.1f clocks per MFMA (serial would be %d)\n", NV, (double)h / (4.0 * iters), 64 + 4 * NV);
}

int main() {
    float *sink;
    unsigned long long *ticks;
    hipMalloc(&sink, 4);
    hipMalloc(&ticks, 8);
    run<0>(sink, ticks); run<2>(sink, ticks); run<4>(sink, ticks); run<8>(sink, ticks); run<12>(sink, ticks); run<16>(sink, ticks);
    runb<0>(sink, ticks); runb<2>(sink, ticks); runb<4>(sink, ticks); runb<6>(sink, ticks); runb<8>(sink, ticks); runb<12>(sink, ticks);
    return 0;
}
