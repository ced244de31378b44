// Microbenchmark: register-chained 64->64 layers on the fp32 MFMA, one wave = 32 rows.
// A operand = weights (global, fragment order), B operand = previous accumulators.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CHECK(x) do { hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} } while(0)

template <int KT, int NT, bool LDSW>
__device__ __forceinline__ void layer(const float4* __restrict__ w, const float* __restrict__ bias, const f32x16 (&in)[KT], f32x16 (&out)[NT], int lane) {
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 a = w[((nt * KT + kt) * 4 + j) * 64 + lane];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, in[kt][4 * j + 0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, in[kt][4 * j + 1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, in[kt][4 * j + 2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, in[kt][4 * j + 3], acc, 0, 0, 0);
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float v = acc[r] + bias[nt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)];
      out[nt][r] = v > 0.f ? v : 0.01f * v;
    }
  }
}

template <int NL>
__global__ __launch_bounds__(64) void chain_kernel(const float4* __restrict__ w, const float* __restrict__ bias, float* out, int steps, long long* cyc) {
  const int lane = threadIdx.x;
  f32x16 x[2], y[2];
  for (int t = 0; t < 2; ++t)
    for (int r = 0; r < 16; ++r) x[t][r] = 0.001f * (float)((lane * 17 + r * 3 + t + blockIdx.x) % 97) - 0.04f;
  long long t0 = clock64();
  for (int s = 0; s < steps; ++s) {
#pragma unroll 1
    for (int l = 0; l < NL; l += 2) {
      layer<2, 2, false>(w + (size_t)l * 1024, bias + l * 64, x, y, lane);
      layer<2, 2, false>(w + (size_t)(l + 1) * 1024, bias + (l + 1) * 64, y, x, lane);
    }
  }
  long long t1 = clock64();
  float acc = 0.f;
  for (int t = 0; t < 2; ++t)
    for (int r = 0; r < 16; ++r) acc += x[t][r];
  out[blockIdx.x * 64 + lane] = acc;
  if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}

int main(int argc, char** argv) {
  const int NL = 10, steps = 40;
  int wgs_per_cu = argc > 1 ? atoi(argv[1]) : 4;
  const int grid = 256 * wgs_per_cu;
  std::vector<float> hw((size_t)NL * 4096), hb(NL * 64);
  for (auto& v : hw) v = ((rand() % 2001) - 1000) * 1e-4f;
  for (auto& v : hb) v = ((rand() % 2001) - 1000) * 1e-4f;
  float *dw, *db, *dout; long long* dc;
  CHECK(hipMalloc(&dw, hw.size() * 4)); CHECK(hipMalloc(&db, hb.size() * 4));
  CHECK(hipMalloc(&dout, grid * 64 * 4)); CHECK(hipMalloc(&dc, grid * 8));
  CHECK(hipMemcpy(dw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(db, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int it = 0; it < 3; ++it) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(chain_kernel<NL>, dim3(grid), dim3(64), 0, 0, (const float4*)dw, db, dout, steps, dc);
    hipEventRecord(e1);
    CHECK(hipDeviceSynchronize());
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> hc(grid);
    CHECK(hipMemcpy(hc.data(), dc, grid * 8, hipMemcpyDeviceToHost));
    double mean = 0; for (auto c : hc) mean += c; mean /= grid;
    const double mfmas = (double)steps * NL * 64;
    const double flop = (double)grid * mfmas * 32 * 32 * 2 * 2;
    printf("wgs/cu=%d  %.3f ms  %.1f TFLOP/s  ticks/MFMA(clock64)=%.1f\n", wgs_per_cu, ms, flop / ms * 1e-9, mean / mfmas);
  }
  return 0;
}
