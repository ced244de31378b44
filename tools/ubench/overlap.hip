// Microbenchmark: do VALU instructions of one wave overlap with MFMAs of the other wave on the same SIMD?
// Each wave alternates a block of NM dependent 32x32x2 fp32 MFMAs and a block of NV dependent VALU FMAs that
// consume the MFMA result and feed the next MFMA block (the structure of the rollout kernel's layers).
//   ./overlap.bin <waves per SIMD> <NM> <NV>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CHECK(x) do { hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} } while(0)

template <int NM, int NV>
__global__ __launch_bounds__(64) void k(float *out, int iters, long long *cyc) {
  const int lane = threadIdx.x;
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.001f * (float)((lane + r) % 13);
  float a = 0.5f + 0.001f * lane, b = 0.25f;
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < NM; ++m) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    float v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = acc[r];
#pragma unroll
    for (int j = 0; j < NV / 16; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = __builtin_fmaf(v[r], 0.999f, 0.001f);
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = v[r];
    b = v[0] * 1e-3f;
  }
  long long t1 = clock64();
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += acc[r];
  out[blockIdx.x * 64 + lane] = s;
  if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NM, int NV>
void run(int wps, int iters) {
  const int grid = 256 * 4 * wps;
  float *dout; long long *dc;
  CHECK(hipMalloc(&dout, grid * 64 * 4)); CHECK(hipMalloc(&dc, grid * 8));
  for (int rep = 0; rep < 2; ++rep) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NM, NV>), dim3(grid), dim3(64), 0, 0, dout, iters, dc);
    hipEventRecord(e1);
    CHECK(hipDeviceSynchronize());
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long *hc = (long long *)malloc(grid * 8);
    CHECK(hipMemcpy(hc, dc, grid * 8, hipMemcpyDeviceToHost));
    double mean = 0; for (int i = 0; i < grid; ++i) mean += hc[i]; mean /= grid;
    if (rep) printf("waves/SIMD=%d NM=%d NV=%d: %.3f ms, clock64 ticks per iteration per wave = %.0f (MFMA-only ideal 64*NM = %d)\n", wps, NM, NV, ms, mean / iters, 64 * NM);
    free(hc);
  }
  hipFree(dout); hipFree(dc);
}

int main(int argc, char **argv) {
  const int iters = 2000;
  for (int wps = 1; wps <= 2; ++wps) {
    run<16, 0>(wps, iters);
    run<0, 256>(wps, iters);
    run<16, 256>(wps, iters);
    run<16, 128>(wps, iters);
    run<16, 64>(wps, iters);
  }
  return 0;
}
