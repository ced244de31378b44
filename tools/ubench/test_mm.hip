// Unit check of the v2 chain GEMM helpers against a host reference (exact small integers).
#include "../../dagnet_amd/csrc/rollout_v2.hip"
#include <stdio.h>
#include <vector>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} } while(0)

__global__ void k_mm(const float* wpk, int off, int off2, const float* X, float* Y, float* YT) {
  const int lane = threadIdx.x, lh = lane >> 5, lc = lane & 31;
  v2::WBuf wp; wp.init(wpk);
  T16 x[2], y[2], yt[2];
  for (int nt = 0; nt < 2; ++nt) for (int r = 0; r < 16; ++r) x[nt][r] = X[lc * 64 + nt * 32 + FR(r) + 4 * lh];
  v2::zero(y); v2::zero(yt);
  v2::WBlk pre = v2::wload(wp, off, lane);
  pre = v2::mm<2, 2, false>(y, wp, off, x, lane, pre, off2);
  pre = v2::mm<2, 2, true>(yt, wp, off2, x, lane, pre, off);
  for (int nt = 0; nt < 2; ++nt) for (int r = 0; r < 16; ++r) {
    Y[lc * 64 + nt * 32 + FR(r) + 4 * lh] = y[nt][r];            // S: lane=row, reg=feature
    YT[(FR(r) + 4 * lh) * 64 + nt * 32 + lc] = yt[nt][r];         // T: lane=feature, reg=row
  }
}

int main() {
  const int N = 64, K = 64;
  std::vector<float> W(N * K), W2(N * K), X(32 * K), pk(2 * 4 * 1024);
  for (auto& v : W) v = (rand() % 7) - 3;
  for (auto& v : W2) v = (rand() % 5) - 2;
  for (auto& v : X) v = (rand() % 9) - 4;
  auto pack = [&](const std::vector<float>& w, float* dst) {
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int j = 0; j < 4; ++j) for (int l = 0; l < 64; ++l) for (int i = 0; i < 4; ++i) {
      int n = a * 32 + (l & 31), k = b * 32 + 8 * j + 4 * (l >> 5) + i;
      dst[((((size_t)a * 2 + b) * 4 + j) * 64 + l) * 4 + i] = w[n * K + k];
    }
  };
  pack(W, pk.data()); pack(W2, pk.data() + 4096);
  float *dpk, *dX, *dY, *dYT;
  CHECK(hipMalloc(&dpk, pk.size() * 4)); CHECK(hipMalloc(&dX, X.size() * 4)); CHECK(hipMalloc(&dY, 32 * 64 * 4)); CHECK(hipMalloc(&dYT, 32 * 64 * 4));
  CHECK(hipMemcpy(dpk, pk.data(), pk.size() * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_mm, dim3(1), dim3(64), 0, 0, dpk, 0, 4096, dX, dY, dYT);
  CHECK(hipDeviceSynchronize());
  std::vector<float> Y(32 * 64), YT(32 * 64);
  CHECK(hipMemcpy(Y.data(), dY, Y.size() * 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(YT.data(), dYT, YT.size() * 4, hipMemcpyDeviceToHost));
  int bad = 0, badT = 0;
  for (int m = 0; m < 32; ++m) for (int n = 0; n < 64; ++n) {
    float r = 0, r2 = 0;
    for (int k = 0; k < 64; ++k) { r += W[n * K + k] * X[m * 64 + k]; r2 += W2[n * K + k] * X[m * 64 + k]; }
    if (r != Y[m * 64 + n]) ++bad;
    if (r2 != YT[m * 64 + n]) ++badT;
  }
  for (int n = 0; n < 8; ++n) { float r = 0; for (int k = 0; k < 64; ++k) r += W[n * K + k] * X[0 * 64 + k]; printf("n=%d ref %.0f got %.0f\n", n, r, Y[n]); }
  printf("S<-S mismatches: %d / 2048   T<-S mismatches: %d / 2048\n", bad, badT);
  return bad + badT != 0;
}
