// Exactness (small integers) and speed of the fp32 MFMA GEMMs in dagnet_amd/csrc/gemm_f32.hip.
#include "../../dagnet_amd/csrc/gemm_f32.hip"
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CHECK(x) do { hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} } while(0)

static int check_nt(int M, int N, int K, int pad) {
  const int lda = K + pad, ldw = K + pad, ldc = N + 3;
  std::vector<float> A((size_t)M * lda), W((size_t)N * ldw), b(N), C((size_t)M * ldc);
  for (auto& v : A) v = (rand() % 7) - 3;
  for (auto& v : W) v = (rand() % 5) - 2;
  for (auto& v : b) v = (rand() % 9) - 4;
  float *dA, *dW, *db, *dC;
  CHECK(hipMalloc(&dA, A.size() * 4)); CHECK(hipMalloc(&dW, W.size() * 4)); CHECK(hipMalloc(&db, N * 4)); CHECK(hipMalloc(&dC, C.size() * 4));
  CHECK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(db, b.data(), N * 4, hipMemcpyHostToDevice)); CHECK(hipMemset(dC, 0xff, C.size() * 4));
  CHECK(dn_gemm_nt(dA, lda, dW, ldw, db, dC, ldc, M, N, K, gemm::ACT_LRELU, 0.5f, 0));
  CHECK(hipDeviceSynchronize());
  CHECK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
  int bad = 0;
  for (int m = 0; m < M; ++m) for (int n = 0; n < N; ++n) {
    float r = b[n];
    for (int k = 0; k < K; ++k) r += A[(size_t)m * lda + k] * W[(size_t)n * ldw + k];
    r = r > 0 ? r : 0.5f * r;
    if (r != C[(size_t)m * ldc + n]) ++bad;
  }
  printf("NT M=%d N=%d K=%d: %d mismatches\n", M, N, K, bad);
  hipFree(dA); hipFree(dW); hipFree(db); hipFree(dC);
  return bad;
}

static int check_tn(int M, int N, int K) {
  const int lda = N + 2, ldb = K + 4;
  std::vector<float> A((size_t)M * lda), B((size_t)M * ldb), C((size_t)N * K);
  for (auto& v : A) v = (rand() % 7) - 3;
  for (auto& v : B) v = (rand() % 5) - 2;
  float *dA, *dB, *dC, *dP;
  CHECK(hipMalloc(&dA, A.size() * 4)); CHECK(hipMalloc(&dB, B.size() * 4)); CHECK(hipMalloc(&dC, C.size() * 4));
  CHECK(hipMalloc(&dP, dn_gemm_tn_workspace(M, N, K) * 4));
  CHECK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
  float *dBias; CHECK(hipMalloc(&dBias, N * 4));
  CHECK(dn_gemm_tn_idx(dA, lda, dB, ldb, nullptr, dC, K, dBias, M, N, K, dP, 0, 0));
  CHECK(hipDeviceSynchronize());
  CHECK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
  std::vector<float> bias(N); CHECK(hipMemcpy(bias.data(), dBias, N * 4, hipMemcpyDeviceToHost));
  int bad = 0;
  for (int n = 0; n < N; ++n) { double r = 0; for (int m = 0; m < M; ++m) r += A[(size_t)m * lda + n]; if ((float)r != bias[n]) ++bad; }
  for (int n = 0; n < N; ++n) for (int k = 0; k < K; ++k) {
    double r = 0;
    for (int m = 0; m < M; ++m) r += (double)A[(size_t)m * lda + n] * B[(size_t)m * ldb + k];
    if ((float)r != C[(size_t)n * K + k]) ++bad;
  }
  printf("TN M=%d N=%d K=%d: %d mismatches\n", M, N, K, bad);
  hipFree(dA); hipFree(dB); hipFree(dC); hipFree(dP);
  return bad;
}

static void time_nt(int M, int N, int K) {
  float *dA, *dW, *dC;
  CHECK(hipMalloc(&dA, (size_t)M * K * 4)); CHECK(hipMalloc(&dW, (size_t)N * K * 4)); CHECK(hipMalloc(&dC, (size_t)M * N * 4));
  CHECK(hipMemset(dA, 0, (size_t)M * K * 4)); CHECK(hipMemset(dW, 0, (size_t)N * K * 4));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 2; ++i) CHECK(dn_gemm_nt(dA, K, dW, K, nullptr, dC, N, M, N, K, 0, 0.f, 0));
  hipEventRecord(e0, 0);
  const int reps = 5;
  for (int i = 0; i < reps; ++i) CHECK(dn_gemm_nt(dA, K, dW, K, nullptr, dC, N, M, N, K, 0, 0.f, 0));
  hipEventRecord(e1, 0); CHECK(hipDeviceSynchronize());
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
  printf("NT M=%d N=%d K=%d: %.3f ms  %.1f TFLOP/s\n", M, N, K, ms, 2.0 * M * N * K / ms * 1e-9);
  hipFree(dA); hipFree(dW); hipFree(dC);
}

static void time_tn(int M, int N, int K) {
  float *dA, *dB, *dC, *dP;
  CHECK(hipMalloc(&dA, (size_t)M * N * 4)); CHECK(hipMalloc(&dB, (size_t)M * K * 4)); CHECK(hipMalloc(&dC, (size_t)N * K * 4));
  CHECK(hipMalloc(&dP, dn_gemm_tn_workspace(M, N, K) * 4));
  CHECK(hipMemset(dA, 0, (size_t)M * N * 4)); CHECK(hipMemset(dB, 0, (size_t)M * K * 4));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  CHECK(dn_gemm_tn(dA, N, dB, K, dC, K, M, N, K, dP, 0, 0));
  hipEventRecord(e0, 0);
  const int reps = 5;
  for (int i = 0; i < reps; ++i) CHECK(dn_gemm_tn(dA, N, dB, K, dC, K, M, N, K, dP, 0, 0));
  hipEventRecord(e1, 0); CHECK(hipDeviceSynchronize());
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
  printf("TN M=%d N=%d K=%d: %.3f ms  %.1f TFLOP/s\n", M, N, K, ms, 2.0 * M * N * K / ms * 1e-9);
}

int main() {
  int bad = 0;
  bad += check_nt(300, 200, 132, 4);
  bad += check_nt(77, 64, 64, 0);
  bad += check_nt(513, 20, 36, 0);
  bad += check_nt(1, 130, 4, 0);
  bad += check_tn(5000, 90, 70);
  bad += check_tn(31, 64, 130);
  time_nt(122880, 960, 1920);
  time_nt(122880, 960, 64);
  time_nt(122880, 64, 960);
  time_nt(409600, 64, 64);
  time_nt(409600, 64, 128);
  time_tn(409600, 64, 64);
  time_tn(122880, 960, 1920);
  return bad != 0;
}
