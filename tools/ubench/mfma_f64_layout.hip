// Lane maps and accumulation order of v_mfma_f64_16x16x4_f64 on gfx950, checked against a host fma chain.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
__global__ void k(const double *A, const double *B, const double *C, double *D) {
  const int l = threadIdx.x, li = l & 15, lk = l >> 4;
  double4_t c;
  for (int r = 0; r < 4; ++r) c[r] = C[(lk + 4 * r) * 16 + li];
  // A[16][8], B[8][16]: two k-steps
  for (int ks = 0; ks < 2; ++ks) c = __builtin_amdgcn_mfma_f64_16x16x4f64(A[li * 8 + 4 * ks + lk], B[(4 * ks + lk) * 16 + li], c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[(lk + 4 * r) * 16 + li] = c[r];
}
int main() {
  double hA[128], hB[128], hC[256], hD[256];
  unsigned s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (double)(s >> 8) / (1 << 24) - 0.5; };
  for (double &v : hA) v = rnd() * 3.0;
  for (double &v : hB) v = rnd() * 7.0;
  for (double &v : hC) v = rnd() * 1e-3;
  double *dA, *dB, *dC, *dD;
  hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dC, sizeof hC); hipMalloc(&dD, sizeof hD);
  hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
  hipMemcpy(dC, hC, sizeof hC, hipMemcpyHostToDevice);
  k<<<1, 64>>>(dA, dB, dC, dD);
  hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
  int exact = 0, close = 0;
  double worst = 0;
  for (int i = 0; i < 16; ++i)
    for (int j = 0; j < 16; ++j) {
      double acc = hC[i * 16 + j];
      for (int kk = 0; kk < 8; ++kk) acc = fma(hA[i * 8 + kk], hB[kk * 16 + j], acc);
      const double d = fabs(acc - hD[i * 16 + j]);
      exact += d == 0.0; close += d < 1e-12; worst = fmax(worst, d);
    }
  printf("mfma_f64_16x16x4: %d / 256 entries bit-equal to the k-ordered fma chain, %d within 1e-12, worst %.3e\n", exact, close, worst);
  return 0;
}
