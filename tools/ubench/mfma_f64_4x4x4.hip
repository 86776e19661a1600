// v_mfma_f64_4x4x4_4b_f64 on gfx950: FOUR independent 4x4x4 blocks per instruction, one per 16-lane group -- the shape of
// one 16-lane team of the nx = 8 backward sweep.  (1) lane maps: which lane supplies A[i][k], B[k][j] and receives
// D[i][j] of its block -- found by probing with one-hot operands; (2) does a two-instruction chain accumulate like the
// k-ordered fma chain (bit-equal); (3) issue cost for a lone wave and for 2 / 4 waves per SIMD, dependent and not.
// hipcc -O3 --offload-arch=gfx950 mfma_f64_4x4x4.hip -o mfma_f64_4x4x4
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
__global__ void k_probe(const double *a, const double *b, const double *c, double *d) {
  const int l = threadIdx.x;
  d[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], c[l], 0, 0, 0);
}
__global__ void k_chain(const double *a0, const double *b0, const double *a1, const double *b1, const double *c, double *d) {
  const int l = threadIdx.x;
  double acc = __builtin_amdgcn_mfma_f64_4x4x4f64(a0[l], b0[l], c[l], 0, 0, 0);
  d[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a1[l], b1[l], acc, 0, 0, 0);
}
#define REP8(X) X X X X X X X X
__global__ void __launch_bounds__(64) k_tput(double *out, int n, int kind) {
  double x0 = threadIdx.x * 1e-3, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, a = 1e-9, b = 1.0 + 1e-12;
  if (kind == 0) for (int k = 0; k < n; ++k) // 4 independent accumulators
    asm volatile(REP8("v_mfma_f64_4x4x4_4b_f64 %0, %4, %5, %0\n v_mfma_f64_4x4x4_4b_f64 %1, %4, %5, %1\n v_mfma_f64_4x4x4_4b_f64 %2, %4, %5, %2\n v_mfma_f64_4x4x4_4b_f64 %3, %4, %5, %3\n")
                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a), "v"(b));
  else for (int k = 0; k < n; ++k) // one dependent chain
    asm volatile(REP8("v_mfma_f64_4x4x4_4b_f64 %0, %4, %5, %0\n v_mfma_f64_4x4x4_4b_f64 %0, %4, %5, %0\n v_mfma_f64_4x4x4_4b_f64 %0, %4, %5, %0\n v_mfma_f64_4x4x4_4b_f64 %0, %4, %5, %0\n")
                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a), "v"(b));
  if (x0 + x1 + x2 + x3 == 1.2345) out[threadIdx.x] = x0;
}
int main() {
  double hA[64], hB[64], hC[64], hD[64], *dA, *dB, *dC, *dD, *dA1, *dB1;
  hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dC, 512); hipMalloc(&dD, 512); hipMalloc(&dA1, 512); hipMalloc(&dB1, 512);
  // ---- lane maps: A one-hot in lane la, B one-hot in lane lb of block 0 -> which D lane lights up ----
  int amap_i[16], amap_k[16], bmap_k[16], bmap_j[16], dlane[4][4];
  for (int i = 0; i < 16; ++i) amap_i[i] = amap_k[i] = bmap_k[i] = bmap_j[i] = -1;
  int hits[16][16];
  for (int la = 0; la < 16; ++la)
    for (int lb = 0; lb < 16; ++lb) {
      for (int l = 0; l < 64; ++l) { hA[l] = 0; hB[l] = 0; hC[l] = 0; }
      hA[la] = 2.0; hB[lb] = 3.0;
      hipMemcpy(dA, hA, 512, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 512, hipMemcpyHostToDevice); hipMemcpy(dC, hC, 512, hipMemcpyHostToDevice);
      k_probe<<<1, 64>>>(dA, dB, dC, dD);
      hipMemcpy(hD, dD, 512, hipMemcpyDeviceToHost);
      hits[la][lb] = -1;
      int n = 0;
      for (int l = 0; l < 64; ++l) if (hD[l] != 0.0) { hits[la][lb] = l; ++n; }
      if (n > 1) hits[la][lb] = -2;
    }
  printf("D lane that receives A(lane la) * B(lane lb), block 0 (-1: none -- the two lanes hold different k):\n      lb:");
  for (int lb = 0; lb < 16; ++lb) printf("%3d", lb);
  printf("\n");
  for (int la = 0; la < 16; ++la) {
    printf("  la %2d:  ", la);
    for (int lb = 0; lb < 16; ++lb) printf("%3d", hits[la][lb]);
    printf("\n");
  }
  // hypothesis printed next to it: A lane = i + 4 k, B lane = j + 4 k, D lane = i + 4 j?  (checked below by a full product)
  unsigned s = 777;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (double)(s >> 8) / (1 << 24) - 0.5; };
  double A0[4][4][4], B0[4][4][4], A1[4][4][4], B1[4][4][4], C0[4][4][4];
  for (int b = 0; b < 4; ++b) for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) {
    A0[b][i][j] = rnd() * 3; B0[b][i][j] = rnd() * 5; A1[b][i][j] = rnd() * 7; B1[b][i][j] = rnd(); C0[b][i][j] = rnd() * 1e-3; }
  for (int hyp = 0; hyp < 4; ++hyp) {
    // hyp bit 0: A lane = i + 4k (0) or k + 4i (1); bit 1: B lane = j + 4k (0) or k + 4j (1); D lane = i + 4j tried first, then j + 4i
    for (int dm = 0; dm < 2; ++dm) {
      double a0[64], b0[64], a1[64], b1[64], c[64];
      for (int b = 0; b < 4; ++b) for (int p = 0; p < 4; ++p) for (int q = 0; q < 4; ++q) {
        const int la = 16 * b + ((hyp & 1) ? q + 4 * p : p + 4 * q);   // A[p][q]: p = i, q = k
        const int lb = 16 * b + ((hyp & 2) ? p + 4 * q : q + 4 * p);   // B[p][q]: p = k, q = j
        const int ld = 16 * b + (dm ? q + 4 * p : p + 4 * q);          // C/D[p][q]
        a0[la] = A0[b][p][q]; a1[la] = A1[b][p][q]; b0[lb] = B0[b][p][q]; b1[lb] = B1[b][p][q]; c[ld] = C0[b][p][q];
      }
      hipMemcpy(dA, a0, 512, hipMemcpyHostToDevice); hipMemcpy(dB, b0, 512, hipMemcpyHostToDevice); hipMemcpy(dA1, a1, 512, hipMemcpyHostToDevice);
      hipMemcpy(dB1, b1, 512, hipMemcpyHostToDevice); hipMemcpy(dC, c, 512, hipMemcpyHostToDevice);
      k_chain<<<1, 64>>>(dA, dB, dA1, dB1, dC, dD);
      hipMemcpy(hD, dD, 512, hipMemcpyDeviceToHost);
      int exact = 0; double worst = 0;
      for (int b = 0; b < 4; ++b) for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) {
        double acc = C0[b][i][j];
        for (int kk = 0; kk < 4; ++kk) acc = fma(A0[b][i][kk], B0[b][kk][j], acc);
        for (int kk = 0; kk < 4; ++kk) acc = fma(A1[b][i][kk], B1[b][kk][j], acc);
        const double dv = hD[16 * b + (dm ? j + 4 * i : i + 4 * j)];
        exact += (dv == acc); worst = fmax(worst, fabs(dv - acc));
      }
      printf("layout hypothesis A lane = %s, B lane = %s, D lane = %s: %d / 64 entries bit-equal to the k-ordered fma chain of two instructions (worst %.2e)\n",
             (hyp & 1) ? "k + 4 i" : "i + 4 k", (hyp & 2) ? "k + 4 j" : "j + 4 k", dm ? "j + 4 i" : "i + 4 j", exact, worst);
    }
  }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int n = 1000;
  for (int wps = 1; wps <= 4; wps *= 2)
    for (int kind = 0; kind < 2; ++kind) {
      k_tput<<<256 * 4 * wps, 64>>>(dD, 10, kind); hipDeviceSynchronize();
      hipEventRecord(e0); k_tput<<<256 * 4 * wps, 64>>>(dD, n, kind); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("%d wave(s) per SIMD, %s v_mfma_f64_4x4x4_4b_f64: %.2f ns per instruction per SIMD (256 FMAs each)\n", wps, kind ? "one dependent chain of" : "4 independent", ms * 1e6 / (n * 32.0 * wps));
    }
  return 0;
}
