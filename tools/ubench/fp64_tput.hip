// Wall-clock FP64 issue throughput per SIMD as a function of waves per SIMD and ILP, plus LDS read
// throughput (ds_read_b64 / b128) and cross-lane (ds_bpermute / DPP) latency.  One wave per block.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int ILP>
__global__ void __launch_bounds__(64) fma_chain(double *out, int n, double a, double b) {
  double x[ILP];
  for (int i = 0; i < ILP; ++i) x[i] = threadIdx.x + i;
  for (int k = 0; k < n; ++k) {
#pragma unroll
    for (int i = 0; i < ILP; ++i) x[i] = fma(x[i], a, b);
  }
  double s = 0;
  for (int i = 0; i < ILP; ++i) s += x[i];
  if (s == 12345.678) out[blockIdx.x * 64 + threadIdx.x] = s;
}
// LDS broadcast-read throughput: every lane reads NR doubles (b128 pairs) per iteration and folds them into FMAs
template <int W>
__global__ void __launch_bounds__(64) lds_read(double *out, int n) {
  __shared__ double sm[512];
  for (int i = threadIdx.x; i < 512; i += 64) sm[i] = i * 1e-3;
  __syncthreads();
  double acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0;
  const int base = (threadIdx.x & 7) * 8;
  for (int k = 0; k < n; ++k) {
    if (W == 2) {
      const double2 *p = reinterpret_cast<const double2 *>(sm + base + ((k & 3) << 6));
#pragma unroll
      for (int i = 0; i < 4; ++i) { double2 v = p[i]; acc0 += v.x; acc1 += v.y; }
    } else {
      const double *p = sm + base + ((k & 3) << 6);
#pragma unroll
      for (int i = 0; i < 8; ++i) { acc0 += p[i]; }
    }
  }
  double s = acc0 + acc1 + acc2 + acc3;
  if (s == 12345.678) out[blockIdx.x * 64 + threadIdx.x] = s;
}
// dependent chain through ds_bpermute (two 32-bit halves of a double) + add
__global__ void __launch_bounds__(64) bperm_chain(double *out, int n) {
  double x = threadIdx.x;
  const int src = ((threadIdx.x + 8) & 63) << 2;
  long long t0 = clock64();
  for (int k = 0; k < n; ++k) {
    int lo = __builtin_amdgcn_ds_bpermute(src, __double2loint(x));
    int hi = __builtin_amdgcn_ds_bpermute(src, __double2hiint(x));
    x = __hiloint2double(hi, lo) + 1.0;
  }
  long long t1 = clock64();
  if (x == 12345.678) out[threadIdx.x] = x;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[1 << 20] = (double)(t1 - t0) / n;
}
// dependent chain through DPP row_shr (within 16 lanes) + add
__global__ void __launch_bounds__(64) dpp_chain(double *out, int n) {
  double x = threadIdx.x;
  long long t0 = clock64();
  for (int k = 0; k < n; ++k) {
    int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), 0x118, 0xf, 0xf, false); // row_shr:8
    int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), 0x118, 0xf, 0xf, false);
    x = __hiloint2double(hi, lo) + 1.0;
  }
  long long t1 = clock64();
  if (x == 12345.678) out[threadIdx.x] = x;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[1 << 20] = (double)(t1 - t0) / n;
}
__global__ void __launch_bounds__(64) clock_rate(double *out, int n) {
  long long t0 = clock64(), w0 = wall_clock64();
  double x = threadIdx.x;
  for (int k = 0; k < n; ++k) x = fma(x, 1.0000001, 1e-9);
  long long t1 = clock64(), w1 = wall_clock64();
  if (x == 12345.678) out[threadIdx.x] = x;
  if (threadIdx.x == 0) { out[0] = (double)(t1 - t0); out[1] = (double)(w1 - w0); }
}
template <typename F>
static double time_ms(F f) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  f(); hipDeviceSynchronize();
  hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms;
}
int main() {
  double *d; hipMalloc(&d, ((1 << 20) + 8) * sizeof(double));
  const int n = 20000;
  double h[2];
  clock_rate<<<1, 64>>>(d, n); hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
  int wc = 0; hipDeviceGetAttribute(&wc, hipDeviceAttributeWallClockRate, 0);
  int sc = 0; hipDeviceGetAttribute(&sc, hipDeviceAttributeClockRate, 0);
  printf("clock64 ticks %.0f, wall_clock64 ticks %.0f (wall clock rate %d kHz, shader clock attr %d kHz) -> clock64 = %.1f MHz\n",
         h[0], h[1], wc, sc, h[0] / h[1] * wc * 1e-3);
  for (int blocks : {1024, 2048, 4096, 8192, 16384}) {
    printf("waves=%5d (%.0f per SIMD): ns per wave-instruction per SIMD:", blocks, blocks / 1024.0);
    double ms;
    const double per_simd = blocks / 1024.0;
    ms = time_ms([&] { fma_chain<1><<<blocks, 64>>>(d, n, 1.0000001, 1e-9); }); printf(" ILP1 %.2f", ms * 1e6 / (n * 1.0 * per_simd));
    ms = time_ms([&] { fma_chain<2><<<blocks, 64>>>(d, n, 1.0000001, 1e-9); }); printf(" ILP2 %.2f", ms * 1e6 / (n * 2.0 * per_simd));
    ms = time_ms([&] { fma_chain<4><<<blocks, 64>>>(d, n, 1.0000001, 1e-9); }); printf(" ILP4 %.2f", ms * 1e6 / (n * 4.0 * per_simd));
    ms = time_ms([&] { fma_chain<8><<<blocks, 64>>>(d, n, 1.0000001, 1e-9); }); printf(" ILP8 %.2f", ms * 1e6 / (n * 8.0 * per_simd));
    ms = time_ms([&] { lds_read<1><<<blocks, 64>>>(d, n); }); printf(" | LDS b64+add per read %.2f", ms * 1e6 / (n * 8.0 * per_simd));
    ms = time_ms([&] { lds_read<2><<<blocks, 64>>>(d, n); }); printf(" b128+2add per read %.2f\n", ms * 1e6 / (n * 4.0 * per_simd));
  }
  bperm_chain<<<1, 64>>>(d, n); hipMemcpy(h, d + (1 << 20), 8, hipMemcpyDeviceToHost); printf("bpermute(2x)+add chain: %.1f ticks\n", h[0]);
  dpp_chain<<<1, 64>>>(d, n); hipMemcpy(h, d + (1 << 20), 8, hipMemcpyDeviceToHost); printf("dpp(2x)+add chain: %.1f ticks\n", h[0]);
  return 0;
}
