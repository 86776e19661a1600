// 64-bit DPP row_newbcast on gfx950 (v_fmac_f64_dpp, v_mov_b64_dpp): (1) do the results equal a shuffle-based
// reference, (2) which software wait states does the hardware really need -- the DPP source written by the VALU
// instruction right before (documented: 2), and the ACCUMULATOR of a chain of dependent v_fmac_f64_dpp (an ordinary
// operand: none expected) -- (3) what the instructions cost for a lone wave and for several waves per SIMD.
// hipcc -O3 --offload-arch=gfx950 dpp_f64.hip -o dpp_f64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#define NB(c) " row_newbcast:" #c " row_mask:0xf bank_mask:0xf\n\t"
// mode 0: s_nop 1 before every DPP instruction; 1: one s_nop 1 in front of the block; 2: none at all
template <int MODE>
__global__ void __launch_bounds__(64) k_check(double *out, const double *in) {
  const int l = threadIdx.x;
  double x = in[l], h0 = in[64 + l], h1 = in[128 + l], h2 = in[192 + l], h3 = in[256 + l], acc = in[320 + l];
  // the DPP source is produced by the VALU instruction immediately before the block
  asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x) : "v"(h3));
  if (MODE == 0)
    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2" NB(0) "s_nop 1\n\tv_fmac_f64_dpp %0, %1, %3" NB(1)
                 "s_nop 1\n\tv_fmac_f64_dpp %0, -%1, %4" NB(2) "s_nop 1\n\tv_fmac_f64_dpp %0, %1, %5" NB(3)
                 : "+v"(acc) : "v"(x), "v"(h0), "v"(h1), "v"(h2), "v"(h3));
  else if (MODE == 1)
    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2" NB(0) "v_fmac_f64_dpp %0, %1, %3" NB(1)
                 "v_fmac_f64_dpp %0, -%1, %4" NB(2) "v_fmac_f64_dpp %0, %1, %5" NB(3)
                 : "+v"(acc) : "v"(x), "v"(h0), "v"(h1), "v"(h2), "v"(h3));
  else
    asm volatile("v_fmac_f64_dpp %0, %1, %2" NB(0) "v_fmac_f64_dpp %0, %1, %3" NB(1)
                 "v_fmac_f64_dpp %0, -%1, %4" NB(2) "v_fmac_f64_dpp %0, %1, %5" NB(3)
                 : "+v"(acc) : "v"(x), "v"(h0), "v"(h1), "v"(h2), "v"(h3));
  // a broadcast of the accumulator right after the chain (mov), then a second-level use
  double b;
  if (MODE == 2) asm volatile("v_mov_b64_dpp %0, %1" NB(5) : "=v"(b) : "v"(acc));
  else asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1" NB(5) : "=v"(b) : "v"(acc));
  out[l] = acc;
  out[64 + l] = b;
}

// more producers right in front of a DPP read, no wait states: a 32-bit VALU write of half of the source
// (v_cndmask_b32, what a select of 0.0 / 1.0 compiles to), a 64-bit move, and a transcendental (v_rsq_f64)
__global__ void __launch_bounds__(64) k_check2(double *out, const double *in) {
  const int l = threadIdx.x;
  double x = in[l], y = in[64 + l], b0, b1, b2, t2, t3, t4;
  // (a) 32-bit write of the high word of the source, DPP read of the pair at once
  asm volatile("v_mov_b64 v[100:101], %2\n\ts_nop 7\n\tv_xor_b32 v101, v101, %3\n\tv_mov_b64_dpp %1, v[100:101]" NB(7) "v_mov_b64 %0, v[100:101]"
               : "=&v"(t2), "=&v"(b0) : "v"(y), "v"(l << 20) : "v100", "v101");
  // (b) 64-bit move, DPP read at once
  asm volatile("v_mov_b64 %0, %2\n\tv_mov_b64_dpp %1, %0" NB(9) : "=&v"(t3), "=&v"(b1) : "v"(y));
  // (c) v_rsq_f64, DPP read at once
  asm volatile("v_rsq_f64 %0, %2\n\tv_mov_b64_dpp %1, %0" NB(3) : "=&v"(t4), "=&v"(b2) : "v"(x));
  out[l] = t2; out[64 + l] = b0; out[128 + l] = t3; out[192 + l] = b1; out[256 + l] = t4; out[320 + l] = b2;
}
// issue cost: a stream of independent / dependent v_fmac_f64_dpp and of plain v_fma_f64 for comparison
#define REP8(X) X X X X X X X X
__global__ void __launch_bounds__(64) k_tput(double *out, int n, int kind) {
  double x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, a = 1e-9, b = 1.0 + 1e-12;
  if (kind == 0) for (int k = 0; k < n; ++k) // independent plain FMAs
    asm volatile(REP8("v_fma_f64 %0, %0, %7, %6\n v_fma_f64 %1, %1, %7, %6\n v_fma_f64 %2, %2, %7, %6\n v_fma_f64 %3, %3, %7, %6\n v_fma_f64 %4, %4, %7, %6\n v_fma_f64 %5, %5, %7, %6\n")
                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5) : "v"(a), "v"(b));
  else if (kind == 1) for (int k = 0; k < n; ++k) // independent DPP FMAs (source a, never rewritten)
    asm volatile(REP8("v_fmac_f64_dpp %0, %6, %7" NB(0) "v_fmac_f64_dpp %1, %6, %7" NB(1) "v_fmac_f64_dpp %2, %6, %7" NB(2)
                      "v_fmac_f64_dpp %3, %6, %7" NB(3) "v_fmac_f64_dpp %4, %6, %7" NB(0) "v_fmac_f64_dpp %5, %6, %7" NB(1))
                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5) : "v"(a), "v"(b));
  else if (kind == 2) for (int k = 0; k < n; ++k) // one dependent chain of DPP FMAs (accumulator)
    asm volatile(REP8("v_fmac_f64_dpp %0, %6, %7" NB(0) "v_fmac_f64_dpp %0, %6, %7" NB(1) "v_fmac_f64_dpp %0, %6, %7" NB(2)
                      "v_fmac_f64_dpp %0, %6, %7" NB(3) "v_fmac_f64_dpp %0, %6, %7" NB(0) "v_fmac_f64_dpp %0, %6, %7" NB(1))
                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5) : "v"(a), "v"(b));
  else if (kind == 3) for (int k = 0; k < n; ++k) // dependent chain of plain FMAs
    asm volatile(REP8("v_fma_f64 %0, %0, %7, %6\n v_fma_f64 %0, %0, %7, %6\n v_fma_f64 %0, %0, %7, %6\n v_fma_f64 %0, %0, %7, %6\n v_fma_f64 %0, %0, %7, %6\n v_fma_f64 %0, %0, %7, %6\n")
                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5) : "v"(a), "v"(b));
  else if (kind == 4) for (int k = 0; k < n; ++k) // chain through the DPP SOURCE: mov_dpp of the value just produced (with the nop)
    asm volatile(REP8("s_nop 1\n\tv_mov_b64_dpp %0, %0" NB(1) "s_nop 1\n\tv_mov_b64_dpp %0, %0" NB(2) "s_nop 1\n\tv_mov_b64_dpp %0, %0" NB(3)
                      "s_nop 1\n\tv_mov_b64_dpp %0, %0" NB(0) "s_nop 1\n\tv_mov_b64_dpp %0, %0" NB(1) "s_nop 1\n\tv_mov_b64_dpp %0, %0" NB(2))
                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5) : "v"(a), "v"(b));
  else for (int k = 0; k < n; ++k) // independent DPP FMAs each behind its own s_nop 1
    asm volatile(REP8("s_nop 1\n\tv_fmac_f64_dpp %0, %6, %7" NB(0) "s_nop 1\n\tv_fmac_f64_dpp %1, %6, %7" NB(1) "s_nop 1\n\tv_fmac_f64_dpp %2, %6, %7" NB(2)
                      "s_nop 1\n\tv_fmac_f64_dpp %3, %6, %7" NB(3) "s_nop 1\n\tv_fmac_f64_dpp %4, %6, %7" NB(0) "s_nop 1\n\tv_fmac_f64_dpp %5, %6, %7" NB(1))
                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5) : "v"(a), "v"(b));
  if (x0 + x1 + x2 + x3 + x4 + x5 == 1.2345) out[threadIdx.x] = x0;
}
int main() {
  double h[384], o[3][128], *din, *dout;
  for (int i = 0; i < 384; ++i) h[i] = sin(0.37 * i + 0.11) + 1.5;
  hipMalloc(&din, sizeof h); hipMalloc(&dout, 128 * 8);
  hipMemcpy(din, h, sizeof h, hipMemcpyHostToDevice);
  for (int m = 0; m < 3; ++m) {
    if (m == 0) k_check<0><<<1, 64>>>(dout, din); else if (m == 1) k_check<1><<<1, 64>>>(dout, din); else k_check<2><<<1, 64>>>(dout, din);
    hipMemcpy(o[m], dout, 128 * 8, hipMemcpyDeviceToHost);
  }
  int bad[3] = {0, 0, 0}, badb[3] = {0, 0, 0};
  for (int l = 0; l < 64; ++l) {
    const int r = l & ~15;
    auto X = [&](int k) { return h[k] * h[256 + k]; };
    double acc = h[320 + l];
    acc = fma(X(r + 0), h[64 + l], acc); acc = fma(X(r + 1), h[128 + l], acc);
    acc = fma(-X(r + 2), h[192 + l], acc); acc = fma(X(r + 3), h[256 + l], acc);
    for (int m = 0; m < 3; ++m) bad[m] += (o[m][l] != acc);
  }
  for (int m = 0; m < 3; ++m) for (int l = 0; l < 64; ++l) badb[m] += (o[m][64 + l] != o[m][(l & ~15) + 5]);
  printf("v_fmac_f64_dpp chain vs host fma reference, lanes that differ: nop-every %d, nop-first %d, no-nop %d\n", bad[0], bad[1], bad[2]);
  printf("v_mov_b64_dpp of the fresh accumulator, lanes that differ: %d %d %d (no-nop variant: hazard exposed if > 0)\n", badb[0], badb[1], badb[2]);
  {
    double *d2, o2[384]; hipMalloc(&d2, 384 * 8);
    k_check2<<<1, 64>>>(d2, din); hipMemcpy(o2, d2, 384 * 8, hipMemcpyDeviceToHost);
    int b[3] = {0, 0, 0};
    for (int l = 0; l < 64; ++l) { const int r = l & ~15; b[0] += (o2[64 + l] != o2[r + 7]); b[1] += (o2[192 + l] != o2[128 + r + 9]); b[2] += (o2[320 + l] != o2[256 + r + 3]); }
    printf("DPP read right after its producer, no wait states, lanes that differ: 32-bit half write %d, v_mov_b64 %d, v_rsq_f64 %d\n", b[0], b[1], b[2]);
  }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char *names[] = {"independent v_fma_f64", "independent v_fmac_f64_dpp", "dependent v_fmac_f64_dpp chain", "dependent v_fma_f64 chain", "dependent s_nop 1 + v_mov_b64_dpp chain", "independent s_nop 1 + v_fmac_f64_dpp"};
  const int n = 2000;
  for (int wps = 1; wps <= 4; wps *= 2)
    for (int kind = 0; kind < 6; ++kind) {
      k_tput<<<256 * 4 * wps, 64>>>(dout, 10, kind); hipDeviceSynchronize();
      hipEventRecord(e0); k_tput<<<256 * 4 * wps, 64>>>(dout, n, kind); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("%d wave(s) per SIMD  %-42s %.2f ns per instruction slot per SIMD\n", wps, names[kind], ms * 1e6 / (n * 48.0 * wps));
    }
  return 0;
}
