// Issue cost of instruction kinds for ONE wave per SIMD (and 2, 4): independent streams of v_fma_f64, 32-bit VALU
// (v_mov_b32, v_cndmask_b32, v_add_u32), AGPR copies, SALU, and FP64 / 32-bit mixes -- the cost model behind the
// "instructions per knot" accounting of DESIGN.md 5.  hipcc -O3 --offload-arch=gfx950 issue_mix.hip -o issue_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(X) X X X X X X X X
__global__ void __launch_bounds__(64) k_fma64(double *out, int n, double a, double b) {
  double x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  for (int k = 0; k < n; ++k)
    asm volatile(REP8("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                      "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n")
                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
  if (x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 == 1.2345) out[threadIdx.x] = x0;
}
__global__ void __launch_bounds__(64) k_mov32(double *out, int n) {
  int x0 = threadIdx.x, x1 = 1, x2 = 2, x3 = 3, x4 = 4, x5 = 5, x6 = 6, x7 = 7;
  for (int k = 0; k < n; ++k)
    asm volatile(REP8("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0\n")
                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
  if (x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 == 12345) out[threadIdx.x] = x0;
}
__global__ void __launch_bounds__(64) k_cnd32(double *out, int n) {
  int x0 = threadIdx.x, x1 = 1, x2 = 2, x3 = 3, x4 = 4, x5 = 5, x6 = 6, x7 = 7;
  for (int k = 0; k < n; ++k)
    asm volatile(REP8("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n"
                      "v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %5, %5, %6, vcc\n v_cndmask_b32 %6, %6, %7, vcc\n v_cndmask_b32 %7, %7, %0, vcc\n")
                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : : "vcc");
  if (x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 == 12345) out[threadIdx.x] = x0;
}
__global__ void __launch_bounds__(64) k_cnd64(double *out, int n) { // e64 form, condition in an SGPR pair
  int x0 = threadIdx.x, x1 = 1, x2 = 2, x3 = 3, x4 = 4, x5 = 5, x6 = 6, x7 = 7;
  for (int k = 0; k < n; ++k)
    asm volatile(REP8("v_cndmask_b32_e64 %0, %0, %1, s[20:21]\n v_cndmask_b32_e64 %1, %1, %2, s[20:21]\n v_cndmask_b32_e64 %2, %2, %3, s[20:21]\n v_cndmask_b32_e64 %3, %3, %4, s[20:21]\n"
                      "v_cndmask_b32_e64 %4, %4, %5, s[20:21]\n v_cndmask_b32_e64 %5, %5, %6, s[20:21]\n v_cndmask_b32_e64 %6, %6, %7, s[20:21]\n v_cndmask_b32_e64 %7, %7, %0, s[20:21]\n")
                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : : "s20", "s21");
  if (x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 == 12345) out[threadIdx.x] = x0;
}
__global__ void __launch_bounds__(64) k_cndind(double *out, int n) { // independent destinations (no operand reuse between neighbours)
  int x0 = threadIdx.x, x1 = 1, x2 = 2, x3 = 3, x4 = 4, x5 = 5, x6 = 6, x7 = 7, c = 9, d = 11;
  for (int k = 0; k < n; ++k)
    asm volatile(REP8("v_cndmask_b32 %0, %8, %9, vcc\n v_cndmask_b32 %1, %8, %9, vcc\n v_cndmask_b32 %2, %8, %9, vcc\n v_cndmask_b32 %3, %8, %9, vcc\n"
                      "v_cndmask_b32 %4, %8, %9, vcc\n v_cndmask_b32 %5, %8, %9, vcc\n v_cndmask_b32 %6, %8, %9, vcc\n v_cndmask_b32 %7, %8, %9, vcc\n")
                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(c), "v"(d) : "vcc");
  if (x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 == 12345) out[threadIdx.x] = x0;
}
__global__ void __launch_bounds__(64) k_cmpmax(double *out, int n, double a) { // v_cmp_f64 -> SGPR pair, v_max_f64, v_add_f64
  double x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
  for (int k = 0; k < n; ++k)
    asm volatile(REP8("v_cmp_gt_f64 s[20:21], %0, %4\n v_max_f64 %0, %0, %4\n v_cmp_lt_f64 s[22:23], %1, %4\n v_add_f64 %1, %1, %4\n"
                      "v_cmp_gt_f64 s[24:25], %2, %4\n v_max_f64 %2, %2, %4\n v_cmp_lt_f64 s[26:27], %3, %4\n v_add_f64 %3, %3, %4\n")
                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
  if (x0 + x1 + x2 + x3 == 1.2345) out[threadIdx.x] = x0;
}
__global__ void __launch_bounds__(64) k_cndmix(double *out, int n) { // v_cndmask (vcc) alternating with v_mov_b32
  int x0 = threadIdx.x, x1 = 1, x2 = 2, x3 = 3, x4 = 4, x5 = 5, x6 = 6, x7 = 7, c = 9, d = 11;
  for (int k = 0; k < n; ++k)
    asm volatile(REP8("v_cndmask_b32 %0, %8, %9, vcc\n v_mov_b32 %1, %2\n v_cndmask_b32 %2, %8, %9, vcc\n v_mov_b32 %3, %4\n"
                      "v_cndmask_b32 %4, %8, %9, vcc\n v_mov_b32 %5, %6\n v_cndmask_b32 %6, %8, %9, vcc\n v_mov_b32 %7, %0\n")
                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(c), "v"(d) : "vcc");
  if (x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 == 12345) out[threadIdx.x] = x0;
}
__global__ void __launch_bounds__(64) k_cndvccw(double *out, int n, double a) { // compare into vcc, then two selects on it (the rsqrt / fmax idiom)
  double x0 = threadIdx.x, x1 = x0 + 1;
  int y0 = 1, y1 = 2, y2 = 3, y3 = 4;
  for (int k = 0; k < n; ++k)
    asm volatile(REP8("v_cmp_gt_f64 vcc, %0, %6\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %2, vcc\n v_add_f64 %0, %0, %6\n"
                      "v_cmp_lt_f64 vcc, %1, %6\n v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %5, %5, %4, vcc\n v_add_f64 %1, %1, %6\n")
                 : "+v"(x0), "+v"(x1), "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3) : "v"(a) : "vcc");
  if (x0 + x1 + y0 + y1 + y2 + y3 == 1.2345) out[threadIdx.x] = x0;
}
__global__ void __launch_bounds__(64) k_acc(double *out, int n) {
  int x0 = threadIdx.x, x1 = 1, x2 = 2, x3 = 3;
  for (int k = 0; k < n; ++k)
    asm volatile(REP8("v_accvgpr_write_b32 a0, %0\n v_accvgpr_write_b32 a1, %1\n v_accvgpr_write_b32 a2, %2\n v_accvgpr_write_b32 a3, %3\n"
                      "v_accvgpr_read_b32 %0, a0\n v_accvgpr_read_b32 %1, a1\n v_accvgpr_read_b32 %2, a2\n v_accvgpr_read_b32 %3, a3\n")
                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : : "a0", "a1", "a2", "a3");
  if (x0 + x1 + x2 + x3 == 12345) out[threadIdx.x] = x0;
}
__global__ void __launch_bounds__(64) k_salu(double *out, int n, int s) {
  int y = s;
  for (int k = 0; k < n; ++k)
    asm volatile(REP8("s_add_u32 %0, %0, 1\n s_and_b32 s20, %0, 3\n s_or_b32 s21, %0, 5\n s_xor_b32 s22, %0, 7\n s_add_u32 s23, s20, s21\n s_and_b32 s24, s22, 9\n s_or_b32 s25, s23, 1\n s_xor_b32 s26, s24, s25\n")
                 : "+s"(y) : : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "scc");
  if (y == 12345) out[threadIdx.x] = y;
}
__global__ void __launch_bounds__(64) k_mix(double *out, int n, double a, double b) { // FP64 and 32-bit alternating
  double x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
  int y0 = threadIdx.x, y1 = 1, y2 = 2, y3 = 3;
  for (int k = 0; k < n; ++k)
    asm volatile(REP8("v_fma_f64 %0, %0, %8, %9\n v_mov_b32 %4, %5\n v_fma_f64 %1, %1, %8, %9\n v_mov_b32 %5, %6\n"
                      "v_fma_f64 %2, %2, %8, %9\n v_mov_b32 %6, %7\n v_fma_f64 %3, %3, %8, %9\n v_mov_b32 %7, %4\n")
                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3) : "v"(a), "v"(b));
  if (x0 + x1 + x2 + x3 + y0 + y1 + y2 + y3 == 1.2345) out[threadIdx.x] = x0;
}
template <typename F>
static double time_ms(F f) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  f(); hipDeviceSynchronize();
  hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms;
}
int main() {
  double *d; hipMalloc(&d, 1 << 20);
  const int n = 4000;
  for (int blocks : {1024, 2048, 4096}) {
    const double per = blocks / 1024.0, cnt = n * 64.0 * per;
    printf("waves per SIMD %.0f: ns per wave-instruction per SIMD:", per);
    printf(" v_fma_f64 %.2f", time_ms([&] { k_fma64<<<blocks, 64>>>(d, n, 1.0000001, 1e-9); }) * 1e6 / cnt);
    printf(" | v_mov_b32 %.2f", time_ms([&] { k_mov32<<<blocks, 64>>>(d, n); }) * 1e6 / cnt);
    printf(" | v_cndmask_b32 %.2f", time_ms([&] { k_cnd32<<<blocks, 64>>>(d, n); }) * 1e6 / cnt);
    printf(" | v_cndmask e64 sgpr %.2f", time_ms([&] { k_cnd64<<<blocks, 64>>>(d, n); }) * 1e6 / cnt);
    printf(" | v_cndmask indep %.2f", time_ms([&] { k_cndind<<<blocks, 64>>>(d, n); }) * 1e6 / cnt);
    printf(" | cndmask(vcc)+mov alternating %.2f", time_ms([&] { k_cndmix<<<blocks, 64>>>(d, n); }) * 1e6 / cnt);
    printf(" | cmp->vcc, 2 cndmask, add %.2f", time_ms([&] { k_cndvccw<<<blocks, 64>>>(d, n, 0.5); }) * 1e6 / cnt);
    printf(" | cmp/max/add f64 %.2f", time_ms([&] { k_cmpmax<<<blocks, 64>>>(d, n, 0.5); }) * 1e6 / cnt);
    printf(" | accvgpr write/read %.2f", time_ms([&] { k_acc<<<blocks, 64>>>(d, n); }) * 1e6 / cnt);
    printf(" | SALU %.2f", time_ms([&] { k_salu<<<blocks, 64>>>(d, n, 3); }) * 1e6 / cnt);
    printf(" | fma64 + mov32 alternating %.2f\n", time_ms([&] { k_mix<<<blocks, 64>>>(d, n, 1.0000001, 1e-9); }) * 1e6 / cnt);
  }
  return 0;
}
