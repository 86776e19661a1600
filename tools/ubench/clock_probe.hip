// Where do the waves of a launch land, and how fast do they run there?  Every wave runs the same independent
// v_fma_f64 stream (8 accumulators: the lone-wave issue limit, no memory) and reports its XCC / SE / CU, the shader
// clock cycles (s_memtime) and the 100 MHz wall clock (s_memrealtime) it took.  Launched on the whole chip and on
// CU-masked streams (hipExtStreamCreateWithCUMask) with 256 and 1024 waves: separates "many waves per XCD" /
// "neighbouring CUs busy" / clock frequency as causes of the backward sweep being slower at one wave per SIMD on the
// whole chip (239 us) than at one wave per CU (184 us).  hipcc -O3 --offload-arch=gfx950 clock_probe.hip -o clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <set>
#include <map>
#define REP8(X) X X X X X X X X
struct Rec { unsigned xcc, hwid; long long cyc, wall, t0; };
__global__ void __launch_bounds__(64) k_probe(Rec *out, int n_, double a, double b, int lds_mix, int quarter) {
  const int n = (quarter && (threadIdx.x & 15) >= 4) ? 0 : n_; // quarter: 3 of 4 lanes sit the loop out (EXEC-masked)
  __shared__ double sh[512];
  double x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  sh[threadIdx.x] = x0; sh[threadIdx.x + 64] = x1;
  __syncthreads();
  const long long c0 = clock64(), w0 = wall_clock64();
  for (int k = 0; k < n; ++k) {
    asm volatile(REP8("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                      "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n")
                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
    if (lds_mix) { x0 += sh[(threadIdx.x + k) & 127]; x1 += sh[(threadIdx.x + 2 * k) & 127]; }
  }
  const long long c1 = clock64(), w1 = wall_clock64();
  if (x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 == 1.2345) sh[0] = x0;
  if (threadIdx.x == 0) {
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    out[blockIdx.x] = Rec{xcc, hw, c1 - c0, w1 - w0, w0};
  }
}
static void run(const char *what, hipStream_t st, int waves, int n, int lds_mix, Rec *d, int quarter = 0, bool list = false) {
  std::vector<Rec> h(waves);
  hipLaunchKernelGGL(k_probe, dim3(waves), dim3(64), 0, st, d, 64, 1.0000001, 1e-9, lds_mix, quarter); // warm
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, st);
  hipLaunchKernelGGL(k_probe, dim3(waves), dim3(64), 0, st, d, n, 1.0000001, 1e-9, lds_mix, quarter);
  hipEventRecord(e1, st);
  hipStreamSynchronize(st);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  hipMemcpy(h.data(), d, sizeof(Rec) * waves, hipMemcpyDeviceToHost);
  std::map<unsigned, std::set<unsigned>> cus; // xcc -> {se, sh, cu}
  std::map<unsigned, int> per_cu;
  double cyc = 0, wall = 0, wmax = 0;
  for (auto &r : h) {
    const unsigned x = r.xcc & 0xf, cu = (r.hwid >> 8) & 0xf, sh = (r.hwid >> 12) & 1, se = (r.hwid >> 13) & 7;
    const unsigned id = (se << 8) | (sh << 4) | cu;
    cus[x].insert(id); per_cu[(x << 12) | id]++;
    cyc += r.cyc; wall += r.wall; if (r.wall > wmax) wmax = r.wall;
  }
  int mx = 0; for (auto &p : per_cu) if (p.second > mx) mx = p.second;
  const double ninstr = 64.0 * n + (lds_mix ? 4.0 * n : 0);
  printf("%-42s %4d waves: %zu XCCs, %zu CUs (max %d waves per CU); kernel %.1f us; per wave %.1f us (max %.1f), %.2f cycles and %.2f ns per instruction, shader clock %.0f MHz\n",
         what, waves, cus.size(), per_cu.size(), mx, ms * 1e3, wall / waves / 100.0, wmax / 100.0, cyc / waves / ninstr,
         wall / waves * 10.0 / ninstr, cyc / wall * 100.0);
  if (list) {
    printf("    CUs used per XCC (se.sh.cu):");
    for (auto &p : cus) { printf("  xcc%u:", p.first); int c = 0; for (unsigned id : p.second) { if (c++ < 40) printf(" %u.%u.%u", id >> 8, (id >> 4) & 1, id & 15); } }
    printf("\n");
  }
}
int main(int argc, char **argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 1000;
  Rec *d; hipMalloc(&d, sizeof(Rec) * 4096);
  hipStream_t s0; hipStreamCreate(&s0);
  auto masked = [&](auto pred) { uint32_t m[8] = {0}; for (int i = 0; i < 256; ++i) if (pred(i)) m[i / 32] |= 1u << (i % 32); hipStream_t s; if (hipExtStreamCreateWithCUMask(&s, 8, m) != hipSuccess) { printf("mask failed\n"); exit(1); } return s; };
  hipStream_t first64 = masked([](int i) { return i < 64; }), every4 = masked([](int i) { return i % 4 == 0; }),
              pairs = masked([](int i) { return i % 8 < 2; }), first128even = masked([](int i) { return i < 128 && i % 2 == 0; }),
              first32 = masked([](int i) { return i < 32; }), every8 = masked([](int i) { return i % 8 == 0; });
  for (int lds = 0; lds < 2; ++lds) {
    printf("---- %s ----\n", lds ? "FMA stream + 2 LDS reads per 64 FMAs" : "FMA stream only");
    run("whole chip", s0, 256, n, lds, d);
    run("whole chip", s0, 1024, n, lds, d);
    run("whole chip", s0, 2048, n, lds, d);
    run("whole chip, 16 of 64 lanes active", s0, 1024, n, lds, d, 1);
    run("whole chip, 16 of 64 lanes active", s0, 256, n, lds, d, 1);
    run("mask: first 64 bits", first64, 256, n, lds, d, 0, lds == 0);
    run("mask: every 4th bit", every4, 256, n, lds, d);
    run("mask: bits with i % 8 < 2", pairs, 256, n, lds, d);
    run("mask: even bits of the first 128", first128even, 256, n, lds, d);
    run("mask: first 32 bits", first32, 128, n, lds, d);
    run("mask: every 8th bit", every8, 128, n, lds, d);
  }
  return 0;
}
