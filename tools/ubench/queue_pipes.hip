// Does a launch with many pending workgroups on one stream delay the START of a small kernel on another stream?  (The one gap in
// a sub-shard's chain is select -> calc, and calc starts when another sub-shard's trial-cost launch -- 16 160 short blocks --
// ends: DESIGN.md 5.R3 (o).)  Stream A: `big` blocks of 64 threads that each spin ~20 us, far more than fit at once.
// Stream B (one of NS other streams), 50 us later: ONE block that records its start time.  Reported: when B's block started
// relative to A's first and last block.  hipcc -O3 --offload-arch=gfx950 queue_pipes.hip -o queue_pipes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ void __launch_bounds__(64) spin_many(long long ticks, long long *first, long long *last, int regs_hog) {
  const long long t0 = wall_clock64();
  if (threadIdx.x == 0) { atomicMin((unsigned long long *)first, (unsigned long long)t0); }
  while (wall_clock64() - t0 < ticks) {}
  if (threadIdx.x == 0) atomicMax((unsigned long long *)last, (unsigned long long)wall_clock64());
}
__global__ void __launch_bounds__(64) probe(long long *start) { if (threadIdx.x == 0) *start = wall_clock64(); }
__global__ void __launch_bounds__(64) delay(long long ticks) { const long long t0 = wall_clock64(); while (wall_clock64() - t0 < ticks) {} }
int main(int argc, char **argv) {
  const int NS = 7, big = argc > 1 ? atoi(argv[1]) : 60000;
  hipStream_t a, b[NS];
  (void)hipStreamCreateWithFlags(&a, hipStreamNonBlocking);
  for (int i = 0; i < NS; ++i) (void)hipStreamCreateWithFlags(&b[i], hipStreamNonBlocking);
  long long *d; (void)hipMalloc(&d, 8 * 4);
  for (int i = 0; i < NS; ++i) {
    for (int rep = 0; rep < 2; ++rep) {
      long long h[3] = {0x7fffffffffffffffLL, 0, 0};
      (void)hipMemcpy(d, h, 24, hipMemcpyHostToDevice);
      // warm both streams
      hipLaunchKernelGGL(delay, dim3(1), dim3(64), 0, a, 100);
      hipLaunchKernelGGL(delay, dim3(1), dim3(64), 0, b[i], 100);
      (void)hipDeviceSynchronize();
      hipLaunchKernelGGL(delay, dim3(1), dim3(64), 0, b[i], 5000);            // B waits ~50 us ...
      hipLaunchKernelGGL(spin_many, dim3(big), dim3(64), 0, a, 2000, d, d + 1, 0); // ... while A's big launch gets going
      hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, b[i], d + 2);
      (void)hipDeviceSynchronize();
      (void)hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
      if (rep == 1)
        printf("other stream %d: A's blocks ran from 0 to %.1f us; the probe on the other stream started at %.1f us\n", i,
               (h[1] - h[0]) / 100.0, (h[2] - h[0]) / 100.0);
    }
  }
  return 0;
}
