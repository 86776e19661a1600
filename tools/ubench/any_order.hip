// Does hipExtLaunchKernel(..., hipExtAnyOrderLaunch) let two kernels of ONE stream overlap on this part?  (hip_ext.h says the
// flag is not supported on GFX9xx for the module-launch variant.)  Two 200 us single-block spin kernels back to back:
// ~400 us = serialised, ~200 us = overlapped.  Also the same pair on two streams, for reference.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
__global__ void spin(long long ticks, int *out) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {}
  if (ticks < 0) *out = 1;
}
int main() {
  int *d; (void)hipMalloc(&d, 4);
  hipStream_t s, s2; (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking); (void)hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
  hipEvent_t e0, e1, ef; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); (void)hipEventCreateWithFlags(&ef, hipEventDisableTiming);
  const long long ticks = 20000; // 100 MHz -> 200 us
  for (int mode = 0; mode < 3; ++mode) {
    for (int rep = 0; rep < 3; ++rep) {
      (void)hipEventRecord(e0, s);
      hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, ticks, d);
      if (mode == 0) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, ticks, d);
      else if (mode == 1) hipExtLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, ticks, d);
      else {
        (void)hipEventRecord(ef, s); // (recorded after the first kernel: the second stream must not wait for it -> fork before)
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s2, ticks, d);
        (void)hipEventRecord(ef, s2);
        (void)hipStreamWaitEvent(s, ef, 0);
      }
      (void)hipEventRecord(e1, s);
      (void)hipEventSynchronize(e1);
      (void)hipDeviceSynchronize();
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      printf("%s: %.1f us\n", mode == 0 ? "same stream, in order" : mode == 1 ? "same stream, second launch hipExtAnyOrderLaunch" : "two streams + event join", ms * 1e3);
    }
  }
  return 0;
}
