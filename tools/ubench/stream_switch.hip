// Why does the first kernel of the next iteration start ~75 us after select_kernel has ended?  Is it what the
// runtime does when the launching thread moves to another stream between two launches on this one?  Stream A: W (writes `mb` MB,
// leaving the caches dirty), S (tiny); then, optionally, a tiny launch on stream B; then N (tiny) on A.  Reported: S end -> N
// start, with and without the launch on B in between, with small and large W.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void writer(double *p, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (double)i; }
__global__ void stamp(long long *t) { if (threadIdx.x == 0) { t[0] = wall_clock64(); for (int i = 0; i < 200; ++i) __builtin_amdgcn_s_sleep(10); t[1] = wall_clock64(); } }
int main() {
  hipStream_t a, b; (void)hipStreamCreateWithFlags(&a, hipStreamNonBlocking); (void)hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
  double *buf; (void)hipMalloc(&buf, (size_t)1 << 30);
  long long *t; (void)hipMalloc(&t, 8 * 8);
  for (int mb : {1, 400}) for (int sw = 0; sw < 3; ++sw) for (int rep = 0; rep < 3; ++rep) {
    (void)hipMemset(t, 0, 64);
    (void)hipDeviceSynchronize();
    hipLaunchKernelGGL(writer, dim3(4096), dim3(256), 0, a, buf, (size_t)mb * 131072);
    hipLaunchKernelGGL(stamp, dim3(1), dim3(64), 0, a, t);          // "select"
    if (sw == 1) hipLaunchKernelGGL(stamp, dim3(1), dim3(64), 0, b, t + 4);  // the thread launches on another stream
    if (sw == 2) { for (int k = 0; k < 7; ++k) hipLaunchKernelGGL(stamp, dim3(1), dim3(64), 0, b, t + 4); }
    hipLaunchKernelGGL(stamp, dim3(1), dim3(64), 0, a, t + 2);      // next kernel on A
    (void)hipDeviceSynchronize();
    long long h[8]; (void)hipMemcpy(h, t, 64, hipMemcpyDeviceToHost);
    if (rep == 2) printf("W writes %3d MB, %s: S end -> N start %.1f us\n", mb, sw == 0 ? "no launch elsewhere in between" : sw == 1 ? "one launch on another stream in between" : "seven launches on another stream in between", (h[2] - h[1]) / 100.0);
  }
  return 0;
}
