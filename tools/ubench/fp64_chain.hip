// One wave per block: cycles per v_fma_f64 for a dependent chain and for ILP 2/4/8, and for LDS round trips.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int ILP>
__global__ void fma_chain(double *out, int n, double a, double b) {
  double x[ILP];
  for (int i = 0; i < ILP; ++i) x[i] = threadIdx.x + i;
  long long t0 = clock64();
  for (int k = 0; k < n; ++k) {
#pragma unroll
    for (int i = 0; i < ILP; ++i) x[i] = fma(x[i], a, b);
  }
  long long t1 = clock64();
  double s = 0;
  for (int i = 0; i < ILP; ++i) s += x[i];
  out[blockIdx.x * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[1 << 20] = (double)(t1 - t0) / ((double)n * ILP);
}
__global__ void lds_roundtrip(double *out, int n) {
  __shared__ double sm[64];
  double x = threadIdx.x;
  long long t0 = clock64();
  for (int k = 0; k < n; ++k) {
    sm[threadIdx.x] = x;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    x = sm[(threadIdx.x + 1) & 63] + 1.0;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  long long t1 = clock64();
  out[blockIdx.x * 64 + threadIdx.x] = x;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[1 << 20] = (double)(t1 - t0) / n;
}
__global__ void rsqrt_chain(double *out, int n) {
  double x = 1.5 + threadIdx.x;
  long long t0 = clock64();
  for (int k = 0; k < n; ++k) x = rsqrt(x) + 1.0;
  long long t1 = clock64();
  out[threadIdx.x] = x;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[1 << 20] = (double)(t1 - t0) / n;
}
int main() {
  double *d; hipMalloc(&d, ((1 << 20) + 8) * sizeof(double));
  double h; const int n = 20000;
  for (int blocks : {1, 1024, 4096}) {
    printf("blocks=%d (waves): ", blocks);
    fma_chain<1><<<blocks, 64>>>(d, n, 1.0000001, 1e-9); hipMemcpy(&h, d + (1 << 20), 8, hipMemcpyDeviceToHost); printf("fma ILP1 %.2f  ", h);
    fma_chain<2><<<blocks, 64>>>(d, n, 1.0000001, 1e-9); hipMemcpy(&h, d + (1 << 20), 8, hipMemcpyDeviceToHost); printf("ILP2 %.2f  ", h);
    fma_chain<4><<<blocks, 64>>>(d, n, 1.0000001, 1e-9); hipMemcpy(&h, d + (1 << 20), 8, hipMemcpyDeviceToHost); printf("ILP4 %.2f  ", h);
    fma_chain<8><<<blocks, 64>>>(d, n, 1.0000001, 1e-9); hipMemcpy(&h, d + (1 << 20), 8, hipMemcpyDeviceToHost); printf("ILP8 %.2f  ", h);
    lds_roundtrip<<<blocks, 64>>>(d, n); hipMemcpy(&h, d + (1 << 20), 8, hipMemcpyDeviceToHost); printf("LDS write->read->add %.1f  ", h);
    rsqrt_chain<<<blocks, 64>>>(d, n); hipMemcpy(&h, d + (1 << 20), 8, hipMemcpyDeviceToHost); printf("rsqrt+add %.1f (clock64 ticks per op)\n", h);
  }
  return 0;
}
