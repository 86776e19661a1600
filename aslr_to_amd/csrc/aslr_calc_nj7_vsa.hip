// nj = 7 with VSA actuation (nx = 28, nu = 14): the model-level kernels (calc / calcDiff sweeps, dam_eval, dam_residuals)
// in a translation unit of their own, so that they compile next to aslr_calc_nj7.hip instead of after it
#include "aslr_calc.inc.hpp"

namespace aslr {

int launch_calc_nj7_vsa(const KArgs &k, bool diff, int mode, double th_gaptol, hipStream_t st) {
  dim3 grid((k.b1 - k.b0 + 63) / 64, k.T + 1), block(64);
  if (diff) hipLaunchKernelGGL((calc_kernel<7, ASLR_DAM_VSA, true, false>), grid, block, 0, st, k, mode, th_gaptol);
  else hipLaunchKernelGGL((calc_kernel<7, ASLR_DAM_VSA, false, false>), grid, block, 0, st, k, mode, th_gaptol);
  HIP_TRY(hipGetLastError());
  return ASLR_OK;
}

int launch_dam_eval_nj7_vsa(const KArgs &k, int mi, int n, const double *x, const double *u, double *xout, double *cost,
                            double *Fx, double *Fu, double *Lx, double *Lu, double *Lxx, double *Lxu, double *Luu,
                            hipStream_t st) {
  dim3 grid((n + 63) / 64), block(64);
  hipLaunchKernelGGL((dam_eval_kernel<7, ASLR_DAM_VSA, false>), grid, block, 0, st, k.desc, mi, k.frame_ref, n, x, u, xout, cost, Fx, Fu, Lx, Lu, Lxx, Lxu, Luu);
  HIP_TRY(hipGetLastError());
  return ASLR_OK;
}

int launch_dam_residuals_nj7_vsa(const KArgs &k, int mi, int n, const double *x, const double *u, double *r, int nr, hipStream_t st) {
  dim3 grid((n + 63) / 64), block(64);
  hipLaunchKernelGGL((dam_residual_kernel<7, ASLR_DAM_VSA, false>), grid, block, 0, st, k.desc, mi, k.frame_ref, n, x, u, r, nr);
  HIP_TRY(hipGetLastError());
  return ASLR_OK;
}

} // namespace aslr
