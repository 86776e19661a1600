// calc / calcDiff kernels instantiated for nj = 7
#include "aslr_calc.inc.hpp"
#include "aslr_calc_team.inc.hpp"

namespace aslr {

int launch_calc_nj7(const KArgs &k, int dam, bool diff, int mode, double th_gaptol, hipStream_t st) {
  dim3 grid((k.b1 - k.b0 + 63) / 64, k.T + 1), block(64);
  if (dam == ASLR_DAM_SEA) {
    {
      if (diff) { // rigid-body part by 8-lane teams (aslr_calc_team.inc.hpp), then products + costs + record per lane
        const dim3 tgrid((k.b1 - k.b0 + 7) / 8, k.T + 1);
        hipLaunchKernelGGL((dyn_team_kernel<7, 0>), tgrid, block, 0, st, k, mode);
        hipLaunchKernelGGL((dyn_team_kernel<7, 1>), tgrid, block, 0, st, k, mode);
        if (mode & kModeSkipConst) hipLaunchKernelGGL((calc_kernel<7, ASLR_DAM_SEA, true, false, true, true>), grid, block, 0, st, k, mode, th_gaptol);
        else hipLaunchKernelGGL((calc_kernel<7, ASLR_DAM_SEA, true, false, true>), grid, block, 0, st, k, mode, th_gaptol);
      }
      else hipLaunchKernelGGL((calc_kernel<7, ASLR_DAM_SEA, false, false>), grid, block, 0, st, k, mode, th_gaptol);
    }
    HIP_TRY(hipGetLastError());
    return ASLR_OK;
  }
  if (dam == ASLR_DAM_VSA) return launch_calc_nj7_vsa(k, diff, mode, th_gaptol, st); // (aslr_calc_nj7_vsa.hip)
  snprintf(err_buf(), kErrLen, "calc: unsupported (nj=7, dam=%d)", dam);
  return ASLR_E_INVALID;
}

int launch_dam_eval_nj7(const KArgs &k, int dam, int mi, int n, const double *x, const double *u, double *xout,
                        double *cost, double *Fx, double *Fu, double *Lx, double *Lu, double *Lxx, double *Lxu,
                        double *Luu, hipStream_t st) {
  dim3 grid((n + 63) / 64), block(64);
  if (dam == ASLR_DAM_SEA) {
    hipLaunchKernelGGL((dam_eval_kernel<7, ASLR_DAM_SEA, false>), grid, block, 0, st, k.desc, mi, k.frame_ref, n, x, u, xout, cost, Fx, Fu, Lx, Lu, Lxx, Lxu, Luu);
    HIP_TRY(hipGetLastError());
    return ASLR_OK;
  }
  if (dam == ASLR_DAM_VSA) return launch_dam_eval_nj7_vsa(k, mi, n, x, u, xout, cost, Fx, Fu, Lx, Lu, Lxx, Lxu, Luu, st);
  snprintf(err_buf(), kErrLen, "dam_eval: unsupported (nj=7, dam=%d)", dam);
  return ASLR_E_INVALID;
}

int launch_dam_residuals_nj7(const KArgs &k, int dam, int mi, int n, const double *x, const double *u, double *r, int nr, hipStream_t st) {
  dim3 grid((n + 63) / 64), block(64);
  if (dam == ASLR_DAM_SEA) {
    hipLaunchKernelGGL((dam_residual_kernel<7, ASLR_DAM_SEA, false>), grid, block, 0, st, k.desc, mi, k.frame_ref, n, x, u, r, nr);
    HIP_TRY(hipGetLastError());
    return ASLR_OK;
  }
  if (dam == ASLR_DAM_VSA) return launch_dam_residuals_nj7_vsa(k, mi, n, x, u, r, nr, st);
  snprintf(err_buf(), kErrLen, "dam_residuals: unsupported (nj=7, dam=%d)", dam);
  return ASLR_E_INVALID;
}

int launch_frame_placement_nj7(const KArgs &k, int fj, const FrameArg &F, int n, const double *x, int64_t stride, double *out, hipStream_t st) {
  dim3 grid((n + 63) / 64), block(64);
  hipLaunchKernelGGL((frame_placement_kernel<7, false>), grid, block, 0, st, k.desc, fj, F, n, x, (long long)stride, out);
  HIP_TRY(hipGetLastError());
  return ASLR_OK;
}

int launch_quasi_static_nj7(const KArgs &k, int dam, int maxiter, double tol, int32_t *iters, hipStream_t st) {
  dim3 grid((k.B + 63) / 64, k.T), block(64);
  if (dam == ASLR_DAM_SEA) {
    hipLaunchKernelGGL((quasi_static_kernel<7, ASLR_DAM_SEA, false>), grid, block, 0, st, k, maxiter, tol, iters);
    HIP_TRY(hipGetLastError());
    return ASLR_OK;
  }
  snprintf(err_buf(), kErrLen, "quasi_static: unsupported (nj=7, dam=%d)", dam);
  return ASLR_E_INVALID;
}

} // namespace aslr
