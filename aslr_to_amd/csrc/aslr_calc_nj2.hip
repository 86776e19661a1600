// calc / calcDiff kernels instantiated for nj = 2 (3-D and planar chain paths)
#include "aslr_calc.inc.hpp"

namespace aslr {

int launch_calc_nj2(const KArgs &k, int dam, bool diff, int mode, double th_gaptol, hipStream_t st) {
  dim3 grid((k.b1 - k.b0 + 63) / 64, k.T + 1), block(64);
  if (dam == ASLR_DAM_SEA) {
    if (k.planar) {
      if (diff && (mode & kModeSkipConst)) hipLaunchKernelGGL((calc_kernel<2, ASLR_DAM_SEA, true, true, false, true>), grid, block, 0, st, k, mode, th_gaptol);
      else if (diff) hipLaunchKernelGGL((calc_kernel<2, ASLR_DAM_SEA, true, true>), grid, block, 0, st, k, mode, th_gaptol);
      else hipLaunchKernelGGL((calc_kernel<2, ASLR_DAM_SEA, false, true>), grid, block, 0, st, k, mode, th_gaptol);
    } else {
      if (diff && (mode & kModeSkipConst)) hipLaunchKernelGGL((calc_kernel<2, ASLR_DAM_SEA, true, false, false, true>), grid, block, 0, st, k, mode, th_gaptol);
      else if (diff) hipLaunchKernelGGL((calc_kernel<2, ASLR_DAM_SEA, true, false>), grid, block, 0, st, k, mode, th_gaptol);
      else hipLaunchKernelGGL((calc_kernel<2, ASLR_DAM_SEA, false, false>), grid, block, 0, st, k, mode, th_gaptol);
    }
    HIP_TRY(hipGetLastError());
    return ASLR_OK;
  }
  if (dam == ASLR_DAM_VSA) {
    if (k.planar) {
      if (diff && (mode & kModeSkipConst)) hipLaunchKernelGGL((calc_kernel<2, ASLR_DAM_VSA, true, true, false, true>), grid, block, 0, st, k, mode, th_gaptol);
      else if (diff) hipLaunchKernelGGL((calc_kernel<2, ASLR_DAM_VSA, true, true>), grid, block, 0, st, k, mode, th_gaptol);
      else hipLaunchKernelGGL((calc_kernel<2, ASLR_DAM_VSA, false, true>), grid, block, 0, st, k, mode, th_gaptol);
    } else {
      if (diff && (mode & kModeSkipConst)) hipLaunchKernelGGL((calc_kernel<2, ASLR_DAM_VSA, true, false, false, true>), grid, block, 0, st, k, mode, th_gaptol);
      else if (diff) hipLaunchKernelGGL((calc_kernel<2, ASLR_DAM_VSA, true, false>), grid, block, 0, st, k, mode, th_gaptol);
      else hipLaunchKernelGGL((calc_kernel<2, ASLR_DAM_VSA, false, false>), grid, block, 0, st, k, mode, th_gaptol);
    }
    HIP_TRY(hipGetLastError());
    return ASLR_OK;
  }
  snprintf(err_buf(), kErrLen, "calc: unsupported (nj=2, dam=%d)", dam);
  return ASLR_E_INVALID;
}

int launch_dam_eval_nj2(const KArgs &k, int dam, int mi, int n, const double *x, const double *u, double *xout,
                        double *cost, double *Fx, double *Fu, double *Lx, double *Lu, double *Lxx, double *Lxu,
                        double *Luu, hipStream_t st) {
  dim3 grid((n + 63) / 64), block(64);
  if (dam == ASLR_DAM_SEA) {
    if (k.planar) hipLaunchKernelGGL((dam_eval_kernel<2, ASLR_DAM_SEA, true>), grid, block, 0, st, k.desc, mi, k.frame_ref, n, x, u, xout, cost, Fx, Fu, Lx, Lu, Lxx, Lxu, Luu);
    else hipLaunchKernelGGL((dam_eval_kernel<2, ASLR_DAM_SEA, false>), grid, block, 0, st, k.desc, mi, k.frame_ref, n, x, u, xout, cost, Fx, Fu, Lx, Lu, Lxx, Lxu, Luu);
    HIP_TRY(hipGetLastError());
    return ASLR_OK;
  }
  if (dam == ASLR_DAM_VSA) {
    if (k.planar) hipLaunchKernelGGL((dam_eval_kernel<2, ASLR_DAM_VSA, true>), grid, block, 0, st, k.desc, mi, k.frame_ref, n, x, u, xout, cost, Fx, Fu, Lx, Lu, Lxx, Lxu, Luu);
    else hipLaunchKernelGGL((dam_eval_kernel<2, ASLR_DAM_VSA, false>), grid, block, 0, st, k.desc, mi, k.frame_ref, n, x, u, xout, cost, Fx, Fu, Lx, Lu, Lxx, Lxu, Luu);
    HIP_TRY(hipGetLastError());
    return ASLR_OK;
  }
  snprintf(err_buf(), kErrLen, "dam_eval: unsupported (nj=2, dam=%d)", dam);
  return ASLR_E_INVALID;
}

int launch_dam_residuals_nj2(const KArgs &k, int dam, int mi, int n, const double *x, const double *u, double *r, int nr, hipStream_t st) {
  if (dam == ASLR_DAM_SEA) return launch_dam_residuals_t<2, ASLR_DAM_SEA>(k, mi, n, x, u, r, nr, st);
  if (dam == ASLR_DAM_VSA) return launch_dam_residuals_t<2, ASLR_DAM_VSA>(k, mi, n, x, u, r, nr, st);
  snprintf(err_buf(), kErrLen, "dam_residuals: unsupported (nj=2, dam=%d)", dam);
  return ASLR_E_INVALID;
}

int launch_frame_placement_nj2(const KArgs &k, int fj, const FrameArg &F, int n, const double *x, int64_t stride, double *out, hipStream_t st) {
  return launch_frame_placement_t<2>(k, fj, F, n, x, stride, out, st);
}

int launch_quasi_static_nj2(const KArgs &k, int dam, int maxiter, double tol, int32_t *iters, hipStream_t st) {
  dim3 grid((k.B + 63) / 64, k.T), block(64);
  if (dam == ASLR_DAM_SEA) {
    if (k.planar) hipLaunchKernelGGL((quasi_static_kernel<2, ASLR_DAM_SEA, true>), grid, block, 0, st, k, maxiter, tol, iters);
    else hipLaunchKernelGGL((quasi_static_kernel<2, ASLR_DAM_SEA, false>), grid, block, 0, st, k, maxiter, tol, iters);
    HIP_TRY(hipGetLastError());
    return ASLR_OK;
  }
  if (dam == ASLR_DAM_VSA) {
    if (k.planar) hipLaunchKernelGGL((quasi_static_kernel<2, ASLR_DAM_VSA, true>), grid, block, 0, st, k, maxiter, tol, iters);
    else hipLaunchKernelGGL((quasi_static_kernel<2, ASLR_DAM_VSA, false>), grid, block, 0, st, k, maxiter, tol, iters);
    HIP_TRY(hipGetLastError());
    return ASLR_OK;
  }
  snprintf(err_buf(), kErrLen, "quasi_static: unsupported (nj=2, dam=%d)", dam);
  return ASLR_E_INVALID;
}

} // namespace aslr

#ifdef ASLR_BWD_PROFILE
// profile builds only (tools/calc_regions.py): read / reset the region table of this translation unit
extern "C" int aslr_debug_calc_prof(unsigned long long *out32, int reset) {
  if (out32 && hipMemcpyFromSymbol(out32, HIP_SYMBOL(aslr::aslr_bwd_prof_dev), 32 * sizeof(unsigned long long)) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[32] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(aslr::aslr_bwd_prof_dev), z, sizeof(z)) != hipSuccess) return -1;
  }
  return 0;
}
#endif
