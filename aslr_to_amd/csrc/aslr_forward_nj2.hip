// forward pass (rollout + trial costs + line search) instantiated for nj = 2
#include "aslr_forward.inc.hpp"

namespace aslr {
int launch_forward_nj2(const KArgs &k, int dam, const SolverDev &sd, const ModelLimits &lim, hipStream_t st) {
  const int nb = k.b1 - k.b0; // trajectories of this launch
  dim3 grid((nb + ASLR_ROLLOUT_TPW - 1) / ASLR_ROLLOUT_TPW), block(64), cgrid((nb + 63) / 64, k.T + 1, ASLR_NALPHA), sgrid((nb + 63) / 64),
      cblock(64), ugrid((nb + 63) / 64, ASLR_NALPHA);
  if (dam == ASLR_DAM_SEA) {
    if (k.planar) {
      if (sd.solver == ASLR_SOLVER_FDDP) hipLaunchKernelGGL((rollout_kernel<2, ASLR_DAM_SEA, true, true>), grid, block, 0, st, k, sd, lim);
      else hipLaunchKernelGGL((rollout_kernel<2, ASLR_DAM_SEA, true, false>), grid, block, 0, st, k, sd, lim);
      if (k.planar_reach) hipLaunchKernelGGL((trial_cost_kernel<2, ASLR_DAM_SEA, true, true>), cgrid, cblock, 0, st, k, sd);
      else hipLaunchKernelGGL((trial_cost_kernel<2, ASLR_DAM_SEA, true>), cgrid, cblock, 0, st, k, sd);
    } else {
      if (sd.solver == ASLR_SOLVER_FDDP) hipLaunchKernelGGL((rollout_kernel<2, ASLR_DAM_SEA, false, true>), grid, block, 0, st, k, sd, lim);
      else hipLaunchKernelGGL((rollout_kernel<2, ASLR_DAM_SEA, false, false>), grid, block, 0, st, k, sd, lim);
      hipLaunchKernelGGL((trial_cost_kernel<2, ASLR_DAM_SEA, false>), cgrid, cblock, 0, st, k, sd);
    }
    hipLaunchKernelGGL((sum_cost_kernel<2>), ugrid, block, 0, st, k, sd);
    hipLaunchKernelGGL((select_kernel<2>), sgrid, block, 0, st, k, sd);
    HIP_TRY(hipGetLastError());
    return ASLR_OK;
  }
  if (dam == ASLR_DAM_VSA) {
    if (k.planar) {
      if (sd.solver == ASLR_SOLVER_FDDP) hipLaunchKernelGGL((rollout_kernel<2, ASLR_DAM_VSA, true, true>), grid, block, 0, st, k, sd, lim);
      else hipLaunchKernelGGL((rollout_kernel<2, ASLR_DAM_VSA, true, false>), grid, block, 0, st, k, sd, lim);
      if (k.planar_reach) hipLaunchKernelGGL((trial_cost_kernel<2, ASLR_DAM_VSA, true, true>), cgrid, cblock, 0, st, k, sd);
      else hipLaunchKernelGGL((trial_cost_kernel<2, ASLR_DAM_VSA, true>), cgrid, cblock, 0, st, k, sd);
    } else {
      if (sd.solver == ASLR_SOLVER_FDDP) hipLaunchKernelGGL((rollout_kernel<2, ASLR_DAM_VSA, false, true>), grid, block, 0, st, k, sd, lim);
      else hipLaunchKernelGGL((rollout_kernel<2, ASLR_DAM_VSA, false, false>), grid, block, 0, st, k, sd, lim);
      hipLaunchKernelGGL((trial_cost_kernel<2, ASLR_DAM_VSA, false>), cgrid, cblock, 0, st, k, sd);
    }
    hipLaunchKernelGGL((sum_cost_kernel<2>), ugrid, block, 0, st, k, sd);
    hipLaunchKernelGGL((select_kernel<2>), sgrid, block, 0, st, k, sd);
    HIP_TRY(hipGetLastError());
    return ASLR_OK;
  }
  snprintf(err_buf(), kErrLen, "forward: unsupported (nj=2, dam=%d)", dam);
  return ASLR_E_INVALID;
}
} // namespace aslr
