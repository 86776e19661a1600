// forward pass (rollout + trial costs + line search) instantiated for nj = 7
#include "aslr_forward.inc.hpp"
#include "aslr_forward_team.inc.hpp"

namespace aslr {
int launch_forward_nj7(const KArgs &k, int dam, const SolverDev &sd, const ModelLimits &lim, hipStream_t st) {
  const int nb = k.b1 - k.b0; // trajectories of this launch
  dim3 block(64), cgrid((nb + 63) / 64, k.T + 1, ASLR_NALPHA), sgrid((nb + 63) / 64),
      cblock(64), ugrid((nb + 63) / 64, ASLR_NALPHA);
  if (dam == ASLR_DAM_SEA) {
    {
      // one block of 16 eight-lane teams per trajectory (aslr_forward_team.inc.hpp)
      if (sd.solver == ASLR_SOLVER_FDDP) hipLaunchKernelGGL((rollout_team_kernel<7, true>), dim3(nb), dim3(128), 0, st, k, sd, lim);
      else hipLaunchKernelGGL((rollout_team_kernel<7, false>), dim3(nb), dim3(128), 0, st, k, sd, lim);
      hipLaunchKernelGGL((trial_cost_kernel<7, ASLR_DAM_SEA, false>), cgrid, cblock, 0, st, k, sd);
    }
    hipLaunchKernelGGL((sum_cost_kernel<7>), ugrid, block, 0, st, k, sd);
    hipLaunchKernelGGL((select_kernel<7>), sgrid, block, 0, st, k, sd);
    HIP_TRY(hipGetLastError());
    return ASLR_OK;
  }
  snprintf(err_buf(), kErrLen, "forward: unsupported (nj=7, dam=%d)", dam);
  return ASLR_E_INVALID;
}
} // namespace aslr

#ifdef ASLR_BWD_PROFILE
// profile builds only (tools/fwd_regions_c5.py): read / reset the region table of this translation unit
extern "C" int aslr_debug_fwd_prof7(unsigned long long *out32, int reset) {
  if (out32 && hipMemcpyFromSymbol(out32, HIP_SYMBOL(aslr::aslr_bwd_prof_dev), 32 * sizeof(unsigned long long)) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[32] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(aslr::aslr_bwd_prof_dev), z, sizeof(z)) != hipSuccess) return -1;
  }
  return 0;
}
#endif
