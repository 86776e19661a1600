// forward pass (rollout + trial costs + line search) instantiated for nj = 2
#include "aslr_forward.inc.hpp"

namespace aslr {
namespace {
template <int DAM, bool PLANAR>
void launch_forward_t(const KArgs &k, const SolverDev &sd, const ModelLimits &lim, hipStream_t st) {
  const int nb = k.b1 - k.b0; // trajectories of this launch
  const int T = k.T, cgx = (nb + 63) / 64;
  const dim3 grid((nb + ASLR_ROLLOUT_TPW - 1) / ASLR_ROLLOUT_TPW), block(64), sgrid(cgx), ugrid(cgx, ASLR_NALPHA);
  const bool fddp = sd.solver == ASLR_SOLVER_FDDP;
  constexpr bool CANFAST = PLANAR;
  const bool fast = CANFAST && k.planar_reach;
  auto rollout = [&](const KArgs &a) {
    if (fddp) hipLaunchKernelGGL((rollout_kernel<2, DAM, PLANAR, true>), grid, block, 0, st, a, sd, lim);
    else hipLaunchKernelGGL((rollout_kernel<2, DAM, PLANAR, false>), grid, block, 0, st, a, sd, lim);
  };
  auto costs = [&](const KArgs &a) { // knots [a.seg_t0, a.seg_t1]
    const dim3 cgrid(cgx, a.seg_t1 - a.seg_t0 + 1, ASLR_NALPHA);
    if constexpr (CANFAST) {
      if (fast) { hipLaunchKernelGGL((trial_cost_kernel<2, DAM, PLANAR, true>), cgrid, block, 0, st, a, sd); return; }
    }
    hipLaunchKernelGGL((trial_cost_kernel<2, DAM, PLANAR, false>), cgrid, block, 0, st, a, sd);
  };
  if (PLANAR && k.pipeline && !fast && T >= 16) {
    // rollout of the first half; then ONE launch in which the rollout continues over the second half while the trial costs
    // of the first half are evaluated next to it; then the trial costs of the second half
    const int Tm = T / 2;
    KArgs a = k;
    a.seg_t0 = 0; a.seg_t1 = Tm;
    rollout(a);
    const int ncost = cgx * Tm * ASLR_NALPHA; // knots 0 .. Tm-1
    const dim3 fgrid(grid.x + ncost);
    if constexpr (PLANAR) {
      if (fddp) hipLaunchKernelGGL((rollout_and_cost_kernel<2, DAM, true, true, false>), fgrid, block, 0, st, k, sd, lim, (int)grid.x, Tm, T, cgx, 0, Tm);
      else hipLaunchKernelGGL((rollout_and_cost_kernel<2, DAM, true, false, false>), fgrid, block, 0, st, k, sd, lim, (int)grid.x, Tm, T, cgx, 0, Tm);
    }
    a.seg_t0 = Tm; a.seg_t1 = T;
    costs(a);
  } else {
    rollout(k);
    costs(k);
  }
  hipLaunchKernelGGL((sum_cost_kernel<2>), ugrid, block, 0, st, k, sd);
  hipLaunchKernelGGL((select_kernel<2>), sgrid, block, 0, st, k, sd);
}
} // namespace

int launch_forward_nj2(const KArgs &k, int dam, const SolverDev &sd, const ModelLimits &lim, hipStream_t st) {
  if (dam == ASLR_DAM_SEA) {
    if (k.planar) launch_forward_t<ASLR_DAM_SEA, true>(k, sd, lim, st); else launch_forward_t<ASLR_DAM_SEA, false>(k, sd, lim, st);
  } else if (dam == ASLR_DAM_VSA) {
    if (k.planar) launch_forward_t<ASLR_DAM_VSA, true>(k, sd, lim, st); else launch_forward_t<ASLR_DAM_VSA, false>(k, sd, lim, st);
  } else {
    snprintf(err_buf(), kErrLen, "forward: unsupported (nj=2, dam=%d)", dam);
    return ASLR_E_INVALID;
  }
  HIP_TRY(hipGetLastError());
  return ASLR_OK;
}
} // namespace aslr
