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
  const int nseg = k.pipeline >= 2 ? (k.pipeline <= 4 ? k.pipeline : 4) : (k.pipeline == 1 ? 2 : 1); // (1: two segments, n >= 2: n)
  if (PLANAR && nseg > 1 && !fast && T >= 16) {
    // rollout of the first segment; then launches in which the rollout continues over the next segment while the trial costs
    // of the previous one are evaluated next to it; then the trial costs of the last segment
    int lo = 0, hi = T / nseg;
    KArgs a = k;
    a.seg_t0 = lo; a.seg_t1 = hi;
    rollout(a);
    for (int sgm = 1; sgm < nseg; ++sgm) {
      const int nlo = hi, nhi = sgm == nseg - 1 ? T : (T * (sgm + 1)) / nseg; // rollout [nlo, nhi], costs of the knots [lo, nlo)
      const int ncost = cgx * (nlo - lo) * ASLR_NALPHA;
      const dim3 fgrid(grid.x + ncost);
      if constexpr (PLANAR) {
        if (fddp) hipLaunchKernelGGL((rollout_and_cost_kernel<2, DAM, true, true, false>), fgrid, block, 0, st, k, sd, lim, (int)grid.x, nlo, nhi, cgx, lo, nlo - lo);
        else hipLaunchKernelGGL((rollout_and_cost_kernel<2, DAM, true, false, false>), fgrid, block, 0, st, k, sd, lim, (int)grid.x, nlo, nhi, cgx, lo, nlo - lo);
      }
      lo = nlo; hi = nhi;
    }
    a.seg_t0 = lo; a.seg_t1 = T;
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
