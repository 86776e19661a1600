// backward pass instantiated for nx = 8 (2-DoF SEA nu = 2, 2-DoF VSA nu = 4)
#include "aslr_backward.inc.hpp"

namespace aslr {
int launch_backward_nx8(const KArgs &k, int nu, int hs, const SolverDev &sd, const ModelLimits &lim, bool all_feasible, hipStream_t st) {
  // wider teams when the batch cannot fill the chip: 1024 trajectories (C2) are 256 waves with 16 lanes each and 512 with 32
  // (measured at nu = 2: 148 -> 134 us per sweep).  At nu = 4 two 32-lane teams per wave win as long as ALL sub-shards
  // together stay at one wave per SIMD (whole shard <= 2048 trajectories: 345 -> 334 us BoxDDP, 184 -> 161 us DDP at 1024) and
  // lose beyond (two waves per SIMD: the gains phase is per-wave work)
  if (hs == 0) hs = ((nu == 2 && k.b1 - k.b0 <= 2048) || k.B <= 2048) ? 4 : (k.B <= 8192 ? 2 : 1);
  if (nu == 2) {
    if (hs == 4) return launch_backward_t<8, 2, 4>(k, sd, lim, all_feasible, st);
    return hs == 2 ? launch_backward_t<8, 2, 2>(k, sd, lim, all_feasible, st) : launch_backward_t<8, 2, 1>(k, sd, lim, all_feasible, st);
  }
  if (nu == 4) {
    if (hs == 4) return launch_backward_t<8, 4, 4>(k, sd, lim, all_feasible, st);
    return hs == 2 ? launch_backward_t<8, 4, 2>(k, sd, lim, all_feasible, st) : launch_backward_t<8, 4, 1>(k, sd, lim, all_feasible, st);
  }
  snprintf(err_buf(), kErrLen, "backward: unsupported (nx=8, nu=%d)", nu);
  return ASLR_E_INVALID;
}
} // namespace aslr

#ifdef ASLR_BWD_PROFILE
// profile builds only (tools/bwd_regions.py): read / reset the region table
extern "C" int aslr_debug_bwd_prof(unsigned long long *out32, int reset) {
  if (out32 && hipMemcpyFromSymbol(out32, HIP_SYMBOL(aslr::aslr_bwd_prof_dev), 32 * sizeof(unsigned long long)) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[32] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(aslr::aslr_bwd_prof_dev), z, sizeof(z)) != hipSuccess) return -1;
  }
  return 0;
}
#endif
