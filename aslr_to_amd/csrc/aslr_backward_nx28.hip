// backward pass instantiated for nx = 28 (7-DoF SEA, nu = 7)
#include "aslr_backward_blk.inc.hpp"

namespace aslr {
// hs: 0 = default (block-per-trajectory LDS kernel, all three solvers),
//     1 / 2 = force the register-column kernel with that many lanes per column (tests, comparisons)
int launch_backward_nx28(const KArgs &k, int nu, int hs, const SolverDev &sd, const ModelLimits &lim, bool all_feasible, hipStream_t st) {
  if (nu == 7) {
    if (hs <= 0) return launch_backward_blk<28, 7>(k, sd, lim, all_feasible, hs == 0, st); // (-1: vector-FMA products)
    if (hs == 0) hs = k.B <= 8192 ? 2 : 1;
    return hs == 2 ? launch_backward_t<28, 7, 2>(k, sd, lim, all_feasible, st) : launch_backward_t<28, 7, 1>(k, sd, lim, all_feasible, st);
  }
  snprintf(err_buf(), kErrLen, "backward: unsupported (nx=28, nu=%d)", nu);
  return ASLR_E_INVALID;
}
} // namespace aslr

#ifdef ASLR_BWD_PROFILE
// profile builds only (tools/bwd_regions.py c5): read / reset the region table of this translation unit
extern "C" int aslr_debug_bwd_prof28(unsigned long long *out32, int reset) {
  if (out32 && hipMemcpyFromSymbol(out32, HIP_SYMBOL(aslr::aslr_bwd_prof_dev), 32 * sizeof(unsigned long long)) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[32] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(aslr::aslr_bwd_prof_dev), z, sizeof(z)) != hipSuccess) return -1;
  }
  return 0;
}
#endif
