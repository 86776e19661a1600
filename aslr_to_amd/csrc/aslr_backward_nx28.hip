// backward pass instantiated for nx = 28 (7-DoF SEA, nu = 7)
#include "aslr_backward.inc.hpp"

namespace aslr {
int launch_backward_nx28(const KArgs &k, int nu, int hs, const SolverDev &sd, const ModelLimits &lim, bool all_feasible, hipStream_t st) {
  if (nu == 7) return hs == 2 ? launch_backward_t<28, 7, 2>(k, sd, lim, all_feasible, st) : launch_backward_t<28, 7, 1>(k, sd, lim, all_feasible, st);
  snprintf(err_buf(), kErrLen, "backward: unsupported (nx=28, nu=%d)", nu);
  return ASLR_E_INVALID;
}
} // namespace aslr
