// forward pass instantiated for nj = 2
#include "aslr_forward.inc.hpp"

namespace aslr {
int launch_forward_nj2(const KArgs &k, int dam, const SolverDev &sd, hipStream_t st) {
  dim3 grid((k.B + 3) / 4), block(64);
  if (dam == ASLR_DAM_SEA) {
    if (k.planar) hipLaunchKernelGGL((forward_kernel<2, ASLR_DAM_SEA, true>), grid, block, 0, st, k, sd);
    else hipLaunchKernelGGL((forward_kernel<2, ASLR_DAM_SEA, false>), grid, block, 0, st, k, sd);
    HIP_TRY(hipGetLastError());
    return ASLR_OK;
  }
  if (dam == ASLR_DAM_VSA) {
    if (k.planar) hipLaunchKernelGGL((forward_kernel<2, ASLR_DAM_VSA, true>), grid, block, 0, st, k, sd);
    else hipLaunchKernelGGL((forward_kernel<2, ASLR_DAM_VSA, false>), grid, block, 0, st, k, sd);
    HIP_TRY(hipGetLastError());
    return ASLR_OK;
  }
  snprintf(err_buf(), kErrLen, "forward: unsupported (nj=2, dam=%d)", dam);
  return ASLR_E_INVALID;
}
} // namespace aslr
