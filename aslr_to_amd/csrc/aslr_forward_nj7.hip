// forward pass instantiated for nj = 7
#include "aslr_forward.inc.hpp"

namespace aslr {
int launch_forward_nj7(const KArgs &k, int dam, const SolverDev &sd, hipStream_t st) {
  dim3 grid((k.B + 3) / 4), block(64);
  if (dam == ASLR_DAM_SEA) {
    hipLaunchKernelGGL((forward_kernel<7, ASLR_DAM_SEA, false>), grid, block, 0, st, k, sd);
    HIP_TRY(hipGetLastError());
    return ASLR_OK;
  }
  snprintf(err_buf(), kErrLen, "forward: unsupported (nj=7, dam=%d)", dam);
  return ASLR_E_INVALID;
}
} // namespace aslr
