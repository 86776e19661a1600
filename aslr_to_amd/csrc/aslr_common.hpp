// aslr_common.hpp -- shared by the per-kernel translation units of libaslr_to_hip.so
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "aslr_device.hpp"

namespace aslr {

char *err_buf();            // thread-local error string (aslr_last_error)
constexpr int kErrLen = 512;

#define HIP_TRY(expr)                                                                               \
  do {                                                                                              \
    hipError_t e_ = (expr);                                                                         \
    if (e_ != hipSuccess) {                                                                         \
      snprintf(aslr::err_buf(), aslr::kErrLen, "%s -> %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, \
               __LINE__);                                                                           \
      return ASLR_E_HIP;                                                                            \
    }                                                                                               \
  } while (0)

constexpr int rec_len_c(int nx, int nu) { return (2 * nx * nx + 2 * nx * nu + nu * nu + nx + nu + 15) / 16 * 16; }

// kernel argument block: device pointers into the workspace
struct KArgs {
  const DevDesc *desc;
  const int32_t *node_model;
  const double *x0;
  const double *frame_ref; // nullable
  double *xs, *us, *xnext, *cost, *deriv, *gaps, *kgain, *kff, *qu, *vx, *vxx, *xs_try, *us_try, *vxxf, *cost_try, *dyn;
  double *traj_f;
  int32_t *traj_i;
  int32_t B, T;
  // trajectories [b0, b1) of the shard this launch covers (the whole shard unless the handle iterates sub-shards on
  // their own streams, aslr_set_subshards): array strides stay B, grids are sized for b1 - b0
  int32_t b0, b1;
  int32_t planar; // the chain qualifies for the planar dynamics path (DevDesc::planar.ok)
  int32_t planar_reach; // ... and the frame-placement costs for the closed-form residual (DevDesc::planar.reach_ok)
  double *iter_log;  // per-iteration log [log_cap][ASLR_LOG_COUNT][B] (aslr_set_iteration_log), or nullptr
  int32_t log_cap;
  // knots [seg_t0, seg_t1] of the horizon this launch covers (kernels that can work on a part of it: the rollout carries
  // its state over through the candidate it has stored; the trial costs are per knot): the whole horizon = [0, T]
  int32_t seg_t0, seg_t1;
  int32_t pipeline; // forward pass in two launches: the trial costs of the first half of the horizon run in the launch that rolls
                    // out the second half (rollout_and_cost_kernel; planar 2-joint chains).  0: off, 1 or 2: two segments (default), 3, 4: more (measured: no better)
};

// -DASLR_EXP_STAMP (tools/stamp_gaps.py): every kernel of an iteration records when its first wave started and its last one
// ended (s_memrealtime, 100 MHz) in the unused head of VXX: per sub-shard (quarter of the shard) 1024 words,
// [iteration][kernel 0..7][first start, last end]; word 1023 counts the iterations (select_kernel, the last kernel, bumps it).
#ifdef ASLR_EXP_STAMP
// (start: the first block of the grid -- blocks are dispatched in order; end: an atomic max over the blocks of the sweeps, whose
//  waves all run at once, and the LAST block of the large streaming grids, where 16 000 atomics on one word would be the kernel)
#define ASLR_STAMP_BEGIN(a, kid)                                                                                       \
  unsigned long long *stamp_w_ = reinterpret_cast<unsigned long long *>((a).vxx) + (size_t)((a).b0 * 4 / (a).B) * 1024;  \
  const unsigned long long stamp_it_ = *reinterpret_cast<volatile unsigned long long *>(stamp_w_ + 1023);               \
  if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && stamp_it_ < 60)                     \
    stamp_w_[(stamp_it_ * 8 + (kid)) * 2] = (unsigned long long)wall_clock64()
#define ASLR_STAMP_END(kid, every_block)                                                                               \
  do {                                                                                                                 \
    if (threadIdx.x == 0 && stamp_it_ < 60) {                                                                          \
      if (every_block) atomicMax(stamp_w_ + (stamp_it_ * 8 + (kid)) * 2 + 1, (unsigned long long)wall_clock64());      \
      else if (blockIdx.x == gridDim.x - 1 && blockIdx.y == gridDim.y - 1 && blockIdx.z == gridDim.z - 1)              \
        stamp_w_[(stamp_it_ * 8 + (kid)) * 2 + 1] = (unsigned long long)wall_clock64();                                \
    }                                                                                                                  \
  } while (0)
#define ASLR_STAMP_NEXT() do { if (blockIdx.x == 0 && threadIdx.x == 0) stamp_w_[1023] = stamp_it_ + 1; } while (0)
#else
#define ASLR_STAMP_BEGIN(a, kid)
#define ASLR_STAMP_END(kid, every_block)
#define ASLR_STAMP_NEXT()
#endif

// Line-search candidates (XS_TRY / US_TRY, layout in include/aslr_to_amd.h): 16-byte piece p of trajectory b at knot t of
// step length ai; W = doubles per candidate, TK = knots stored (T + 1 or T)
template <int W>
ASLR_DEV size_t cand_piece(int ai, int t, int b, int p, int B, int TK) {
  const size_t slab = (size_t)ASLR_CAND_SLAB(B, W);
  const size_t in_slab = ASLR_CAND_INTERLEAVED(W) ? ((size_t)(b >> 2) * (W / 2) + p) * 8 + (size_t)(b & 3) * 2 : (size_t)b * W + 2 * p;
  return ((size_t)ai * TK + t) * slab + in_slab;
}

// node -> action-model index, read through the constant address space: the table is never written by a kernel, and a
// scalar load keeps it out of vmcnt (as a vector load its wait drained every prefetch issued just before it)
ASLR_DEV int node_model_at(const KArgs &a, int t) {
  return ((const int32_t __attribute__((address_space(4))) *)(a.node_model))[t];
}

// solver parameters by value
struct SolverDev {
  int32_t solver, fixed_iterations;
  double th_stop, th_grad, th_gaptol, th_stepdec, th_stepinc, th_acceptstep, th_acceptnegstep;
  double reg_min, reg_max, reg_incfactor, reg_decfactor;
  int32_t boxqp_maxiter;
  double boxqp_th_acceptstep, boxqp_th_grad, boxqp_reg;
  int32_t standalone; // 1: API-level single pass (no retry, no solver-state updates)
  int32_t store_v;    // 1: write VX / VXX
  int32_t maxiter_traj; // > 0: a trajectory stops by itself after this many iterations (pool solves); 0: the host loop bounds them
};

// control limits of the (at most ASLR_MAX_MODELS) action models, passed by value so the backward loop
// never chases model pointers
struct ModelLimits {
  int32_t has[ASLR_MAX_MODELS];
  double lb[ASLR_MAX_MODELS][ASLR_MAX_NU], ub[ASLR_MAX_MODELS][ASLR_MAX_NU];
};

__device__ __forceinline__ bool is_bad(double v) { return isnan(v) || isinf(v) || v >= 1e30; }
// Crocoddyl's raiseIfNaN(v.lpNorm<Infinity>()) on a short vector: `s1` is the 1-norm of the same entries, which the
// kernels accumulate anyway.  s1 < 1e30 proves every entry finite and below 1e30 (the common case: one compare);
// otherwise (NaN, Inf, or a 1-norm that crossed 1e30 while the inf-norm may not have) the entries decide one by one.
template <int N>
__device__ __forceinline__ bool inf_norm_bad(double s1, const double (&v)[N]) {
  if (__ballot(!(s1 < 1e30)) == 0ull) return false; // (wave-uniform: a real branch around the rare path)
  bool bad = false;
  _Pragma("unroll") for (int i = 0; i < N; ++i) bad = bad || is_bad(fabs(v[i]));
  return bad;
}

// calc_kernel mode bits
constexpr int kModeCommit = 1;    // copy the accepted candidate XS_TRY/US_TRY[acc] into XS/US
constexpr int kModeSolver = 2;    // honour RECALC/DONE flags and compute gaps
constexpr int kModeNoCompute = 4;
constexpr int kModeSkipConst = 8; // record chunks that depend on the model only are in place already: do not rewrite them

// launchers, one translation unit per (kernel family, size)
int launch_calc_nj2(const KArgs &k, int dam, bool diff, int mode, double th_gaptol, hipStream_t st);
int launch_calc_nj7(const KArgs &k, int dam, bool diff, int mode, double th_gaptol, hipStream_t st);
int launch_dam_eval_nj2(const KArgs &k, int dam, int mi, int n, const double *x, const double *u, double *xout,
                        double *cost, double *Fx, double *Fu, double *Lx, double *Lu, double *Lxx, double *Lxu,
                        double *Luu, hipStream_t st);
int launch_dam_eval_nj7(const KArgs &k, int dam, int mi, int n, const double *x, const double *u, double *xout,
                        double *cost, double *Fx, double *Fu, double *Lx, double *Lu, double *Lxx, double *Lxu,
                        double *Luu, hipStream_t st);
int launch_dam_residuals_nj2(const KArgs &k, int dam, int mi, int n, const double *x, const double *u, double *r, int nr, hipStream_t st);
int launch_dam_residuals_nj7(const KArgs &k, int dam, int mi, int n, const double *x, const double *u, double *r, int nr, hipStream_t st);
struct FrameArg { double R[9], p[3]; }; // local placement of a frame on its joint, by value
int launch_frame_placement_nj2(const KArgs &k, int fj, const FrameArg &F, int n, const double *x, int64_t stride, double *out, hipStream_t st);
int launch_frame_placement_nj7(const KArgs &k, int fj, const FrameArg &F, int n, const double *x, int64_t stride, double *out, hipStream_t st);
int launch_calc_nj7_vsa(const KArgs &k, bool diff, int mode, double th_gaptol, hipStream_t st);
int launch_dam_eval_nj7_vsa(const KArgs &k, int mi, int n, const double *x, const double *u, double *xout, double *cost,
                            double *Fx, double *Fu, double *Lx, double *Lu, double *Lxx, double *Lxu, double *Luu, hipStream_t st);
int launch_dam_residuals_nj7_vsa(const KArgs &k, int mi, int n, const double *x, const double *u, double *r, int nr, hipStream_t st);
int launch_quasi_static_nj2(const KArgs &k, int dam, int maxiter, double tol, int32_t *iters, hipStream_t st);
int launch_quasi_static_nj7(const KArgs &k, int dam, int maxiter, double tol, int32_t *iters, hipStream_t st);
int launch_backward_nx8(const KArgs &k, int nu, int hs, const SolverDev &sd, const ModelLimits &lim, bool all_feasible, hipStream_t st);
int launch_backward_nx28(const KArgs &k, int nu, int hs, const SolverDev &sd, const ModelLimits &lim, bool all_feasible, hipStream_t st);
int launch_forward_nj2(const KArgs &k, int dam, const SolverDev &sd, const ModelLimits &lim, hipStream_t st);
int launch_forward_nj7(const KArgs &k, int dam, const SolverDev &sd, const ModelLimits &lim, hipStream_t st);

} // namespace aslr
