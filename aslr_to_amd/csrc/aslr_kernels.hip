// aslr_kernels.hip -- gfx950 kernels and the C ABI of include/aslr_to_amd.h.
//
// Three kernels make one lock-step DDP iteration over a batch of B independent shooting problems:
//
//   calc_kernel      one LANE per (trajectory, knot): commits the last accepted line-search
//                    candidate, evaluates IntegratedActionModelEulerASR.calc/calcDiff
//                    (integrated_action.py:13-42) and streams the 16-double-aligned DERIV record out
//                    through an LDS transpose so every global store is a full 128-B line.
//   backward_kernel  one TEAM of lanes per trajectory (8/16/32 lanes, "lane j owns column j" of every
//                    nx-column matrix, optional row split): SolverDDP.backwardPass + computeGains /
//                    BoxQP, Vxx/Vx resident in registers, per-knot blocks staged
//                    through LDS, next knot's record prefetched while the current one computes.
//                    Crocoddyl's catch-"backward_error"-and-regularise loop runs inside the kernel.
//   forward_kernel   one 16-lane team per trajectory, lane a rolls out step length 2^-a: all
//                    ASLR_NALPHA candidates of the sequential line search are evaluated at once and
//                    the FIRST acceptable one in Crocoddyl's order is taken, then
//                    the per-trajectory solver state (regularisation, feasibility, stop) is updated.
//
// Nothing here synchronises the device or allocates; all buffers live in the caller's workspace.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "aslr_device.hpp"

using namespace aslr;

namespace {

thread_local char g_err[512] = {0};

#define HIP_TRY(expr)                                                                            \
  do {                                                                                           \
    hipError_t e_ = (expr);                                                                      \
    if (e_ != hipSuccess) {                                                                      \
      snprintf(g_err, sizeof g_err, "%s -> %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__,  \
               __LINE__);                                                                        \
      return ASLR_E_HIP;                                                                         \
    }                                                                                            \
  } while (0)

constexpr int rec_len_c(int nx, int nu) { return (2 * nx * nx + 2 * nx * nu + nu * nu + nx + nu + 15) / 16 * 16; }

// kernel argument block: device pointers into the workspace
struct KArgs {
  const DevDesc *desc;
  const int32_t *node_model;
  const double *x0;
  const double *frame_ref; // nullable
  double *xs, *us, *xnext, *cost, *deriv, *gaps, *kgain, *kff, *qu, *vx, *vxx, *xs_try, *us_try, *vxxf;
  double *traj_f;
  int32_t *traj_i;
  int32_t B, T;
};

// solver parameters by value
struct SolverDev {
  int32_t solver, fixed_iterations;
  double th_stop, th_grad, th_gaptol, th_stepdec, th_stepinc, th_acceptstep, th_acceptnegstep;
  double reg_min, reg_max, reg_incfactor, reg_decfactor;
  int32_t boxqp_maxiter;
  double boxqp_th_acceptstep, boxqp_th_grad, boxqp_reg;
  int32_t standalone; // 1: API-level single pass (no retry, no solver-state updates)
  int32_t store_v;    // 1: write VX / VXX
};

__device__ __forceinline__ bool is_bad(double v) { return isnan(v) || isinf(v) || v >= 1e30; }

// =================================================================================================
// calc / calcDiff
// =================================================================================================
constexpr int kChunk = 16;            // doubles per knot per LDS flush (one 128-B line)
constexpr int kLdsStride = kChunk + 1; // odd stride: conflict-free ds_write_b64 across lanes

// mode bits
constexpr int kModeCommit = 1;  // copy the accepted candidate XS_TRY/US_TRY[acc] into XS/US
constexpr int kModeSolver = 2;  // honour RECALC/DONE flags and compute gaps
constexpr int kModeNoCompute = 4;

template <int NJ, int DAM, bool DIFF>
__global__ void __launch_bounds__(64) calc_kernel(KArgs a, int mode, double th_gaptol) {
  constexpr int NX = 4 * NJ, NU = ModelDims<NJ, DAM>::nu;
  using RL = RecLayout<NJ, NU>;
  constexpr int REC = RL::len;
  __shared__ double sm[64 * kLdsStride];
  __shared__ int act[64];

  const int lane = threadIdx.x, t = blockIdx.y, B = a.B, T = a.T;
  const int b0 = blockIdx.x * 64, bq = b0 + lane;
  const bool valid = bq < B;
  const int b = valid ? bq : B - 1;
  const int32_t *TI = a.traj_i;

  int acc = -1, recalc = 1, done = 0, feasible = 1;
  if (mode & (kModeCommit | kModeSolver)) {
    acc = TI[ASLR_TI_ACCEPTED * B + b];
    if (!(mode & kModeCommit)) acc = -1;
  }
  if (mode & kModeSolver) {
    recalc = TI[ASLR_TI_RECALC * B + b];
    done = TI[ASLR_TI_DONE * B + b];
    feasible = TI[ASLR_TI_FEASIBLE * B + b];
  }
  const size_t tb = (size_t)t * B + b;
  const size_t TB1 = (size_t)(T + 1) * B, TB = (size_t)T * B;

  // ---- x, u of this knot (from the accepted candidate when there is one) ----
  double x[NX], u[NU];
  {
    const double *src = acc >= 0 ? a.xs_try + ((size_t)acc * TB1 + tb) * NX : a.xs + tb * NX;
    ASLR_UNROLL for (int i = 0; i < NX; ++i) x[i] = src[i];
    if (acc >= 0 && valid) {
      double *dst = a.xs + tb * NX;
      ASLR_UNROLL for (int i = 0; i < NX; ++i) dst[i] = x[i];
    }
  }
  if (t < T) {
    const double *src = acc >= 0 ? a.us_try + ((size_t)acc * TB + tb) * NU : a.us + tb * NU;
    ASLR_UNROLL for (int i = 0; i < NU; ++i) u[i] = src[i];
    if (acc >= 0 && valid) {
      double *dst = a.us + tb * NU;
      ASLR_UNROLL for (int i = 0; i < NU; ++i) dst[i] = u[i];
    }
  } else {
    ASLR_UNROLL for (int i = 0; i < NU; ++i) u[i] = 0.0;
  }
  const bool compute = valid && recalc && !done && !(mode & kModeNoCompute);
  if (__ballot(compute) == 0ull) return; // wave-uniform

  const DevDesc &D = *a.desc;
  const DevModel &dm = D.models[a.node_model[t]];
  const double *fref = a.frame_ref ? a.frame_ref + 12 * (size_t)b : nullptr;

  double xnext[NX], cost = 0.0;
  KnotDiff<NJ, NU> kd;
  if (compute) {
    knot_eval<NJ, DAM, DIFF>(D.chain, dm, fref, x, t < T ? u : nullptr, xnext, cost, DIFF ? &kd : nullptr);
    double *xn = a.xnext + tb * NX;
    ASLR_UNROLL for (int i = 0; i < NX; ++i) xn[i] = xnext[i];
    a.cost[tb] = cost;
    // gaps (SolverDDP::calcDiff): f[0] = x0 - xs[0]; f[t+1] = xnext_t - xs[t+1]
    if ((mode & kModeSolver) && !feasible) {
      double mx = 0.0;
      if (t < T) {
        const size_t tb1 = tb + B;
        const double *src = acc >= 0 ? a.xs_try + ((size_t)acc * TB1 + tb1) * NX : a.xs + tb1 * NX;
        double *g = a.gaps + tb1 * NX;
        ASLR_UNROLL for (int i = 0; i < NX; ++i) {
          const double f = xnext[i] - src[i];
          g[i] = f;
          mx = fmax(mx, fabs(f));
        }
      }
      if (t == 0) {
        const double *x0 = a.x0 + (size_t)b * NX;
        double *g = a.gaps + tb * NX;
        ASLR_UNROLL for (int i = 0; i < NX; ++i) {
          const double f = x0[i] - x[i];
          g[i] = f;
          mx = fmax(mx, fabs(f));
        }
      }
      if (mx >= th_gaptol) a.traj_i[ASLR_TI_GAPFLAG * B + b] = 1;
    }
  }
  if constexpr (DIFF) {
  // ---- stream the record out: 16 doubles per knot per flush, transposed through LDS ----
  act[lane] = compute ? 1 : 0;
  const double dt = dm.m.dt;
  double *rec0 = a.deriv + ((size_t)t * B + b0) * REC;
  static_for<0, REC / kChunk>([&](auto cc) {
    constexpr int c = decltype(cc)::value;
    if (compute) {
      static_for<0, kChunk>([&](auto ii) {
        constexpr int i = decltype(ii)::value;
        sm[lane * kLdsStride + i] = rec_elem<NJ, NU, c * kChunk + i>(kd, dt);
      });
    }
    __syncthreads();
    ASLR_UNROLL for (int i = 0; i < kChunk / 2; ++i) {
      const int idx = lane + 64 * i, k = idx >> 3, e = (idx & 7) * 2;
      if (act[k]) {
        double2 v2;
        v2.x = sm[k * kLdsStride + e];
        v2.y = sm[k * kLdsStride + e + 1];
        *reinterpret_cast<double2 *>(rec0 + (size_t)k * REC + c * kChunk + e) = v2;
      }
    }
    __syncthreads();
  });
  }
}

// DAM-level evaluation of arbitrary points (aslr_dam_eval): dense continuous blocks, one lane per point
template <int NJ, int DAM>
__global__ void __launch_bounds__(64) dam_eval_kernel(const DevDesc *desc, int mi, const double *frame_ref, int n,
                                                      const double *xin, const double *uin, double *xout,
                                                      double *cost, double *Fx, double *Fu, double *Lx, double *Lu,
                                                      double *Lxx, double *Lxu, double *Luu) {
  constexpr int NX = 4 * NJ, NU = ModelDims<NJ, DAM>::nu, NV = 2 * NJ;
  const int p = blockIdx.x * 64 + threadIdx.x;
  if (p >= n) return;
  const DevDesc &D = *desc;
  const DevModel &dm = D.models[mi];
  double x[NX], u[NU], xnext[NX], c, xo[NV];
  ASLR_UNROLL for (int i = 0; i < NX; ++i) x[i] = xin[(size_t)p * NX + i];
  ASLR_UNROLL for (int i = 0; i < NU; ++i) u[i] = uin[(size_t)p * NU + i];
  KnotDiff<NJ, NU> kd;
  knot_eval<NJ, DAM, true>(D.chain, dm, frame_ref, x, u, xnext, c, &kd, xo);
  if (cost) cost[p] = c;
  if (xout) {
    ASLR_UNROLL for (int i = 0; i < NV; ++i) xout[(size_t)p * NV + i] = xo[i];
  }
  if (Fx) {
    double *o = Fx + (size_t)p * NV * NX;
    ASLR_UNROLL for (int i = 0; i < NJ; ++i)
      ASLR_UNROLL for (int j = 0; j < NJ; ++j) {
        o[i * NX + j] = kd.Aqq[i][j];
        o[i * NX + NJ + j] = kd.Aqm[i][j];
        o[i * NX + 2 * NJ + j] = kd.Aqv[i][j];
        o[i * NX + 3 * NJ + j] = 0.0;
        o[(NJ + i) * NX + j] = kd.Bk[i][j];
        o[(NJ + i) * NX + NJ + j] = -kd.Bk[i][j];
        o[(NJ + i) * NX + 2 * NJ + j] = 0.0;
        o[(NJ + i) * NX + 3 * NJ + j] = 0.0;
      }
  }
  if (Fu) {
    double *o = Fu + (size_t)p * NV * NU;
    ASLR_UNROLL for (int i = 0; i < NJ; ++i)
      ASLR_UNROLL for (int j = 0; j < NU; ++j) { o[i * NU + j] = kd.Ful[i][j]; o[(NJ + i) * NU + j] = kd.Fum[i][j]; }
  }
  if (Lx) { ASLR_UNROLL for (int i = 0; i < NX; ++i) Lx[(size_t)p * NX + i] = kd.Lx[i]; }
  if (Lu) { ASLR_UNROLL for (int i = 0; i < NU; ++i) Lu[(size_t)p * NU + i] = kd.Lu[i]; }
  if (Lxx) {
    double *o = Lxx + (size_t)p * NX * NX;
    for (int i = 0; i < NX * NX; ++i) o[i] = 0.0;
    ASLR_UNROLL for (int i = 0; i < NJ; ++i)
      ASLR_UNROLL for (int j = 0; j < NJ; ++j) o[i * NX + j] = kd.Lqq[i][j];
    ASLR_UNROLL for (int i = 0; i < NX; ++i) o[i * NX + i] += kd.Lxxd[i];
  }
  if (Lxu) { for (int i = 0; i < NX * NU; ++i) Lxu[(size_t)p * NX * NU + i] = 0.0; }
  if (Luu) {
    double *o = Luu + (size_t)p * NU * NU;
    for (int i = 0; i < NU * NU; ++i) o[i] = 0.0;
    ASLR_UNROLL for (int i = 0; i < NU; ++i) o[i * NU + i] = kd.Luud[i];
  }
}

// =================================================================================================
// backward pass
// =================================================================================================
// BoxQP on register arrays, evaluated redundantly by every lane of a team.  The free
// subspace is handled by masking: clamped rows/columns of H become identity rows, which makes the
// Cholesky of the masked matrix reproduce the factor of Hff exactly (the extra operands are 0 / 1).
template <int NU>
__device__ __forceinline__ bool boxqp(const double (&H)[NU][NU], const double (&q)[NU], const double (&lb)[NU],
                                      const double (&ub)[NU], double (&x)[NU], bool (&cm)[NU], double (&Hinv)[NU][NU],
                                      const SolverDev &sp) {
  bool bad = false, finished = false;
  ASLR_UNROLL for (int i = 0; i < NU; ++i) x[i] = fmax(fmin(x[i], ub[i]), lb[i]);
  ASLR_UNROLL for (int i = 0; i < NU; ++i) {
    cm[i] = false;
    ASLR_UNROLL for (int j = 0; j < NU; ++j) Hinv[i][j] = 0.0;
  }
  for (int it = 0; it < sp.boxqp_maxiter && !finished; ++it) {
    double g[NU];
    ASLR_UNROLL for (int i = 0; i < NU; ++i) {
      double s = q[i];
      ASLR_UNROLL for (int j = 0; j < NU; ++j) s += H[i][j] * x[j];
      g[i] = s;
    }
    double gnorm = 0.0;
    int nf = 0;
    ASLR_UNROLL for (int j = 0; j < NU; ++j) {
      cm[j] = (x[j] == lb[j] && g[j] > 0.0) || (x[j] == ub[j] && g[j] < 0.0);
      if (!cm[j]) { gnorm = fmax(gnorm, fabs(g[j])); ++nf; }
    }
    double L[NU][NU];
    ASLR_UNROLL for (int i = 0; i < NU; ++i)
      ASLR_UNROLL for (int j = 0; j < NU; ++j)
        L[i][j] = (cm[i] || cm[j]) ? (i == j ? 1.0 : 0.0) : (H[i][j] + (i == j ? sp.boxqp_reg : 0.0));
    const bool conv = (gnorm <= sp.boxqp_th_grad) || nf == 0;
    if (chol<NU>(L)) { if (nf > 0) bad = true; }
    // explicit inverse of the free block (Crocoddyl keeps Hff_inv and forms K = Quu_inv Qxu^T)
    ASLR_UNROLL for (int j = 0; j < NU; ++j) {
      double e[NU];
      ASLR_UNROLL for (int i = 0; i < NU; ++i) e[i] = (i == j && !cm[j]) ? 1.0 : 0.0;
      chol_solve<NU>(L, e);
      ASLR_UNROLL for (int i = 0; i < NU; ++i) Hinv[i][j] = (cm[i] || cm[j]) ? 0.0 : e[i];
    }
    if (conv || bad) { finished = true; continue; }
    // dx_f = -Hff^-1 (q_f + H_fc x_c) - x_f
    double rhs[NU], dx[NU];
    ASLR_UNROLL for (int i = 0; i < NU; ++i) {
      double s = -q[i];
      ASLR_UNROLL for (int j = 0; j < NU; ++j) if (cm[j]) s -= H[i][j] * x[j];
      rhs[i] = cm[i] ? 0.0 : s;
    }
    chol_solve<NU>(L, rhs);
    ASLR_UNROLL for (int i = 0; i < NU; ++i) dx[i] = cm[i] ? 0.0 : rhs[i] - x[i];
    double fold = 0.0;
    ASLR_UNROLL for (int i = 0; i < NU; ++i) {
      double s = 0.0;
      ASLR_UNROLL for (int j = 0; j < NU; ++j) s += H[i][j] * x[j];
      fold += 0.5 * x[i] * s + q[i] * x[i];
    }
    double alpha = 1.0;
    bool found = false;
    for (int al = 0; al < ASLR_NALPHA && !found; ++al, alpha *= 0.5) {
      double xn[NU], fnew = 0.0, gd = 0.0;
      ASLR_UNROLL for (int i = 0; i < NU; ++i) xn[i] = fmax(fmin(x[i] + alpha * dx[i], ub[i]), lb[i]);
      ASLR_UNROLL for (int i = 0; i < NU; ++i) {
        double s = 0.0;
        ASLR_UNROLL for (int j = 0; j < NU; ++j) s += H[i][j] * xn[j];
        fnew += 0.5 * xn[i] * s + q[i] * xn[i];
        gd += g[i] * (x[i] - xn[i]);
      }
      if (fold - fnew > sp.boxqp_th_acceptstep * gd) {
        ASLR_UNROLL for (int i = 0; i < NU; ++i) x[i] = xn[i];
        found = true;
      }
    }
  }
  return bad;
}

template <int NX, int NU, int HS>
struct BwdCfg {
  static constexpr int NXP = NX <= 8 ? 8 : 32;
  static constexpr int TEAM = NXP * HS;
  static constexpr int TPW = 64 / TEAM;
  static constexpr int RPL = (NX + HS - 1) / HS;
  static constexpr int REC = rec_len_c(NX, NU);
  static constexpr int oFx = 0, oFu = oFx + NX * NX, oLxx = oFu + NX * NU, oLxu = oLxx + NX * NX,
                       oLuu = oLxu + NX * NU, oLx = oLuu + NU * NU, oLu = oLx + NX;
  // LDS arrays per team (doubles)
  static constexpr int sRec = 0, sAT = sRec + REC, sBT = sAT + NX * NX, sQux = sBT + NX * NU,
                       sVT = sQux + NU * NX, sQuu = sVT + NX * NX, sQu = sQuu + NU * NU, sVx = sQu + NU,
                       sEnd = sVx + NX;
  static constexpr int LDS_TEAM = (sEnd + 1) / 2 * 2;
  static constexpr int NPRE = (REC / 2 + TEAM - 1) / TEAM; // double2 prefetch registers per lane
};

template <int NX, int NU, int HS>
__global__ void __launch_bounds__(64) backward_kernel(KArgs a, SolverDev sp) {
  using C = BwdCfg<NX, NU, HS>;
  constexpr int NXP = C::NXP, TEAM = C::TEAM, TPW = C::TPW, RPL = C::RPL, REC = C::REC;
  extern __shared__ double smem[];

  const int lane = threadIdx.x, team = lane / TEAM, lt = lane % TEAM, j = lt % NXP, h = lt / NXP;
  const int B = a.B, T = a.T;
  const int bq = blockIdx.x * TPW + team;
  const bool team_valid = bq < B;
  const int b = team_valid ? bq : B - 1;
  const bool col_valid = j < NX;
  const int jj = col_valid ? j : NX - 1;
  const bool writer = team_valid && col_valid && h == 0;
  const int r0 = h * RPL;
  double *sm = smem + team * C::LDS_TEAM;
  double *rec = sm + C::sRec, *AT = sm + C::sAT, *BT = sm + C::sBT, *QuxL = sm + C::sQux, *VT = sm + C::sVT,
         *QuuL = sm + C::sQuu, *QuL = sm + C::sQu, *VxL = sm + C::sVx;

  int32_t *TI = a.traj_i;
  double *TF = a.traj_f;
  const DevDesc &D = *a.desc;

  // ---- prologue: solver-state bookkeeping that Crocoddyl does inside calcDiff ----
  int done = 0, feasible = TI[ASLR_TI_FEASIBLE * B + b], status = TI[ASLR_TI_STATUS * B + b];
  if (!sp.standalone) {
    done = TI[ASLR_TI_DONE * B + b];
    const int recalc = TI[ASLR_TI_RECALC * B + b];
    if (!done && recalc) {
      if (!feasible) feasible = TI[ASLR_TI_GAPFLAG * B + b] ? 0 : 1;
      // cost_ = sum of node costs, in node order
      double csum = 0.0;
      for (int t = 0; t <= T; ++t) csum += a.cost[(size_t)t * B + b];
      if (lt == 0 && team_valid) TF[ASLR_TF_COST * B + b] = csum;
    }
    __syncthreads();
    if (lt == 0 && team_valid) {
      TI[ASLR_TI_FEASIBLE * B + b] = feasible;
      TI[ASLR_TI_ACCEPTED * B + b] = -1;
      TI[ASLR_TI_GAPFLAG * B + b] = 0;
    }
  }
  bool need = team_valid && !done;
  if (__ballot(need) == 0ull) return;
  double xreg = TF[ASLR_TF_XREG * B + b];
  const bool fddp = sp.solver == ASLR_SOLVER_FDDP;
  const bool box = sp.solver == ASLR_SOLVER_BOXDDP;

  double d1 = 0.0, d2 = 0.0, stop = 0.0, dgf = 0.0, dqf = 0.0;
  while (__ballot(need) != 0ull) {
    bool failed = false;
    d1 = d2 = stop = dgf = dqf = 0.0;
    double Pcol[NX], pvec[NX], Vx_own;
    // ---- terminal node: Vxx = Lxx (+xreg), Vx = Lx (+ Vxx f) ----
    {
      const double *rT = a.deriv + ((size_t)T * B + b) * REC;
      ASLR_UNROLL for (int r = 0; r < NX; ++r) Pcol[r] = rT[C::oLxx + r * NX + jj] + (r == jj && !isnan(xreg) ? xreg : 0.0);
      Vx_own = rT[C::oLx + jj];
      if (!feasible) {
        const double *f = a.gaps + ((size_t)T * B + b) * NX;
        double vf = 0.0;
        ASLR_UNROLL for (int r = 0; r < NX; ++r) vf += Pcol[r] * f[r];
        Vx_own += vf;
        if (fddp) {
          dgf -= Vx_own * f[jj];
          dqf += f[jj] * vf;
          if (writer && need) a.vxxf[((size_t)T * B + b) * NX + jj] = vf;
        }
      }
      if (sp.store_v && writer && need) {
        a.vx[((size_t)T * B + b) * NX + jj] = Vx_own;
        double *o = a.vxx + ((size_t)T * B + b) * NX * NX;
        ASLR_UNROLL for (int r = 0; r < NX; ++r) o[r * NX + jj] = Pcol[r];
      }
      __syncthreads();
      VxL[jj] = Vx_own;
      __syncthreads();
      ASLR_UNROLL for (int l = 0; l < NX; ++l) pvec[l] = VxL[l];
    }
    // ---- prefetch the record of knot T-1 ----
    double2 pre[C::NPRE];
    {
      const double2 *src = reinterpret_cast<const double2 *>(a.deriv + ((size_t)(T - 1) * B + b) * REC);
      ASLR_UNROLL for (int i = 0; i < C::NPRE; ++i) {
        const int idx = lt + TEAM * i;
        if (idx < REC / 2) pre[i] = src[idx];
      }
    }
    for (int t = T - 1; t >= 0; --t) {
      const size_t tb = (size_t)t * B + b;
      // stage the record in LDS, start the next load
      {
        double2 *dst = reinterpret_cast<double2 *>(rec);
        ASLR_UNROLL for (int i = 0; i < C::NPRE; ++i) {
          const int idx = lt + TEAM * i;
          if (idx < REC / 2) dst[idx] = pre[i];
        }
      }
      __syncthreads();
      if (t > 0) {
        const double2 *src = reinterpret_cast<const double2 *>(a.deriv + ((size_t)(t - 1) * B + b) * REC);
        ASLR_UNROLL for (int i = 0; i < C::NPRE; ++i) {
          const int idx = lt + TEAM * i;
          if (idx < REC / 2) pre[i] = src[idx];
        }
      }
      // ---- step 1: A = Fx^T P (my rows of column jj), Bc = Fu^T P (column jj), Qx, Qu ----
      double Fxcol[NX], Fucol[NX];
      ASLR_UNROLL for (int l = 0; l < NX; ++l) Fxcol[l] = rec[C::oFx + l * NX + jj];
      const int ju = jj < NU ? jj : NU - 1;
      ASLR_UNROLL for (int l = 0; l < NX; ++l) Fucol[l] = rec[C::oFu + l * NU + ju];
      {
        double Arow[RPL], Bc[NU];
        ASLR_UNROLL for (int i = 0; i < RPL; ++i) Arow[i] = 0.0;
        ASLR_UNROLL for (int c = 0; c < NU; ++c) Bc[c] = 0.0;
        ASLR_UNROLL for (int l = 0; l < NX; ++l) {
          ASLR_UNROLL for (int i = 0; i < RPL; ++i) {
            const int r = r0 + i < NX ? r0 + i : NX - 1;
            Arow[i] += rec[C::oFx + l * NX + r] * Pcol[l];
          }
          ASLR_UNROLL for (int c = 0; c < NU; ++c) Bc[c] += rec[C::oFu + l * NU + c] * Pcol[l];
        }
        ASLR_UNROLL for (int i = 0; i < RPL; ++i) if (r0 + i < NX) AT[jj * NX + r0 + i] = Arow[i];
        ASLR_UNROLL for (int c = 0; c < NU; ++c) BT[jj * NU + c] = Bc[c];
      }
      double Qx, Qu_own;
      {
        double s = 0.0, s2 = 0.0;
        ASLR_UNROLL for (int l = 0; l < NX; ++l) { s += Fxcol[l] * pvec[l]; s2 += Fucol[l] * pvec[l]; }
        Qx = rec[C::oLx + jj] + s;
        Qu_own = rec[C::oLu + ju] + s2;
      }
      __syncthreads();
      // ---- step 2: Qxx (my rows), Qux (column jj), Quu (column jj < NU) ----
      double Qxx[RPL], Qux[NU];
      {
        double acc[RPL], accu[NU], accq[NU];
        ASLR_UNROLL for (int i = 0; i < RPL; ++i) acc[i] = 0.0;
        ASLR_UNROLL for (int c = 0; c < NU; ++c) { accu[c] = 0.0; accq[c] = 0.0; }
        ASLR_UNROLL for (int l = 0; l < NX; ++l) {
          ASLR_UNROLL for (int i = 0; i < RPL; ++i) {
            const int r = r0 + i < NX ? r0 + i : NX - 1;
            acc[i] += AT[l * NX + r] * Fxcol[l];
          }
          ASLR_UNROLL for (int c = 0; c < NU; ++c) {
            const double bt = BT[l * NU + c];
            accu[c] += bt * Fxcol[l];
            accq[c] += bt * Fucol[l];
          }
        }
        ASLR_UNROLL for (int i = 0; i < RPL; ++i) {
          const int r = r0 + i < NX ? r0 + i : NX - 1;
          Qxx[i] = rec[C::oLxx + r * NX + jj] + acc[i];
        }
        ASLR_UNROLL for (int c = 0; c < NU; ++c) {
          Qux[c] = rec[C::oLxu + jj * NU + c] + accu[c];
          QuxL[c * NX + jj] = Qux[c];
        }
        if (j < NU) {
          ASLR_UNROLL for (int c = 0; c < NU; ++c)
            QuuL[c * NU + j] = rec[C::oLuu + c * NU + j] + accq[c] + ((c == j && !isnan(xreg)) ? xreg : 0.0);
          QuL[j] = Qu_own;
        }
      }
      __syncthreads();
      // ---- step 3: gains (redundant per lane) ----
      double Quu[NU][NU], qu[NU], kv[NU], Kc[NU];
      ASLR_UNROLL for (int c = 0; c < NU; ++c) {
        qu[c] = QuL[c];
        ASLR_UNROLL for (int e = 0; e < NU; ++e) Quu[c][e] = QuuL[c * NU + e];
      }
      const DevModel &dm = D.models[a.node_model[t]];
      const bool use_box = box && dm.m.has_u_limits && feasible;
      if (use_box) {
        double lb[NU], ub[NU], xq[NU], Hinv[NU][NU];
        bool cm[NU];
        const double *ut = a.us + tb * NU, *k0 = a.kff + tb * NU;
        ASLR_UNROLL for (int c = 0; c < NU; ++c) {
          lb[c] = dm.m.u_lb[c] - ut[c];
          ub[c] = dm.m.u_ub[c] - ut[c];
          xq[c] = k0[c];
        }
        if (boxqp<NU>(Quu, qu, lb, ub, xq, cm, Hinv, sp)) failed = true;
        ASLR_UNROLL for (int c = 0; c < NU; ++c) {
          double s = 0.0;
          ASLR_UNROLL for (int e = 0; e < NU; ++e) s += Hinv[c][e] * Qux[e];
          Kc[c] = s;
          kv[c] = -xq[c];
          if (cm[c]) qu[c] = 0.0;
        }
      } else {
        double L[NU][NU];
        ASLR_UNROLL for (int c = 0; c < NU; ++c)
          ASLR_UNROLL for (int e = 0; e < NU; ++e) L[c][e] = Quu[c][e];
        if (chol<NU>(L)) failed = true;
        ASLR_UNROLL for (int c = 0; c < NU; ++c) { kv[c] = qu[c]; Kc[c] = Qux[c]; }
        chol_solve<NU>(L, kv);
        chol_solve<NU>(L, Kc);
      }
      double Quuk[NU];
      ASLR_UNROLL for (int c = 0; c < NU; ++c) {
        double s = 0.0;
        ASLR_UNROLL for (int e = 0; e < NU; ++e) s += Quu[c][e] * kv[e];
        Quuk[c] = s;
      }
      ASLR_UNROLL for (int c = 0; c < NU; ++c) { d1 += qu[c] * kv[c]; d2 -= kv[c] * Quuk[c]; stop += qu[c] * qu[c]; }
      {
        double s = 0.0, s2 = 0.0;
        ASLR_UNROLL for (int c = 0; c < NU; ++c) { s += Kc[c] * Quuk[c]; s2 += Kc[c] * qu[c]; }
        Vx_own = Qx + s - 2.0 * s2;
      }
      {
        double acc[RPL];
        ASLR_UNROLL for (int i = 0; i < RPL; ++i) acc[i] = 0.0;
        ASLR_UNROLL for (int c = 0; c < NU; ++c)
          ASLR_UNROLL for (int i = 0; i < RPL; ++i) {
            const int r = r0 + i < NX ? r0 + i : NX - 1;
            acc[i] += QuxL[c * NX + r] * Kc[c];
          }
        ASLR_UNROLL for (int i = 0; i < RPL; ++i) if (r0 + i < NX) VT[jj * NX + r0 + i] = Qxx[i] - acc[i];
      }
      const bool st_ok = writer && need && !failed;
      if (st_ok) {
        double *Kout = a.kgain + tb * NU * NX;
        ASLR_UNROLL for (int c = 0; c < NU; ++c) Kout[c * NX + jj] = Kc[c];
        if (j < NU) {
          double kj = kv[0], qj = qu[0];
          ASLR_UNROLL for (int c = 1; c < NU; ++c) if (c == j) { kj = kv[c]; qj = qu[c]; }
          a.kff[tb * NU + j] = kj;
          a.qu[tb * NU + j] = qj;
        }
      }
      __syncthreads();
      // ---- step 4: symmetrise, regularise, gap term, publish Vx ----
      ASLR_UNROLL for (int r = 0; r < NX; ++r) {
        const double cv = VT[jj * NX + r], rv = VT[r * NX + jj];
        Pcol[r] = (r == jj) ? cv : 0.5 * (cv + rv);
        if (r == jj && !isnan(xreg)) Pcol[r] += xreg;
      }
      if (!feasible) {
        const double *f = a.gaps + tb * NX;
        double vf = 0.0;
        ASLR_UNROLL for (int r = 0; r < NX; ++r) vf += Pcol[r] * f[r];
        Vx_own += vf;
        if (fddp) {
          dgf -= Vx_own * f[jj];
          dqf += f[jj] * vf;
          if (st_ok) a.vxxf[tb * NX + jj] = vf;
        }
      }
      {
        bool bad = isnan(Vx_own) || is_bad(fabs(Vx_own));
        ASLR_UNROLL for (int r = 0; r < NX; ++r) bad = bad || isnan(Pcol[r]) || is_bad(fabs(Pcol[r]));
        // team-wide OR (teams are aligned lane groups of the wave)
        const unsigned long long m = __ballot(bad);
        const unsigned long long tm = (TEAM == 64 ? ~0ull : ((1ull << TEAM) - 1ull)) << (team * TEAM);
        if (m & tm) failed = true;
      }
      if (sp.store_v && st_ok && !failed) {
        a.vx[tb * NX + jj] = Vx_own;
        double *o = a.vxx + tb * NX * NX;
        ASLR_UNROLL for (int r = 0; r < NX; ++r) o[r * NX + jj] = Pcol[r];
      }
      VxL[jj] = Vx_own;
      __syncthreads();
      ASLR_UNROLL for (int l = 0; l < NX; ++l) pvec[l] = VxL[l];
    }
    // ---- end of sweep: publish or regularise and retry ----
    if (need) {
      if (!failed) {
        if (fddp) { // reduce the per-column gap terms over the team (one contributor per column)
          if (!(col_valid && h == 0)) { dgf = 0.0; dqf = 0.0; }
          ASLR_UNROLL for (int off = TEAM / 2; off > 0; off >>= 1) {
            dgf += __shfl_xor(dgf, off);
            dqf += __shfl_xor(dqf, off);
          }
        }
        if (lt == 0) {
          TF[ASLR_TF_STOP * B + b] = stop;
          if (fddp) {
            TF[ASLR_TF_DG * B + b] = d1 + dgf;
            TF[ASLR_TF_DQ * B + b] = d2 + dqf;
          }
          TF[ASLR_TF_D1 * B + b] = d1;
          TF[ASLR_TF_D2 * B + b] = d2;
          TF[ASLR_TF_XREG * B + b] = xreg;
          TI[ASLR_TI_STATUS * B + b] = status;
        }
        need = false;
      } else {
        status |= ASLR_ST_BACKWARD_ERR;
        if (sp.standalone) {
          if (lt == 0) TI[ASLR_TI_STATUS * B + b] = status;
          need = false;
        } else {
          xreg *= sp.reg_incfactor;
          if (xreg > sp.reg_max) xreg = sp.reg_max;
          if (xreg == sp.reg_max) {
            status |= ASLR_ST_REG_MAX;
            if (lt == 0) {
              TF[ASLR_TF_XREG * B + b] = xreg;
              TI[ASLR_TI_STATUS * B + b] = status;
              TI[ASLR_TI_DONE * B + b] = 1;
            }
            need = false;
          }
        }
      }
    }
  }
}

// =================================================================================================
// forward pass + line search + solver-state update
// =================================================================================================
template <int NJ, int DAM>
__global__ void __launch_bounds__(64) forward_kernel(KArgs a, SolverDev sp) {
  constexpr int NX = 4 * NJ, NU = ModelDims<NJ, DAM>::nu;
  constexpr int TEAM = 16, TPW = 4;
  const int lane = threadIdx.x, team = lane / TEAM, al = lane % TEAM;
  const int B = a.B, T = a.T;
  const int bq = blockIdx.x * TPW + team;
  const bool team_valid = bq < B;
  const int b = team_valid ? bq : B - 1;
  int32_t *TI = a.traj_i;
  double *TF = a.traj_f;
  const int done = sp.standalone ? 0 : TI[ASLR_TI_DONE * B + b];
  const bool live = team_valid && !done;
  if (__ballot(live) == 0ull) return;
  const bool lane_on = live && al < ASLR_NALPHA;
  const int ai = al < ASLR_NALPHA ? al : ASLR_NALPHA - 1;
  const double alpha = 1.0 / (double)(1 << ai);
  const int feasible = TI[ASLR_TI_FEASIBLE * B + b];
  const bool fddp = sp.solver == ASLR_SOLVER_FDDP, box = sp.solver == ASLR_SOLVER_BOXDDP;
  const bool use_gaps = fddp && !(feasible || alpha == 1.0);
  const DevDesc &D = *a.desc;
  const double *fref = a.frame_ref ? a.frame_ref + 12 * (size_t)b : nullptr;
  const size_t TB1 = (size_t)(T + 1) * B, TB = (size_t)T * B;

  double x[NX], cost_try = 0.0, dv = 0.0;
  bool fail = false;
  {
    const double *x0 = a.x0 + (size_t)b * NX;
    ASLR_UNROLL for (int i = 0; i < NX; ++i) x[i] = x0[i];
  }
  for (int t = 0; t <= T; ++t) {
    const size_t tb = (size_t)t * B + b;
    const double *xr = a.xs + tb * NX;
    double dx[NX];
    if (use_gaps) {
      const double *f = a.gaps + tb * NX;
      ASLR_UNROLL for (int i = 0; i < NX; ++i) x[i] = x[i] + f[i] * (alpha - 1.0);
    }
    ASLR_UNROLL for (int i = 0; i < NX; ++i) dx[i] = x[i] - xr[i];
    if (fddp && !feasible) { // dv -= fs . Vxx (xs - xs_try)
      const double *vf = a.vxxf + tb * NX;
      double s = 0.0;
      ASLR_UNROLL for (int i = 0; i < NX; ++i) s += vf[i] * (xr[i] - x[i]);
      dv -= s;
    }
    if (lane_on) {
      double *o = a.xs_try + ((size_t)ai * TB1 + tb) * NX;
      ASLR_UNROLL for (int i = 0; i < NX; ++i) o[i] = x[i];
    }
    const DevModel &dm = D.models[a.node_model[t]];
    double xnext[NX], c;
    if (t < T) {
      double u[NU];
      const double *ur = a.us + tb * NU, *kr = a.kff + tb * NU, *Kr = a.kgain + tb * NU * NX;
      ASLR_UNROLL for (int i = 0; i < NU; ++i) {
        double s = ur[i] - kr[i] * alpha;
        ASLR_UNROLL for (int jx = 0; jx < NX; ++jx) s -= Kr[i * NX + jx] * dx[jx];
        u[i] = s;
      }
      if (box && dm.m.has_u_limits) {
        ASLR_UNROLL for (int i = 0; i < NU; ++i) u[i] = fmin(fmax(u[i], dm.m.u_lb[i]), dm.m.u_ub[i]);
      }
      if (lane_on) {
        double *o = a.us_try + ((size_t)ai * TB + tb) * NU;
        ASLR_UNROLL for (int i = 0; i < NU; ++i) o[i] = u[i];
      }
      knot_eval<NJ, DAM, false>(D.chain, dm, fref, x, u, xnext, c, nullptr);
      cost_try += c;
      double mx = 0.0;
      bool nan = false;
      ASLR_UNROLL for (int i = 0; i < NX; ++i) { nan = nan || isnan(xnext[i]); mx = fmax(mx, fabs(xnext[i])); }
      if (is_bad(cost_try) || nan || is_bad(mx)) fail = true;
      ASLR_UNROLL for (int i = 0; i < NX; ++i) x[i] = xnext[i];
    } else {
      knot_eval<NJ, DAM, false>(D.chain, dm, fref, x, nullptr, xnext, c, nullptr);
      cost_try += c;
      if (is_bad(cost_try)) fail = true;
    }
  }
  if (lane_on) {
    TF[(ASLR_TF_COST_TRY0 + ai) * B + b] = fail ? NAN : cost_try;
    TF[(ASLR_TF_DVTRY0 + ai) * B + b] = dv;
  }
  if (sp.standalone) return;

  // ---- line search: first acceptable alpha in Crocoddyl's order (every lane of the team agrees) ----
  const double cost0 = TF[ASLR_TF_COST * B + b];
  double d1 = TF[ASLR_TF_D1 * B + b], d2 = TF[ASLR_TF_D2 * B + b];
  const double dg = TF[ASLR_TF_DG * B + b], dq = TF[ASLR_TF_DQ * B + b];
  int accepted = -1, status = TI[ASLR_TI_STATUS * B + b];
  double dV = 0.0, dVexp = 0.0, step = 1.0, cost_acc = cost0;
  for (int s = 0; s < ASLR_NALPHA; ++s) {
    const int src = team * TEAM + s;
    const double c_s = __shfl(cost_try, src);
    const double dv_s = __shfl(dv, src);
    const int fail_s = __shfl((int)fail, src);
    if (accepted >= 0) continue;
    const double as = 1.0 / (double)(1 << s);
    step = as;
    if (fail_s) { status |= ASLR_ST_FORWARD_ERR; continue; }
    dV = cost0 - c_s;
    bool acc = false;
    if (fddp) {
      d1 = dg + dv_s;
      d2 = dq - 2.0 * dv_s;
      dVexp = as * (d1 + 0.5 * as * d2);
      if (dVexp >= 0.0) acc = (d1 < sp.th_grad) || (dV > sp.th_acceptstep * dVexp);
      else acc = (!feasible) && (dV > sp.th_acceptnegstep * dVexp);
    } else {
      dVexp = as * (d1 + 0.5 * as * d2);
      if (dVexp >= 0.0) acc = (d1 < sp.th_grad) || (!feasible) || (dV > sp.th_acceptstep * dVexp);
    }
    if (acc) { accepted = s; cost_acc = c_s; }
  }
  if (al == 0 && live) {
    int was_feasible = TI[ASLR_TI_WAS_FEASIBLE * B + b];
    int feas = feasible, fin = 0;
    double xreg = TF[ASLR_TF_XREG * B + b];
    if (accepted >= 0) {
      was_feasible = feasible;
      feas = fddp ? (was_feasible || step == 1.0) : 1;
      TF[ASLR_TF_COST * B + b] = cost_acc;
    }
    if (step > sp.th_stepdec) {
      xreg /= sp.reg_decfactor;
      if (xreg < sp.reg_min) xreg = sp.reg_min;
    }
    if (step <= sp.th_stepinc) {
      xreg *= sp.reg_incfactor;
      if (xreg > sp.reg_max) xreg = sp.reg_max;
      if (xreg == sp.reg_max) { status |= ASLR_ST_REG_MAX; fin = 1; }
    }
    const double stop = TF[ASLR_TF_STOP * B + b];
    if (!fin && !sp.fixed_iterations && was_feasible && stop < sp.th_stop) { status |= ASLR_ST_CONVERGED; fin = 1; }
    TI[ASLR_TI_ITER * B + b] += 1;
    TI[ASLR_TI_NTRIALS * B + b] += (accepted >= 0 ? accepted + 1 : ASLR_NALPHA);
    TI[ASLR_TI_STATUS * B + b] = status;
    TI[ASLR_TI_FEASIBLE * B + b] = feas;
    TI[ASLR_TI_WAS_FEASIBLE * B + b] = was_feasible;
    TI[ASLR_TI_RECALC * B + b] = accepted >= 0 ? 1 : 0;
    TI[ASLR_TI_ACCEPTED * B + b] = accepted;
    TI[ASLR_TI_DONE * B + b] = fin;
    TF[ASLR_TF_XREG * B + b] = xreg;
    TF[ASLR_TF_STEP * B + b] = step;
    TF[ASLR_TF_DV * B + b] = dV;
    TF[ASLR_TF_DVEXP * B + b] = dVexp;
    if (fddp) { TF[ASLR_TF_D1 * B + b] = d1; TF[ASLR_TF_D2 * B + b] = d2; }
  }
}

// per-trajectory solver state at solve() entry
__global__ void init_state_kernel(KArgs a, double reg0, int is_feasible) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= a.B) return;
  const int B = a.B;
  for (int r = 0; r < ASLR_TF_COUNT; ++r) a.traj_f[r * B + b] = 0.0;
  for (int r = 0; r < ASLR_TI_COUNT; ++r) a.traj_i[r * B + b] = 0;
  a.traj_f[ASLR_TF_XREG * B + b] = reg0;
  a.traj_i[ASLR_TI_FEASIBLE * B + b] = is_feasible;
  a.traj_i[ASLR_TI_RECALC * B + b] = 1;
  a.traj_i[ASLR_TI_ACCEPTED * B + b] = -1;
}

__global__ void reset_accepted_kernel(KArgs a) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < a.B) a.traj_i[ASLR_TI_ACCEPTED * a.B + b] = -1;
}

} // namespace

// =================================================================================================
// host side: handle, workspace carving, dispatch
// =================================================================================================
struct aslr_problem {
  aslr_problem_desc_t desc; // host copy (pointers nulled)
  int nj, nx, nu, dam, rec;
  char *ws;
  int64_t ws_bytes;
  aslr_region_t regions[ASLR_R_COUNT];
  KArgs k;
  int32_t *h_done; // pinned staging for count_active
};

namespace {

int64_t align_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }

int desc_dims(const aslr_problem_desc_t *d, int *nj, int *nx, int *nu, int *dam) {
  if (!d || d->B <= 0 || d->T <= 0 || d->nmodels <= 0 || d->nmodels > ASLR_MAX_MODELS) return ASLR_E_INVALID;
  if (d->chain.nj <= 0 || d->chain.nj > ASLR_MAX_NJ) return ASLR_E_INVALID;
  *nj = d->chain.nj;
  *nx = 4 * d->chain.nj;
  *nu = d->models[0].nu;
  *dam = d->models[0].dam;
  for (int i = 0; i < d->nmodels; ++i) {
    const aslr_model_t &m = d->models[i];
    if (m.nu != *nu || m.dam != *dam) return ASLR_E_INVALID; // one (nu, dam) per problem
    if (m.dam == ASLR_DAM_VSA ? m.nu != 2 * *nj : m.nu != *nj) return ASLR_E_INVALID;
    if (m.ncosts < 0 || m.ncosts > ASLR_MAX_COSTS) return ASLR_E_INVALID;
    for (int c = 0; c < m.ncosts; ++c) {
      const aslr_cost_t &ct = m.costs[c];
      if (ct.type < 0 || ct.type > ASLR_COST_STIFFNESS) return ASLR_E_INVALID;
      if (ct.type == ASLR_COST_FRAME_PLACEMENT && (ct.frame_joint < 0 || ct.frame_joint >= *nj)) return ASLR_E_INVALID;
      if (ct.type == ASLR_COST_STIFFNESS && m.dam != ASLR_DAM_VSA) return ASLR_E_INVALID;
      if (ct.type == ASLR_COST_PENDULUM && *nj < 2) return ASLR_E_INVALID;
    }
  }
  return ASLR_OK;
}

void carve(const aslr_problem_desc_t *d, int nx, int nu, aslr_region_t *r, int64_t *total) {
  const int64_t B = d->B, T = d->T, T1 = T + 1, rec = rec_len_c(nx, nu), D = sizeof(double);
  int64_t sizes[ASLR_R_COUNT];
  sizes[ASLR_R_XS] = T1 * B * nx * D;
  sizes[ASLR_R_US] = T * B * nu * D;
  sizes[ASLR_R_XNEXT] = T1 * B * nx * D;
  sizes[ASLR_R_COST] = T1 * B * D;
  sizes[ASLR_R_DERIV] = T1 * B * rec * D;
  sizes[ASLR_R_GAPS] = T1 * B * nx * D;
  sizes[ASLR_R_KGAIN] = T * B * nu * nx * D;
  sizes[ASLR_R_KFF] = T * B * nu * D;
  sizes[ASLR_R_QU] = T * B * nu * D;
  sizes[ASLR_R_VX] = T1 * B * nx * D;
  sizes[ASLR_R_VXX] = T1 * B * nx * nx * D;
  sizes[ASLR_R_XS_TRY] = (int64_t)ASLR_NALPHA * T1 * B * nx * D;
  sizes[ASLR_R_US_TRY] = (int64_t)ASLR_NALPHA * T * B * nu * D;
  sizes[ASLR_R_TRAJ_F] = (int64_t)ASLR_TF_COUNT * B * D;
  sizes[ASLR_R_TRAJ_I] = (int64_t)ASLR_TI_COUNT * B * sizeof(int32_t);
  sizes[ASLR_R_X0] = B * nx * D;
  sizes[ASLR_R_FRAME_REF] = B * 12 * D;
  sizes[ASLR_R_VXXF] = T1 * B * nx * D;
  sizes[ASLR_R_DESC] = sizeof(DevDesc);
  sizes[ASLR_R_NODE_MODEL] = T1 * sizeof(int32_t);
  int64_t off = 0;
  for (int i = 0; i < ASLR_R_COUNT; ++i) {
    r[i].offset = off;
    r[i].bytes = sizes[i];
    off += align_up(sizes[i], 256);
  }
  *total = off;
}

// Gauss-Jordan inverse of the motor inertia B (np.linalg.inv(self.B), free_fwddyn_asr.py:41)
bool invert(int n, const double *A, double *Ainv) {
  std::vector<double> a(n * 2 * n);
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) { a[i * 2 * n + j] = A[i * n + j]; a[i * 2 * n + n + j] = i == j ? 1.0 : 0.0; }
  for (int col = 0; col < n; ++col) {
    int piv = col;
    for (int r = col + 1; r < n; ++r) if (std::fabs(a[r * 2 * n + col]) > std::fabs(a[piv * 2 * n + col])) piv = r;
    if (a[piv * 2 * n + col] == 0.0) return false;
    if (piv != col) for (int j = 0; j < 2 * n; ++j) std::swap(a[col * 2 * n + j], a[piv * 2 * n + j]);
    const double d = a[col * 2 * n + col];
    for (int j = 0; j < 2 * n; ++j) a[col * 2 * n + j] /= d;
    for (int r = 0; r < n; ++r) if (r != col) {
      const double f = a[r * 2 * n + col];
      if (f != 0.0) for (int j = 0; j < 2 * n; ++j) a[r * 2 * n + j] -= f * a[col * 2 * n + j];
    }
  }
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) Ainv[i * n + j] = a[i * 2 * n + n + j];
  return true;
}

SolverDev to_dev(const aslr_solver_params_t *sp, int standalone, int store_v) {
  SolverDev s;
  s.solver = sp->solver; s.fixed_iterations = sp->fixed_iterations;
  s.th_stop = sp->th_stop; s.th_grad = sp->th_grad; s.th_gaptol = sp->th_gaptol;
  s.th_stepdec = sp->th_stepdec; s.th_stepinc = sp->th_stepinc; s.th_acceptstep = sp->th_acceptstep;
  s.th_acceptnegstep = sp->th_acceptnegstep;
  s.reg_min = sp->reg_min; s.reg_max = sp->reg_max; s.reg_incfactor = sp->reg_incfactor;
  s.reg_decfactor = sp->reg_decfactor;
  s.boxqp_maxiter = sp->boxqp_maxiter; s.boxqp_th_acceptstep = sp->boxqp_th_acceptstep;
  s.boxqp_th_grad = sp->boxqp_th_grad; s.boxqp_reg = sp->boxqp_reg;
  s.standalone = standalone; s.store_v = store_v;
  return s;
}

// ---- launch helpers, dispatching on (nj, dam) ----
#define DISPATCH_MODEL(p, CALL)                                                   \
  do {                                                                            \
    if ((p)->nj == 2 && (p)->dam == ASLR_DAM_SEA) { CALL(2, ASLR_DAM_SEA); }      \
    else if ((p)->nj == 2 && (p)->dam == ASLR_DAM_VSA) { CALL(2, ASLR_DAM_VSA); } \
    else if ((p)->nj == 7 && (p)->dam == ASLR_DAM_SEA) { CALL(7, ASLR_DAM_SEA); } \
    else { snprintf(g_err, sizeof g_err, "unsupported (nj=%d, dam=%d): built for nj=2 SEA/VSA, nj=7 SEA", (p)->nj, (p)->dam); return ASLR_E_INVALID; } \
  } while (0)

int launch_calc(aslr_problem *p, bool diff, int mode, double th_gaptol, hipStream_t st) {
  dim3 grid((p->desc.B + 63) / 64, p->desc.T + 1), block(64);
#define CALL(NJv, DAMv)                                                                              \
  if (diff) hipLaunchKernelGGL((calc_kernel<NJv, DAMv, true>), grid, block, 0, st, p->k, mode, th_gaptol); \
  else hipLaunchKernelGGL((calc_kernel<NJv, DAMv, false>), grid, block, 0, st, p->k, mode, th_gaptol)
  DISPATCH_MODEL(p, CALL);
#undef CALL
  HIP_TRY(hipGetLastError());
  return ASLR_OK;
}

template <int NX, int NU, int HS>
int launch_backward_t(aslr_problem *p, const SolverDev &sd, hipStream_t st) {
  using C = BwdCfg<NX, NU, HS>;
  const int blocks = (p->desc.B + C::TPW - 1) / C::TPW;
  const size_t lds = (size_t)C::TPW * C::LDS_TEAM * sizeof(double);
  hipLaunchKernelGGL((backward_kernel<NX, NU, HS>), dim3(blocks), dim3(64), lds, st, p->k, sd);
  HIP_TRY(hipGetLastError());
  return ASLR_OK;
}

int backward_hs(const aslr_problem *p) {
  // rows of each column split over HS lanes: wider teams when the batch cannot fill the chip
  const char *e = getenv("ASLR_BWD_HS");
  if (e) return atoi(e);
  return p->desc.B <= 8192 ? 2 : 1;
}

int launch_backward(aslr_problem *p, const SolverDev &sd, hipStream_t st) {
  const int hs = backward_hs(p);
  if (p->nx == 8 && p->nu == 2) return hs == 2 ? launch_backward_t<8, 2, 2>(p, sd, st) : launch_backward_t<8, 2, 1>(p, sd, st);
  if (p->nx == 8 && p->nu == 4) return hs == 2 ? launch_backward_t<8, 4, 2>(p, sd, st) : launch_backward_t<8, 4, 1>(p, sd, st);
  if (p->nx == 28 && p->nu == 7) return hs == 2 ? launch_backward_t<28, 7, 2>(p, sd, st) : launch_backward_t<28, 7, 1>(p, sd, st);
  snprintf(g_err, sizeof g_err, "unsupported (nx=%d, nu=%d)", p->nx, p->nu);
  return ASLR_E_INVALID;
}

int launch_forward(aslr_problem *p, const SolverDev &sd, hipStream_t st) {
  dim3 grid((p->desc.B + 3) / 4), block(64);
#define CALL(NJv, DAMv) hipLaunchKernelGGL((forward_kernel<NJv, DAMv>), grid, block, 0, st, p->k, sd)
  DISPATCH_MODEL(p, CALL);
#undef CALL
  HIP_TRY(hipGetLastError());
  return ASLR_OK;
}

} // namespace

extern "C" {

int aslr_abi_version(void) { return ASLR_ABI_VERSION; }

int64_t aslr_sizeof(int which) {
  switch (which) {
  case 0: return sizeof(aslr_chain_t);
  case 1: return sizeof(aslr_cost_t);
  case 2: return sizeof(aslr_model_t);
  case 3: return sizeof(aslr_problem_desc_t);
  case 4: return sizeof(aslr_solver_params_t);
  case 5: return sizeof(aslr_region_t);
  default: return -1;
  }
}

int32_t aslr_record_len(int32_t nx, int32_t nu) { return rec_len_c(nx, nu); }

void aslr_solver_params_default(aslr_solver_params_t *p, int32_t solver) {
  memset(p, 0, sizeof *p);
  p->solver = solver;
  p->maxiter = 100;
  p->reg_init = NAN;
  p->th_stop = 1e-9;
  p->th_grad = 1e-12;
  p->th_gaptol = 1e-16;
  p->th_stepdec = 0.5;
  p->th_stepinc = 0.01;
  p->th_acceptstep = 0.1;
  p->th_acceptnegstep = 2.0;
  p->reg_min = 1e-9;
  p->reg_max = 1e9;
  p->reg_incfactor = 10.0;
  p->reg_decfactor = 10.0;
  p->boxqp_maxiter = 100;
  p->boxqp_th_acceptstep = 0.1;
  p->boxqp_th_grad = 1e-9;
  p->boxqp_reg = 1e-9;
}

int64_t aslr_workspace_bytes(const aslr_problem_desc_t *desc) {
  int nj, nx, nu, dam;
  if (desc_dims(desc, &nj, &nx, &nu, &dam)) return ASLR_E_INVALID;
  aslr_region_t r[ASLR_R_COUNT];
  int64_t total;
  carve(desc, nx, nu, r, &total);
  return total;
}

int aslr_problem_create(const aslr_problem_desc_t *desc, void *workspace, int64_t workspace_bytes, void *stream,
                        aslr_problem_t **out) {
  if (!out) return ASLR_E_INVALID;
  *out = nullptr;
  int nj, nx, nu, dam;
  if (desc_dims(desc, &nj, &nx, &nu, &dam)) { snprintf(g_err, sizeof g_err, "invalid problem description"); return ASLR_E_INVALID; }
  if (!desc->node_model || !desc->x0) return ASLR_E_INVALID;
  for (int t = 0; t <= desc->T; ++t)
    if (desc->node_model[t] < 0 || desc->node_model[t] >= desc->nmodels) return ASLR_E_INVALID;
  if (!((nj == 2) || (nj == 7 && dam == ASLR_DAM_SEA))) {
    snprintf(g_err, sizeof g_err, "unsupported (nj=%d, dam=%d): built for nj=2 SEA/VSA, nj=7 SEA", nj, dam);
    return ASLR_E_INVALID;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { snprintf(g_err, sizeof g_err, "no HIP device"); return ASLR_E_NODEVICE; }
  aslr_problem *p = new (std::nothrow) aslr_problem();
  if (!p) return ASLR_E_INVALID;
  p->desc = *desc;
  p->desc.node_model = nullptr; p->desc.x0 = nullptr; p->desc.frame_ref = nullptr;
  p->nj = nj; p->nx = nx; p->nu = nu; p->dam = dam; p->rec = rec_len_c(nx, nu);
  int64_t total;
  carve(desc, nx, nu, p->regions, &total);
  if (!workspace || workspace_bytes < total || (reinterpret_cast<uintptr_t>(workspace) & 255u)) {
    snprintf(g_err, sizeof g_err, "workspace needs %lld bytes, 256-B aligned (got %lld)", (long long)total, (long long)workspace_bytes);
    delete p;
    return ASLR_E_WORKSPACE;
  }
  p->ws = static_cast<char *>(workspace);
  p->ws_bytes = workspace_bytes;
  hipStream_t st = static_cast<hipStream_t>(stream);
  auto reg = [&](int id) { return p->ws + p->regions[id].offset; };
  // device description (with Binv)
  DevDesc *hd = new DevDesc();
  memset(hd, 0, sizeof *hd);
  hd->chain = desc->chain;
  for (int i = 0; i < desc->nmodels; ++i) {
    hd->models[i].m = desc->models[i];
    if (!invert(nj, desc->models[i].B, hd->models[i].Binv)) {
      snprintf(g_err, sizeof g_err, "motor inertia B of model %d is singular", i);
      delete hd; delete p;
      return ASLR_E_INVALID;
    }
  }
  hipError_t e = hipMemcpyAsync(reg(ASLR_R_DESC), hd, sizeof(DevDesc), hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = hipMemcpyAsync(reg(ASLR_R_NODE_MODEL), desc->node_model, sizeof(int32_t) * (desc->T + 1), hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = hipMemcpyAsync(reg(ASLR_R_X0), desc->x0, sizeof(double) * desc->B * nx, hipMemcpyHostToDevice, st);
  if (e == hipSuccess && desc->frame_ref) e = hipMemcpyAsync(reg(ASLR_R_FRAME_REF), desc->frame_ref, sizeof(double) * desc->B * 12, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = hipMemsetAsync(reg(ASLR_R_TRAJ_I), 0, p->regions[ASLR_R_TRAJ_I].bytes, st);
  if (e == hipSuccess) e = hipMemsetAsync(reg(ASLR_R_TRAJ_F), 0, p->regions[ASLR_R_TRAJ_F].bytes, st);
  if (e == hipSuccess) e = hipMemsetAsync(reg(ASLR_R_GAPS), 0, p->regions[ASLR_R_GAPS].bytes, st);
  if (e == hipSuccess) e = hipMemsetAsync(reg(ASLR_R_KFF), 0, p->regions[ASLR_R_KFF].bytes, st);
  if (e == hipSuccess) e = hipMemsetAsync(reg(ASLR_R_VXXF), 0, p->regions[ASLR_R_VXXF].bytes, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st); // the host staging buffers die below
  delete hd;
  if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void **>(&p->h_done), sizeof(int32_t) * desc->B, 0);
  if (e != hipSuccess) {
    snprintf(g_err, sizeof g_err, "upload failed: %s", hipGetErrorString(e));
    delete p;
    return ASLR_E_HIP;
  }
  KArgs &k = p->k;
  k.desc = reinterpret_cast<const DevDesc *>(reg(ASLR_R_DESC));
  k.node_model = reinterpret_cast<const int32_t *>(reg(ASLR_R_NODE_MODEL));
  k.x0 = reinterpret_cast<const double *>(reg(ASLR_R_X0));
  k.frame_ref = desc->frame_ref ? reinterpret_cast<const double *>(reg(ASLR_R_FRAME_REF)) : nullptr;
  k.xs = (double *)reg(ASLR_R_XS); k.us = (double *)reg(ASLR_R_US); k.xnext = (double *)reg(ASLR_R_XNEXT);
  k.cost = (double *)reg(ASLR_R_COST); k.deriv = (double *)reg(ASLR_R_DERIV); k.gaps = (double *)reg(ASLR_R_GAPS);
  k.kgain = (double *)reg(ASLR_R_KGAIN); k.kff = (double *)reg(ASLR_R_KFF); k.qu = (double *)reg(ASLR_R_QU);
  k.vx = (double *)reg(ASLR_R_VX); k.vxx = (double *)reg(ASLR_R_VXX); k.xs_try = (double *)reg(ASLR_R_XS_TRY);
  k.us_try = (double *)reg(ASLR_R_US_TRY); k.vxxf = (double *)reg(ASLR_R_VXXF);
  k.traj_f = (double *)reg(ASLR_R_TRAJ_F); k.traj_i = (int32_t *)reg(ASLR_R_TRAJ_I);
  k.B = desc->B; k.T = desc->T;
  *out = p;
  return ASLR_OK;
}

int aslr_problem_destroy(aslr_problem_t *p) {
  if (!p) return ASLR_OK;
  if (p->h_done) (void)hipHostFree(p->h_done);
  delete p;
  return ASLR_OK;
}

int aslr_problem_region(const aslr_problem_t *p, int32_t region_id, aslr_region_t *out) {
  if (!p || !out || region_id < 0 || region_id >= ASLR_R_COUNT) return ASLR_E_INVALID;
  *out = p->regions[region_id];
  return ASLR_OK;
}

int aslr_calc(aslr_problem_t *p, void *stream) {
  if (!p) return ASLR_E_INVALID;
  return launch_calc(p, false, 0, -1.0, static_cast<hipStream_t>(stream));
}

int aslr_calc_diff(aslr_problem_t *p, void *stream) {
  if (!p) return ASLR_E_INVALID;
  return launch_calc(p, true, 0, -1.0, static_cast<hipStream_t>(stream));
}

int aslr_backward_pass(aslr_problem_t *p, const aslr_solver_params_t *sp, void *stream) {
  if (!p || !sp) return ASLR_E_INVALID;
  return launch_backward(p, to_dev(sp, 1, 1), static_cast<hipStream_t>(stream));
}

int aslr_forward_pass(aslr_problem_t *p, const aslr_solver_params_t *sp, void *stream) {
  if (!p || !sp) return ASLR_E_INVALID;
  return launch_forward(p, to_dev(sp, 1, 0), static_cast<hipStream_t>(stream));
}

int aslr_iterate(aslr_problem_t *p, const aslr_solver_params_t *sp, int32_t first, void *stream) {
  if (!p || !sp) return ASLR_E_INVALID;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (first) {
    const double reg0 = std::isnan(sp->reg_init) ? sp->reg_min : sp->reg_init;
    hipLaunchKernelGGL(init_state_kernel, dim3((p->desc.B + 255) / 256), dim3(256), 0, st, p->k, reg0, sp->is_feasible);
    HIP_TRY(hipGetLastError());
  }
  const SolverDev sd = to_dev(sp, 0, 0);
  int rc = launch_calc(p, true, kModeCommit | kModeSolver, sp->th_gaptol, st);
  if (rc) return rc;
  rc = launch_backward(p, sd, st);
  if (rc) return rc;
  return launch_forward(p, sd, st);
}

int aslr_finalize(aslr_problem_t *p, void *stream) {
  if (!p) return ASLR_E_INVALID;
  hipStream_t st = static_cast<hipStream_t>(stream);
  int rc = launch_calc(p, false, kModeCommit | kModeNoCompute, -1.0, st);
  if (rc) return rc;
  hipLaunchKernelGGL(reset_accepted_kernel, dim3((p->desc.B + 255) / 256), dim3(256), 0, st, p->k);
  HIP_TRY(hipGetLastError());
  return ASLR_OK;
}

int aslr_count_active(aslr_problem_t *p, void *stream, int32_t *active) {
  if (!p || !active) return ASLR_E_INVALID;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int B = p->desc.B;
  HIP_TRY(hipMemcpyAsync(p->h_done, p->k.traj_i + (size_t)ASLR_TI_DONE * B, sizeof(int32_t) * B, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  int n = 0;
  for (int b = 0; b < B; ++b) n += p->h_done[b] ? 0 : 1;
  *active = n;
  return ASLR_OK;
}

int aslr_solve(aslr_problem_t *p, const aslr_solver_params_t *sp, int32_t poll_every, void *stream, int32_t *iters_done) {
  if (!p || !sp) return ASLR_E_INVALID;
  int it = 0;
  for (; it < sp->maxiter; ++it) {
    int rc = aslr_iterate(p, sp, it == 0, stream);
    if (rc) return rc;
    if (!sp->fixed_iterations && poll_every > 0 && (it + 1) % poll_every == 0 && it + 1 < sp->maxiter) {
      int32_t active = 0;
      rc = aslr_count_active(p, stream, &active);
      if (rc) return rc;
      if (active == 0) { ++it; break; }
    }
  }
  if (iters_done) *iters_done = it;
  return aslr_finalize(p, stream);
}

int aslr_dam_eval(aslr_problem_t *p, int32_t model_index, int32_t n, const double *x, const double *u, double *xout,
                  double *cost, double *Fx, double *Fu, double *Lx, double *Lu, double *Lxx, double *Lxu, double *Luu,
                  void *stream) {
  if (!p || n <= 0 || model_index < 0 || model_index >= p->desc.nmodels || !x || !u) return ASLR_E_INVALID;
  hipStream_t st = static_cast<hipStream_t>(stream);
  dim3 grid((n + 63) / 64), block(64);
#define CALL(NJv, DAMv)                                                                                         \
  hipLaunchKernelGGL((dam_eval_kernel<NJv, DAMv>), grid, block, 0, st, p->k.desc, model_index, p->k.frame_ref, n, \
                     x, u, xout, cost, Fx, Fu, Lx, Lu, Lxx, Lxu, Luu)
  DISPATCH_MODEL(p, CALL);
#undef CALL
  HIP_TRY(hipGetLastError());
  return ASLR_OK;
}

const char *aslr_last_error(void) { return g_err; }

} // extern "C"
