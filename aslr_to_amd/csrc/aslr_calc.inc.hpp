// aslr_calc.inc.hpp -- calc / calcDiff / dam_eval kernels (design: DESIGN.md section 4.1)
#pragma once
#include "aslr_common.hpp"

#ifndef ASLR_CALC_WAVES
#define ASLR_CALC_WAVES 1 // waves per SIMD the register allocator must allow: the kernel is bound by the record
                          // write (734 MB per sweep at C3), 1 / 2 / 3 waves measured 171 / 177 / 244 us
#endif

namespace aslr {
// =================================================================================================
// calc / calcDiff
// =================================================================================================
constexpr int kChunk = 16;            // doubles per knot per LDS flush (one 128-B line)
constexpr int kLdsStride = kChunk + 1; // odd stride: conflict-free ds_write_b64 across lanes


// PRE: the rigid-body part of the knot was computed by dyn_team_kernel (aslr_calc_team.inc.hpp) into DYN
// SKIPC: the model-only record chunks are in place (kModeSkipConst, known at launch): compiled out, and with them the
// parts of the compact derivative set only they read (the steady-state sweeps of a solve run this variant)
template <int NJ, int DAM, bool DIFF, bool PLANAR, bool PRE = false, bool SKIPC = false>
__global__ void __launch_bounds__(64, ASLR_CALC_WAVES) calc_kernel(KArgs a, int mode, double th_gaptol) {
  constexpr int NX = 4 * NJ, NU = ModelDims<NJ, DAM>::nu;
  using RL = RecLayout<NJ, NU>;
  constexpr int REC = RL::len;
  __shared__ double sm[64 * kLdsStride];
  __shared__ int act[64];
  // PRE (large chain): each lane's nj x nj block of Lxx is accumulated here, not in registers (odd stride: the lanes of a
  // wave hit different banks)
  __shared__ double lqqL[PRE ? 64 * (NJ * NJ) : 1];

  const int lane = threadIdx.x, t = blockIdx.y, B = a.B, T = a.T;
  ASLR_STAMP_BEGIN(a, 0);
  const int b0 = a.b0 + blockIdx.x * 64, bq = b0 + lane; // (b0: first trajectory of this block)
  const bool valid = bq < a.b1;
  const int b = valid ? bq : a.b1 - 1;
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
  ASLR_PROF_DECL;

  // ---- x, u of this knot (from the accepted candidate when there is one) ----
  double x[NX], u[NU];
  {
    if (acc >= 0 && ASLR_CAND_INTERLEAVED(NX)) {
      ASLR_UNROLL for (int p = 0; p < NX / 2; ++p) {
        const double2 v = *reinterpret_cast<const double2 *>(a.xs_try + cand_piece<NX>(acc, t, b, p, B, T + 1));
        x[2 * p] = v.x; x[2 * p + 1] = v.y;
      }
    } else {
      const double *src = acc >= 0 ? a.xs_try + ((size_t)acc * TB1 + tb) * NX : a.xs + tb * NX;
      ASLR_UNROLL for (int i = 0; i < NX; ++i) x[i] = src[i];
    }
    if (acc >= 0 && valid) {
      double *dst = a.xs + tb * NX;
      ASLR_UNROLL for (int i = 0; i < NX; ++i) dst[i] = x[i];
    }
  }
  if (t < T) {
    if (acc >= 0 && ASLR_CAND_INTERLEAVED(NU)) {
      ASLR_UNROLL for (int p = 0; p < NU / 2; ++p) {
        const double2 v = *reinterpret_cast<const double2 *>(a.us_try + cand_piece<NU>(acc, t, b, p, B, T));
        u[2 * p] = v.x; u[2 * p + 1] = v.y;
      }
    } else {
      const double *src = acc >= 0 ? a.us_try + ((size_t)acc * TB + tb) * NU : a.us + tb * NU;
      ASLR_UNROLL for (int i = 0; i < NU; ++i) u[i] = src[i];
    }
    if (acc >= 0 && valid) {
      double *dst = a.us + tb * NU;
      ASLR_UNROLL for (int i = 0; i < NU; ++i) dst[i] = u[i];
    }
  } else {
    ASLR_UNROLL for (int i = 0; i < NU; ++i) u[i] = 0.0;
  }
  const bool compute = valid && recalc && !done && !(mode & kModeNoCompute);
  if (__ballot(compute) == 0ull) return; // wave-uniform
#ifdef ASLR_BWD_PROFILE
  { double wsum = 0.0; ASLR_UNROLL for (int i = 0; i < NX; ++i) wsum += x[i]; ASLR_UNROLL for (int i = 0; i < NU; ++i) wsum += u[i];
    asm volatile("" : : "v"(wsum)); } // (profile build: the inputs have arrived)
#endif
  ASLR_PROF(0);

  const DevDesc &D = *a.desc;
  const DevModel &dm = D.models[node_model_at(a, t)];
  const double *fref = a.frame_ref ? a.frame_ref + 12 * (size_t)b : nullptr;

  double xnext[NX], cost = 0.0;
  KnotDiff<NJ, NU> kd;
  if (compute) {
    using CH = std::conditional_t<PLANAR, ChainPlanar<NJ>, Chain3D<NJ>>;
    const typename CH::Consts cc(D);
    ModelRegs<NJ, NU> mr;
    mr.load(dm);
    // PRE (large chain): the evaluation is split -- dynamics derivatives first, their part of the record streamed
    // out, then the cost stack -- so that the ~400 doubles of the compact derivative set are never all live
    constexpr int what = (DIFF ? kEvalDiff : (kEvalDyn | kEvalCost)) | (PRE ? (kEvalPre | kEvalSkipCost) : 0);
    knot_eval<NJ, DAM, what, CH>(cc, mr, dm, fref, x, t < T ? u : nullptr, xnext, cost, DIFF ? &kd : nullptr, nullptr,
                                 PRE ? a.dyn + tb * dyn_len_c(NJ) : nullptr);
    ASLR_PROF(1);
    double *xn = a.xnext + tb * NX;
    ASLR_UNROLL for (int i = 0; i < NX; ++i) xn[i] = xnext[i];
    if constexpr (!PRE) a.cost[tb] = cost;
    // gaps (SolverDDP::calcDiff): f[0] = x0 - xs[0]; f[t+1] = xnext_t - xs[t+1]
    if ((mode & kModeSolver) && !feasible) {
      double mx = 0.0;
      if (t < T) {
        const size_t tb1 = tb + B;
        double nxt[NX]; // the state the next knot starts from
        if (acc >= 0 && ASLR_CAND_INTERLEAVED(NX)) {
          ASLR_UNROLL for (int p = 0; p < NX / 2; ++p) {
            const double2 v = *reinterpret_cast<const double2 *>(a.xs_try + cand_piece<NX>(acc, t + 1, b, p, B, T + 1));
            nxt[2 * p] = v.x; nxt[2 * p + 1] = v.y;
          }
        } else {
          const double *src = acc >= 0 ? a.xs_try + ((size_t)acc * TB1 + tb1) * NX : a.xs + tb1 * NX;
          ASLR_UNROLL for (int i = 0; i < NX; ++i) nxt[i] = src[i];
        }
        double *g = a.gaps + tb1 * NX;
        ASLR_UNROLL for (int i = 0; i < NX; ++i) {
          const double f = xnext[i] - nxt[i];
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
  const bool all_on = __ballot(compute) == ~0ull; // the usual case: no per-record test in the store loop
  const double dt = dm.m.dt;
  double *rec0 = a.deriv + ((size_t)t * B + b0) * REC;
  auto flush_chunk = [&](auto cc) {
    constexpr int c = decltype(cc)::value;
    // chunks of structural zeros (all of Lxu, most of Lxx / Luu for the larger chain) were written once, by the
    // zero fill of DERIV at problem creation
    if constexpr (rec_chunk_is_zero<NJ, NU, c, kChunk>()) return;
    // chunks that depend on the model only (cost-weight diagonals of Lxx / Luu; all of Fu for SEA) are written by the
    // first full sweep
    if constexpr (rec_chunk_is_model_only<NJ, NU, c, kChunk, DAM == ASLR_DAM_SEA>()) {
      if constexpr (SKIPC) return;
    }
    if (compute) {
      static_for<0, kChunk>([&](auto ii) {
        constexpr int i = decltype(ii)::value;
        sm[lane * kLdsStride + i] = rec_elem<NJ, NU, c * kChunk + i, PRE>(kd, dt);
      });
    }
    wave_sync();
    ASLR_UNROLL for (int i = 0; i < kChunk / 2; ++i) {
      const int idx = lane + 64 * i, k = idx >> 3, e = (idx & 7) * 2;
      if (all_on || act[k]) {
        double2 v2;
        v2.x = sm[k * kLdsStride + e];
        v2.y = sm[k * kLdsStride + e + 1];
        // streamed once, read once by the backward sweep: non-temporal
        typedef double nt_double2 __attribute__((ext_vector_type(2)));
        nt_double2 nv;
        nv.x = v2.x; nv.y = v2.y;
        __builtin_nontemporal_store(nv, reinterpret_cast<nt_double2 *>(rec0 + (size_t)k * REC + c * kChunk + e));
      }
    }
    wave_sync();
  };
  // chunks made of Fx / Fu only come first; with PRE the cost stack is evaluated between the two groups
  constexpr int C1 = PRE ? RL::oLxx / kChunk : 0;
  static_for<0, C1>(flush_chunk);
  if constexpr (PRE) {
    if (compute) {
      using CH = std::conditional_t<PLANAR, ChainPlanar<NJ>, Chain3D<NJ>>;
      const typename CH::Consts cc(D);
      ModelRegs<NJ, NU> mr;
      mr.load(dm);
      double xn2[NX], cost2 = 0.0;
      // x and u are handed over as pointers into the committed trajectory (this lane wrote or read them above), not as the
      // register copies: the cost terms load the entries they use where they use them, and the 70 registers are free
      // while the frame-placement Jacobian and Hessian are formed
      kd.lqq_mem = lqqL + lane * (NJ * NJ);
      knot_eval<NJ, DAM, kEvalDiff | kEvalSkipDyn | kEvalLqqMem, CH>(cc, mr, dm, fref, *reinterpret_cast<const double (*)[NX]>(a.xs + tb * NX),
                                                       t < T ? a.us + tb * NU : nullptr, xn2, cost2, &kd);
      a.cost[(size_t)t * B + b] = cost2;
    }
  }
  static_for<C1, REC / kChunk>(flush_chunk);
  ASLR_PROF(2);
  ASLR_PROF_COUNT(15);
  ASLR_PROF_FLUSH;
  }
  ASLR_STAMP_END(0, false);
}

// ShootingProblem.quasiStatic (examples/two_dof_sea.py:78): Crocoddyl's base-class
// Gauss-Newton per running node, u = 0; repeat { calc, calcDiff; du = -pinv(Fu) (xnext - x); u += du }
// until |du| <= tol or maxiter.  One lane per (trajectory, node).  pinv(Fu) acts through the normal equations
// and a thresholded eigen-decomposition (pinv_normal_solve), so a rank-deficient Fu -- VSA at q_l = q_m, where
// the stiffness columns vanish -- gets the minimum-norm update Crocoddyl's SVD pseudo-inverse gives.
template <int NJ, int DAM, bool PLANAR>
__global__ void __launch_bounds__(64) quasi_static_kernel(KArgs a, int maxiter, double tol, int32_t *iters_out) {
  constexpr int NX = 4 * NJ, NU = ModelDims<NJ, DAM>::nu, NV = 2 * NJ;
  using CH = std::conditional_t<PLANAR, ChainPlanar<NJ>, Chain3D<NJ>>;
  const int B = a.B, T = a.T, t = blockIdx.y;
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= B || t >= T) return;
  const size_t tb = (size_t)t * B + b;
  const DevDesc &D = *a.desc;
  const DevModel &dm = D.models[node_model_at(a, t)];
  const double *fref = a.frame_ref ? a.frame_ref + 12 * (size_t)b : nullptr;
  const typename CH::Consts cc(D);
  ModelRegs<NJ, NU> mr;
  mr.load(dm);
  double x[NX], u[NU];
  ASLR_UNROLL for (int i = 0; i < NX; ++i) x[i] = a.xs[tb * NX + i];
  ASLR_UNROLL for (int i = 0; i < NU; ++i) u[i] = 0.0;
  int it = 0;
  for (; it < maxiter; ++it) {
    double xnext[NX], c;
    KnotDiff<NJ, NU> kd;
    knot_eval<NJ, DAM, kEvalDiff, CH>(cc, mr, dm, fref, x, u, xnext, c, &kd);
    // Fu = dt [dt A_u ; A_u] (integrated_action.py:36-37); rows: positions (link, motor), velocities
    const double dt = mr.dt;
    double A[NU][NU], rhs[NU];
    ASLR_UNROLL for (int i = 0; i < NU; ++i) {
      double r = 0.0;
      ASLR_UNROLL for (int l = 0; l < NV; ++l) {
        const double au = l < NJ ? kd.Ful[l < NJ ? l : 0][i] : kd.Fum[l >= NJ ? l - NJ : 0][i];
        const double fp = dt * (au * dt), fv = dt * au;
        r += fp * (xnext[l] - x[l]) + fv * (xnext[NV + l] - x[NV + l]);
      }
      rhs[i] = -r;
      ASLR_UNROLL for (int j = 0; j < NU; ++j) {
        double s2 = 0.0;
        ASLR_UNROLL for (int l = 0; l < NV; ++l) {
          const double ai = l < NJ ? kd.Ful[l < NJ ? l : 0][i] : kd.Fum[l >= NJ ? l - NJ : 0][i];
          const double aj = l < NJ ? kd.Ful[l < NJ ? l : 0][j] : kd.Fum[l >= NJ ? l - NJ : 0][j];
          s2 += (dt * (ai * dt)) * (dt * (aj * dt)) + (dt * ai) * (dt * aj);
        }
        A[i][j] = s2;
      }
    }
    {
      double g[NU];
      ASLR_UNROLL for (int i = 0; i < NU; ++i) g[i] = rhs[i];
      pinv_normal_solve<NU>(NX, A, g, rhs);
    }
    double nrm = 0.0;
    ASLR_UNROLL for (int i = 0; i < NU; ++i) { u[i] += rhs[i]; nrm += rhs[i] * rhs[i]; }
    if (sqrt(nrm) <= tol) break;
  }
  ASLR_UNROLL for (int i = 0; i < NU; ++i) a.us[tb * NU + i] = u[i];
  if (iters_out) iters_out[tb] = it;
}

// DAM-level evaluation of arbitrary points (aslr_dam_eval): dense continuous blocks, one lane per point
template <int NJ, int DAM, bool PLANAR>
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
  using CH = std::conditional_t<PLANAR, ChainPlanar<NJ>, Chain3D<NJ>>;
  const typename CH::Consts cc(D);
  ModelRegs<NJ, NU> mr;
  mr.load(dm);
  knot_eval<NJ, DAM, kEvalDiff, CH>(cc, mr, dm, frame_ref, x, u, xnext, c, &kd, xo);
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

// aslr_dam_residuals: the stacked cost residuals (data.r) of arbitrary points, one lane per point
template <int NJ, int DAM, bool PLANAR>
__global__ void __launch_bounds__(64) dam_residual_kernel(const DevDesc *desc, int mi, const double *frame_ref, int n,
                                                          const double *xin, const double *uin, double *r, int nr) {
  constexpr int NX = 4 * NJ, NU = ModelDims<NJ, DAM>::nu;
  const int p = blockIdx.x * 64 + threadIdx.x;
  if (p >= n) return;
  const DevDesc &D = *desc;
  const DevModel &dm = D.models[mi];
  double x[NX], u[NU], xnext[NX], c;
  ASLR_UNROLL for (int i = 0; i < NX; ++i) x[i] = xin[(size_t)p * NX + i];
  ASLR_UNROLL for (int i = 0; i < NU; ++i) u[i] = uin[(size_t)p * NU + i];
  using CH = std::conditional_t<PLANAR, ChainPlanar<NJ>, Chain3D<NJ>>;
  const typename CH::Consts cc(D);
  ModelRegs<NJ, NU> mr;
  mr.load(dm);
  knot_eval<NJ, DAM, kEvalCost | kEvalResid, CH>(cc, mr, dm, frame_ref, x, u, xnext, c, nullptr, nullptr, nullptr,
                                                 r + (size_t)p * nr);
}

// aslr_frame_placement: oMf = oMi[fj] * F at the link positions of n points (data.pinocchio.oMf of the scripts)
template <int NJ, bool PLANAR>
__global__ void __launch_bounds__(64) frame_placement_kernel(const DevDesc *desc, int fj, FrameArg F, int n,
                                                             const double *xin, long long stride, double *out) {
  const int p = blockIdx.x * 64 + threadIdx.x;
  if (p >= n) return;
  using CH = std::conditional_t<PLANAR, ChainPlanar<NJ>, Chain3D<NJ>>;
  const typename CH::Consts cc(*desc);
  double q[NJ];
  ASLR_UNROLL for (int i = 0; i < NJ; ++i) q[i] = xin[(size_t)p * stride + i];
  const SE3d oMj = CH::world_of(cc, q, fj);
  SE3d Fl;
  ASLR_UNROLL for (int i = 0; i < 9; ++i) Fl.R.a[i] = F.R[i];
  Fl.p = V3{F.p[0], F.p[1], F.p[2]};
  const SE3d oMf = se3_mul(oMj, Fl);
  double *o = out + (size_t)p * 12;
  ASLR_UNROLL for (int i = 0; i < 9; ++i) o[i] = oMf.R.a[i];
  o[9] = oMf.p.x; o[10] = oMf.p.y; o[11] = oMf.p.z;
}

template <int NJ>
int launch_frame_placement_t(const KArgs &k, int fj, const FrameArg &F, int n, const double *x, int64_t stride, double *out,
                             hipStream_t st) {
  dim3 grid((n + 63) / 64), block(64);
  if (k.planar) hipLaunchKernelGGL((frame_placement_kernel<NJ, true>), grid, block, 0, st, k.desc, fj, F, n, x, (long long)stride, out);
  else hipLaunchKernelGGL((frame_placement_kernel<NJ, false>), grid, block, 0, st, k.desc, fj, F, n, x, (long long)stride, out);
  HIP_TRY(hipGetLastError());
  return ASLR_OK;
}

template <int NJ, int DAM>
int launch_dam_residuals_t(const KArgs &k, int mi, int n, const double *x, const double *u, double *r, int nr, hipStream_t st) {
  dim3 grid((n + 63) / 64), block(64);
  if (k.planar) hipLaunchKernelGGL((dam_residual_kernel<NJ, DAM, true>), grid, block, 0, st, k.desc, mi, k.frame_ref, n, x, u, r, nr);
  else hipLaunchKernelGGL((dam_residual_kernel<NJ, DAM, false>), grid, block, 0, st, k.desc, mi, k.frame_ref, n, x, u, r, nr);
  HIP_TRY(hipGetLastError());
  return ASLR_OK;
}

} // namespace aslr
