// aslr_forward.inc.hpp -- forward pass of Solver{DDP,FDDP,BoxDDP} as three kernels:
//
//   rollout_kernel     16-lane team per trajectory, lane a rolls out step length 2^-a: ONLY the serial
//                      part stays on the T-step dependency chain (control law u = us - alpha k - K dx,
//                      box clamp, dynamics, Euler step); every lane stores its candidate XS_TRY[a], US_TRY[a];
//                      the inputs shared by the step lengths arrive per team through LDS-DMA, one knot ahead.
//   trial_cost_kernel  one lane per (alpha, knot, trajectory): the cost stack on the stored candidates,
//                      embarrassingly parallel (the frame-placement log map is half of a knot evaluation
//                      and does not feed the state recursion).
//   select_kernel      one lane per trajectory: sums the node costs of each candidate in rollout order,
//                      takes the FIRST acceptable alpha in Crocoddyl's order and updates the solver state
//                      (feasibility, cost, regularisation schedule, stop, status).
//
// The rollout and the trial costs take a SEGMENT of the horizon (KArgs::seg_t0 / seg_t1), and rollout_and_cost_kernel is one
// launch with both roles: its first blocks roll out a segment, the others evaluate the trial costs of the segment before.
// The launchers (aslr_forward_nj2.hip) use it to take half of the trial costs off the latency chain of an iteration.
#pragma once
#include "aslr_common.hpp"

#ifndef ASLR_ROLLOUT_TPW
#define ASLR_ROLLOUT_TPW 4 // trajectories (16-lane teams) per rollout wave: 4 fills the wave; 2 leaves half of it idle but puts
                           // two waves on every SIMD at 4096 trajectories, which hide each other's latency
#endif

namespace aslr {

// FDDP: the gap-contracting rollout and the dv terms of SolverFDDP are compiled in (two more 8-double prefetch
// buffers per lane); the DDP / BoxDDP variant does not carry them.
// Knots [s0, s1] of the horizon (the whole of it: [0, T]): a launch that starts at s0 > 0 continues the rollout a previous
// launch stopped at s0 -- state from the candidate stored there, dv / failure flag from the per-trajectory slots -- and
// repeats nothing that launch has already accounted for at knot s0 (gap contraction, the dv term).
template <int NJ, int DAM, bool PLANAR, bool FDDP>
ASLR_DEV void rollout_body(const KArgs &a, const SolverDev &sp, const ModelLimits &lim, int vblock, int s0, int s1) {
  constexpr int NX = 4 * NJ, NU = ModelDims<NJ, DAM>::nu;
  constexpr int TEAM = 16, TPW = ASLR_ROLLOUT_TPW;
  using CH = std::conditional_t<PLANAR, ChainPlanar<NJ>, Chain3D<NJ>>;
  const int lane = threadIdx.x, team = lane / TEAM, al = lane % TEAM;
  const int B = a.B, T = a.T;
  const int bq = a.b0 + vblock * TPW + team;
  const bool team_valid = team < TPW && bq < a.b1;
  const int b = team_valid ? bq : a.b1 - 1;
  int32_t *TI = a.traj_i;
  double *TF = a.traj_f;
  const int done = sp.standalone ? 0 : TI[ASLR_TI_DONE * B + b];
  const bool live = team_valid && !done;
  if (__ballot(live) == 0ull) return;
  const bool lane_on = live && al < ASLR_NALPHA;
  const int ai = al < ASLR_NALPHA ? al : ASLR_NALPHA - 1;
  const double alpha = 1.0 / (double)(1 << ai);
  const int feasible = TI[ASLR_TI_FEASIBLE * B + b];
  const bool fddp = FDDP && sp.solver == ASLR_SOLVER_FDDP, box = sp.solver == ASLR_SOLVER_BOXDDP;
  const bool use_gaps = fddp && !(feasible || alpha == 1.0);
  const bool need_dv = fddp && !feasible;
  const DevDesc &D = *a.desc;

  double x[NX], dv = 0.0;
  bool fail = false;
  const bool cont = s0 > 0;
  if (!cont) {
    const double *x0 = a.x0 + (size_t)b * NX;
    ASLR_UNROLL for (int i = 0; i < NX; ++i) x[i] = x0[i];
  } else {
    ASLR_UNROLL for (int p = 0; p < NX / 2; ++p) {
      const double2 v = *reinterpret_cast<const double2 *>(a.xs_try + cand_piece<NX>(ai, s0, b, p, B, T + 1));
      x[2 * p] = v.x; x[2 * p + 1] = v.y;
    }
    dv = TF[(ASLR_TF_DVTRY0 + ai) * B + b];
    fail = TI[(ASLR_TI_TRYFAIL0 + ai) * B + b] != 0;
  }
  const typename CH::Consts cc(D, true); // (true: this kernel loops over knots -- sin / cos constants in registers)
  ModelRegs<NJ, NU> mr;
  int m_loaded = -1, lim_has = 0;
  double lim_lb[NU], lim_ub[NU]; // control limits of the loaded model (kernel arguments behind a runtime index: scalar loads)
  ASLR_UNROLL for (int i = 0; i < NU; ++i) { lim_lb[i] = 0.0; lim_ub[i] = 0.0; }
  // Per-knot inputs shared by the step lengths of a trajectory -- [K | xs | us | k | gaps | Vxx f] -- are fetched ONCE
  // per team, one knot ahead, straight into LDS (global_load_lds_dwordx4: lane lt fetches the 16-byte pieces lt and
  // lt + 16 of the list; element e of team tm lands at (e / 32) * 128 + 32 tm + e % 32 of the parity buffer).
  // The wait for them is an explicit s_waitcnt vmcnt(NST): the candidate stores of the previous knot were issued AFTER
  // these loads and may stay in flight -- a compiler-placed wait on prefetch registers was vmcnt(0), i.e. every knot
  // also waited for its own candidate stores to reach memory.
  constexpr int oK = 0, oXr = oK + NU * NX, oU = oXr + NX, oKf = oU + NU, oFg = oKf + NU, oVf = oFg + NX,
                NE = FDDP ? oVf + NX : oFg, NPI = (NE / 2 + TEAM - 1) / TEAM, BS = 2 * TEAM, DMAW = 128;
  static_assert(NU % 2 == 0 && NX % 2 == 0, "16-byte pieces");
  __shared__ __attribute__((aligned(16))) double stgD[2][NPI * DMAW];
  constexpr int NST = NX / 2 + NU / 2; // a lower bound of the store instructions per knot (16 bytes each at most)
  const char *dsrc[NPI];
  size_t dstr[NPI];
  bool don[NPI], dctl[NPI];
  ASLR_UNROLL for (int q = 0; q < NPI; ++q) {
    const int e = 2 * (al + TEAM * q); // first element of this lane's piece
    dsrc[q] = reinterpret_cast<const char *>(a.xs); dstr[q] = 0; don[q] = false; dctl[q] = false;
    if (e < oXr) { dsrc[q] = reinterpret_cast<const char *>(a.kgain + (size_t)b * NU * NX + e); dstr[q] = (size_t)B * NU * NX * 8; don[q] = true; dctl[q] = true; }
    else if (e < oU) { dsrc[q] = reinterpret_cast<const char *>(a.xs + (size_t)b * NX + (e - oXr)); dstr[q] = (size_t)B * NX * 8; don[q] = true; }
    else if (e < oKf) { dsrc[q] = reinterpret_cast<const char *>(a.us + (size_t)b * NU + (e - oU)); dstr[q] = (size_t)B * NU * 8; don[q] = true; dctl[q] = true; }
    else if (e < oFg) { dsrc[q] = reinterpret_cast<const char *>(a.kff + (size_t)b * NU + (e - oKf)); dstr[q] = (size_t)B * NU * 8; don[q] = true; dctl[q] = true; }
    else if (FDDP && e < oVf) { dsrc[q] = reinterpret_cast<const char *>(a.gaps + (size_t)b * NX + (e - oFg)); dstr[q] = (size_t)B * NX * 8; don[q] = need_dv; }
    else if (FDDP && e < oVf + NX) { dsrc[q] = reinterpret_cast<const char *>(a.vxxf + (size_t)b * NX + (e - oVf)); dstr[q] = (size_t)B * NX * 8; don[q] = need_dv; }
  }
  int mi_next = 0;
  auto prefetch = [&](int t) {
    const unsigned base = lds_address(stgD[t & 1]);
    ASLR_UNROLL for (int q = 0; q < NPI; ++q) {
      if (don[q] && (!dctl[q] || t < T)) dma16<0, false>(dsrc[q] + (size_t)t * dstr[q], base + q * DMAW * 8);
    }
    mi_next = node_model_at(a, t);
  };
  prefetch(s0);
  for (int t = s0; t <= s1; ++t) {
    // inputs of knot t have landed (issued one knot ago, before that knot's NST candidate stores)
    if (t == s0) wait_vmcnt<0>(); else wait_vmcnt<NST>();
    const bool seam = cont && t == s0; // the previous launch has been here
    wave_sync();
    const double *stgT = stgD[t & 1] + team * BS;
    auto S = [&](int e) -> double { return stgT[(e / BS) * DMAW + e % BS]; };
    const int mi = mi_next;
    if (t < s1) prefetch(t + 1); // in flight while knot t computes (the other parity buffer: its readers finished
                                 // before the wave_sync above)
    double dx[NX];
    if (use_gaps && !seam) { ASLR_UNROLL for (int i = 0; i < NX; ++i) x[i] = x[i] + S(oFg + i) * (alpha - 1.0); }
    ASLR_UNROLL for (int i = 0; i < NX; ++i) dx[i] = x[i] - S(oXr + i);
    if (need_dv && !seam) { // dv -= fs . Vxx (xs - xs_try)
      double s = 0.0;
      ASLR_UNROLL for (int i = 0; i < NX; ++i) s += S(oVf + i) * (S(oXr + i) - x[i]);
      dv -= s;
    }
    if (lane_on) { // (piece-interleaved slab: the 4 teams of the wave write neighbouring 16-byte pieces)
      ASLR_UNROLL for (int p = 0; p < NX / 2; ++p)
        *reinterpret_cast<double2 *>(a.xs_try + cand_piece<NX>(ai, t, b, p, B, T + 1)) = make_double2(x[2 * p], x[2 * p + 1]);
    }
    if (t == s1) break; // (the state of knot s1 is stored: a later launch continues from it; s1 = T: the terminal state)
    double u[NU];
    ASLR_UNROLL for (int i = 0; i < NU; ++i) {
      double s = S(oU + i) - S(oKf + i) * alpha;
      ASLR_UNROLL for (int jx = 0; jx < NX; ++jx) s -= S(oK + i * NX + jx) * dx[jx];
      u[i] = s;
    }
    const int m_now = mi;
    const DevModel &dm = D.models[m_now];
    if (m_now != m_loaded) { // wave-uniform: model constants and control limits, re-read only when the model changes
      mr.load(dm);
      m_loaded = m_now;
      lim_has = lim.has[m_now];
      ASLR_UNROLL for (int i = 0; i < NU; ++i) { lim_lb[i] = lim.lb[m_now][i]; lim_ub[i] = lim.ub[m_now][i]; }
    }
    if (box && lim_has) {
      ASLR_UNROLL for (int i = 0; i < NU; ++i) u[i] = fmin(fmax(u[i], lim_lb[i]), lim_ub[i]);
    }
    if (lane_on) {
      ASLR_UNROLL for (int p = 0; p < NU / 2; ++p)
        *reinterpret_cast<double2 *>(a.us_try + cand_piece<NU>(ai, t, b, p, B, T)) = make_double2(u[2 * p], u[2 * p + 1]);
    }
    double xnext[NX], c;
    knot_eval<NJ, DAM, kEvalDyn, CH>(cc, mr, dm, nullptr, x, u, xnext, c, nullptr);
    double mx = 0.0;
    ASLR_UNROLL for (int i = 0; i < NX; ++i) { mx += fabs(xnext[i]); x[i] = xnext[i]; }
    if (inf_norm_bad<NX>(mx, xnext)) fail = true; // NaN / Inf / |xnext|_inf >= 1e30 ("forward_error")
  }
  if (lane_on) {
    TI[(ASLR_TI_TRYFAIL0 + ai) * B + b] = fail ? 1 : 0;
    TF[(ASLR_TF_DVTRY0 + ai) * B + b] = dv;
  }
}
template <int NJ, int DAM, bool PLANAR, bool FDDP>
__global__ void __launch_bounds__(64) rollout_kernel(KArgs a, SolverDev sp, ModelLimits lim) {
  ASLR_STAMP_BEGIN(a, 2);
  rollout_body<NJ, DAM, PLANAR, FDDP>(a, sp, lim, blockIdx.x, a.seg_t0, a.seg_t1);
  ASLR_STAMP_END(2, true);
}

// cost of every stored candidate knot: COST_TRY[a][t][b]
// FAST (planar chains with PlanarChain::reach_ok): closed-form frame-placement residual
template <int NJ, int DAM, bool PLANAR, bool FAST = false>
ASLR_DEV void trial_cost_body(const KArgs &a, const SolverDev &sp, int vbx, int t, int ai) {
  constexpr int NX = 4 * NJ, NU = ModelDims<NJ, DAM>::nu;
  using CH = std::conditional_t<PLANAR, ChainPlanar<NJ>, Chain3D<NJ>>;
  const int B = a.B, T = a.T;
  const int bq = a.b0 + vbx * 64 + threadIdx.x;
  const bool valid = bq < a.b1;
  const int b = valid ? bq : a.b1 - 1;
  const int done = sp.standalone ? 0 : a.traj_i[ASLR_TI_DONE * B + b];
  if (!valid || done) return;
  const size_t TB1 = (size_t)(T + 1) * B, TB = (size_t)T * B, tb = (size_t)t * B + b;
  double x[NX], u[NU], xnext[NX], c = 0.0;
  if (NX % 2 == 0 && NU % 2 == 0) {
    ASLR_UNROLL for (int p = 0; p < NX / 2; ++p) {
      const double2 v = *reinterpret_cast<const double2 *>(a.xs_try + cand_piece<NX>(ai, t, b, p, B, T + 1));
      x[2 * p] = v.x; x[2 * p + 1] = v.y;
    }
    if (t < T) {
      ASLR_UNROLL for (int p = 0; p < NU / 2; ++p) {
        const double2 v = *reinterpret_cast<const double2 *>(a.us_try + cand_piece<NU>(ai, t, b, p, B, T));
        u[2 * p] = v.x; u[2 * p + 1] = v.y;
      }
    }
  } else { // plain slabs (odd widths)
    const double *xs = a.xs_try + ((size_t)ai * TB1 + tb) * NX;
    ASLR_UNROLL for (int i = 0; i < NX; ++i) x[i] = xs[i];
    if (t < T) {
      const double *us = a.us_try + ((size_t)ai * TB + tb) * NU;
      ASLR_UNROLL for (int i = 0; i < NU; ++i) u[i] = us[i];
    }
  }
  const DevDesc &D = *a.desc;
  const DevModel &dm = D.models[node_model_at(a, t)];
  const double *fref = a.frame_ref ? a.frame_ref + 12 * (size_t)b : nullptr;
  const typename CH::Consts cc(D);
  ModelRegs<NJ, NU> mr;
  mr.load(dm);
  knot_eval<NJ, DAM, kEvalCost | (FAST && PLANAR ? kEvalFastReach : 0), CH>(cc, mr, dm, fref, x, t < T ? u : nullptr, xnext, c, nullptr);
  a.cost_try[(size_t)ai * TB1 + tb] = c;
}
// grid (ceil(nb / 64), knots of the segment, step lengths): knots a.seg_t0 + blockIdx.y
template <int NJ, int DAM, bool PLANAR, bool FAST = false>
__global__ void __launch_bounds__(64) trial_cost_kernel(KArgs a, SolverDev sp) {
  ASLR_STAMP_BEGIN(a, 4);
  trial_cost_body<NJ, DAM, PLANAR, FAST>(a, sp, blockIdx.x, a.seg_t0 + blockIdx.y, blockIdx.z);
  ASLR_STAMP_END(4, false);
}
// One launch, two roles: blocks [0, nroll) continue the rollout over the knots [r0, r1] while the others evaluate the trial
// costs of the knots [c0, c0 + cknots) the previous rollout launch has stored (block order = dispatch order: the sweep waves
// take their SIMDs first, the cost waves fill in next to them).  Cost block v - nroll = (x, knot, step length), x fastest.
template <int NJ, int DAM, bool PLANAR, bool FDDP, bool FAST>
__global__ void __launch_bounds__(64) rollout_and_cost_kernel(KArgs a, SolverDev sp, ModelLimits lim, int nroll, int r0, int r1,
                                                              int cgx, int c0, int cknots) {
  ASLR_STAMP_BEGIN(a, 3);
  if ((int)blockIdx.x < nroll) {
    rollout_body<NJ, DAM, PLANAR, FDDP>(a, sp, lim, blockIdx.x, r0, r1);
    ASLR_STAMP_END(3, true);
  } else {
    const int v = blockIdx.x - nroll, vx = v % cgx, rest = v / cgx;
    trial_cost_body<NJ, DAM, PLANAR, FAST>(a, sp, vx, c0 + rest % cknots, rest / cknots);
    ASLR_STAMP_END(3, false);
  }
}

// cost_try of each candidate: its node costs summed in rollout order, one lane per (trajectory, alpha);
// all T+1 loads of a lane are independent, only the adds are ordered
template <int TAG>
__global__ void __launch_bounds__(64) sum_cost_kernel(KArgs a, SolverDev sp) {
  ASLR_STAMP_BEGIN(a, 5);
  const int B = a.B, T = a.T, s = blockIdx.y;
  const int b = a.b0 + blockIdx.x * 64 + threadIdx.x;
  if (b >= a.b1) return;
  if (!sp.standalone && a.traj_i[ASLR_TI_DONE * B + b]) return;
  const double *src = a.cost_try + (size_t)s * (T + 1) * B + b;
  double acc = 0.0;
  int t = 0;
  for (; t + 16 <= T + 1; t += 16) {
    double v[16];
    ASLR_UNROLL for (int i = 0; i < 16; ++i) v[i] = src[(size_t)(t + i) * B];
    ASLR_UNROLL for (int i = 0; i < 16; ++i) acc += v[i];
  }
  for (; t <= T; ++t) acc += src[(size_t)t * B];
  const bool fail = a.traj_i[(ASLR_TI_TRYFAIL0 + s) * B + b] || is_bad(acc);
  a.traj_f[(ASLR_TF_COST_TRY0 + s) * B + b] = fail ? NAN : acc;
  ASLR_STAMP_END(5, true);
}

// line search + solver-state update, one lane per trajectory (model independent)
template <int TAG>
__global__ void __launch_bounds__(64) select_kernel(KArgs a, SolverDev sp) {
  ASLR_STAMP_BEGIN(a, 6);
  const int B = a.B;
  const int b = a.b0 + blockIdx.x * 64 + threadIdx.x;
  if (b >= a.b1) return;
  int32_t *TI = a.traj_i;
  double *TF = a.traj_f;
  if (!sp.standalone && TI[ASLR_TI_DONE * B + b]) return;
  const int feasible = TI[ASLR_TI_FEASIBLE * B + b];
  const bool fddp = sp.solver == ASLR_SOLVER_FDDP;
  double c_try[ASLR_NALPHA];
  int fail_s[ASLR_NALPHA];
  ASLR_UNROLL for (int s = 0; s < ASLR_NALPHA; ++s) {
    c_try[s] = TF[(ASLR_TF_COST_TRY0 + s) * B + b]; // NaN marks a failed trial (sum_cost_kernel)
    fail_s[s] = isnan(c_try[s]);
  }
  if (sp.standalone) return;

  const double cost0 = TF[ASLR_TF_COST * B + b];
  double d1 = TF[ASLR_TF_D1 * B + b], d2 = TF[ASLR_TF_D2 * B + b];
  const double dg = TF[ASLR_TF_DG * B + b], dq = TF[ASLR_TF_DQ * B + b];
  int accepted = -1, status = TI[ASLR_TI_STATUS * B + b];
  double dV = 0.0, dVexp = 0.0, step = 1.0, cost_acc = cost0;
  ASLR_UNROLL for (int s = 0; s < ASLR_NALPHA; ++s) {
    if (accepted < 0) {
      const double as = 1.0 / (double)(1 << s);
      step = as;
      if (fail_s[s]) {
        status |= ASLR_ST_FORWARD_ERR;
      } else {
        dV = cost0 - c_try[s];
        bool acc = false;
        if (fddp) {
          const double dv_s = TF[(ASLR_TF_DVTRY0 + s) * B + b];
          d1 = dg + dv_s;
          d2 = dq - 2.0 * dv_s;
          dVexp = as * (d1 + 0.5 * as * d2);
          if (dVexp >= 0.0) acc = (d1 < sp.th_grad) || (dV > sp.th_acceptstep * dVexp);
          else acc = (!feasible) && (dV > sp.th_acceptnegstep * dVexp);
        } else {
          dVexp = as * (d1 + 0.5 * as * d2);
          if (dVexp >= 0.0) acc = (d1 < sp.th_grad) || (!feasible) || (dV > sp.th_acceptstep * dVexp);
        }
        if (acc) { accepted = s; cost_acc = c_try[s]; }
      }
    }
  }
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
  const int status_cb = status; // what a callback of this iteration reads: Crocoddyl calls them before its convergence test
  if (!fin && !sp.fixed_iterations && was_feasible && stop < sp.th_stop) { status |= ASLR_ST_CONVERGED; fin = 1; }
  if (sp.maxiter_traj > 0 && TI[ASLR_TI_ITER * B + b] + 1 >= sp.maxiter_traj) fin = 1; // (pool solves: maxiter reached)
  const int it = TI[ASLR_TI_ITER * B + b];
  TI[ASLR_TI_ITER * B + b] = it + 1;
  if (a.iter_log && it < a.log_cap) { // what the callbacks of this iteration would read (aslr_set_iteration_log)
    double *lg = a.iter_log + (size_t)it * ASLR_LOG_COUNT * B + b;
    lg[(size_t)ASLR_LOG_COST * B] = cost_acc;
    lg[(size_t)ASLR_LOG_STOP * B] = stop;
    lg[(size_t)ASLR_LOG_XREG * B] = xreg;
    lg[(size_t)ASLR_LOG_STEP * B] = step;
    lg[(size_t)ASLR_LOG_D1 * B] = d1;
    lg[(size_t)ASLR_LOG_D2 * B] = d2;
    lg[(size_t)ASLR_LOG_DV * B] = dV;
    lg[(size_t)ASLR_LOG_DVEXP * B] = dVexp;
    lg[(size_t)ASLR_LOG_ACCEPTED * B] = (double)accepted;
    lg[(size_t)ASLR_LOG_STATUS * B] = (double)status_cb;
    lg[(size_t)ASLR_LOG_FEASIBLE * B] = (double)feas;
  }
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
  ASLR_STAMP_END(6, true);
  ASLR_STAMP_NEXT();
}

} // namespace aslr
