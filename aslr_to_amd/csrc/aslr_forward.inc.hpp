// aslr_forward.inc.hpp -- forward rollout + line search + solver-state update kernel
#pragma once
#include "aslr_common.hpp"

namespace aslr {
// =================================================================================================
// forward pass + line search + solver-state update
// =================================================================================================
template <int NJ, int DAM, bool PLANAR>
__global__ void __launch_bounds__(64) forward_kernel(KArgs a, SolverDev sp) {
  constexpr int NX = 4 * NJ, NU = ModelDims<NJ, DAM>::nu;
  constexpr int TEAM = 16, TPW = 4;
  using CH = std::conditional_t<PLANAR, ChainPlanar<NJ>, Chain3D<NJ>>;
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
      knot_eval<NJ, DAM, kEvalDyn | kEvalCost, CH>(D, dm, fref, x, u, xnext, c, nullptr);
      cost_try += c;
      double mx = 0.0;
      bool nan = false;
      ASLR_UNROLL for (int i = 0; i < NX; ++i) { nan = nan || isnan(xnext[i]); mx = fmax(mx, fabs(xnext[i])); }
      if (is_bad(cost_try) || nan || is_bad(mx)) fail = true;
      ASLR_UNROLL for (int i = 0; i < NX; ++i) x[i] = xnext[i];
    } else {
      knot_eval<NJ, DAM, kEvalCost, CH>(D, dm, fref, x, nullptr, xnext, c, nullptr);
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


} // namespace aslr
