// aslr_forward_team.inc.hpp -- rollout of the larger chain (7-DoF SEA): A TEAM OF 8 LANES PER (trajectory, alpha).
//
// Same arithmetic as rollout_kernel (aslr_forward.inc.hpp) -- SolverDDP / FDDP / BoxDDP forwardPass, SURVEY.md
// B.2, B.4, B.5 -- but the knot evaluation, which for a 7-joint chain is 8 recursive Newton-Euler sweeps (one
// per column of the joint-space inertia plus the nonlinear effects) followed by a 7 x 7 inverse, is spread over
// the lanes of the team instead of running back to back in one lane (where its ~170 live doubles spill):
//
//   lane c < NJ : row c of the control law, joint c's rotation, column c of M (RNEA with unit acceleration
//                 e_c, no velocity, no gravity: adding the exact zeros of the general recursion changes no
//                 bit), column c of M^-1, entry c of the accelerations and of the Euler step;
//   lane NJ     : the nonlinear effects (RNEA with the velocity and gravity, zero acceleration).
//
// All 8 lanes therefore run ONE instruction stream (the general RNEA) on different inputs.  Team-shared data
// (state, control, joint rotations, M, M^-1) lives in LDS; the teams of a wave never interact, so a
// wave-level LDS fence is the only synchronisation.  One block = 2 waves = the 10 step lengths of one
// trajectory (teams 10..15 idle): the C5 shard of 512 trajectories is 1024 waves, one per SIMD.
// Per-knot inputs shared by the step lengths (K, k, us, xs, gaps, Vxx f) are staged per wave in LDS and
// prefetched one knot ahead.
#pragma once
#include "aslr_common.hpp"

namespace aslr {

// RNEA(q, v, a) of rnea<NJ, false>() with the joint rotations read from LDS (Rl[NJ][9], row-major)
template <int NJ>
ASLR_DEV void rnea_lds(chain_cp cp, const double *Rl, const double (&vv)[NJ], const double (&aa)[NJ], V3 grav,
                       double (&tau)[NJ]) {
  SV vp = sv_zero(), ap = SV{neg(grav), V3{0, 0, 0}};
  SV f[NJ];
  ASLR_UNROLL for (int i = 0; i < NJ; ++i) {
    SE3d X;
    X.R = m3(Rl + 9 * i);
    X.p = v3(cp->joint_p[i]);
    const V3 ax = v3(cp->axis[i]);
    const SV vJ = SV{V3{0, 0, 0}, vv[i] * ax};
    const SV Xv = motion_actinv(X, vp);
    const SV vi = Xv + vJ;
    const SV Xa = motion_actinv(X, ap);
    SV ai = Xa + crm(vi, vJ);
    ai.ang = ai.ang + aa[i] * ax;
    const V3 com = v3(cp->com[i]);
    const M3 I = m3(cp->inertia[i]);
    const SV h = inertia_mul(cp->mass[i], com, I, vi);
    f[i] = inertia_mul(cp->mass[i], com, I, ai) + crf(vi, h);
    vp = vi;
    ap = ai;
  }
  ASLR_UNROLL for (int i = NJ - 1; i >= 0; --i) {
    tau[i] = dot(v3(cp->axis[i]), f[i].ang);
    if (i > 0) {
      SE3d X;
      X.R = m3(Rl + 9 * i);
      X.p = v3(cp->joint_p[i]);
      f[i - 1] = f[i - 1] + force_act(X, f[i]);
    }
  }
}

// column jc (runtime) of A^-1 exactly as spd_inverse_fast<N>() computes it (terms it skips are exact zeros here)
template <int N>
ASLR_DEV void spd_inverse_col(const double (&A)[N][N], int jc, double (&e)[N]) {
  double L[N][N], rinv[N];
  ASLR_UNROLL for (int j = 0; j < N; ++j) {
    double d = A[j][j];
    ASLR_UNROLL for (int k = 0; k < j; ++k) d -= L[j][k] * L[j][k];
    d = sqrt(d);
    L[j][j] = d;
    rinv[j] = 1.0 / d;
    ASLR_UNROLL for (int i = j + 1; i < N; ++i) {
      double s = A[i][j];
      ASLR_UNROLL for (int k = 0; k < j; ++k) s -= L[i][k] * L[j][k];
      L[i][j] = s * rinv[j];
    }
  }
  ASLR_UNROLL for (int i = 0; i < N; ++i) {
    double s = (i == jc) ? 1.0 : 0.0;
    ASLR_UNROLL for (int k = 0; k < i; ++k) s -= L[i][k] * e[k];
    e[i] = (i < jc) ? 0.0 : s * rinv[i];
  }
  ASLR_UNROLL for (int i = N - 1; i >= 0; --i) {
    double s = e[i];
    ASLR_UNROLL for (int k = i + 1; k < N; ++k) s -= L[k][i] * e[k];
    e[i] = s * rinv[i];
  }
}

template <int NJ>
struct FwdTeam {
  static_assert(NJ >= 2 && NJ <= 7, "team of 8 lanes: NJ inertia columns + the nonlinear effects");
  static constexpr int NX = 4 * NJ, NU = NJ; // SEA
  // per-wave stage (doubles): K, us, k, xs, gaps, Vxx f of the current knot
  static constexpr int oK = 0, oU = oK + NU * NX, oKf = oU + NU, oXr = oKf + NU, oFg = oXr + NX, oVf = oFg + NX,
                       STG = (oVf + NX + 1) / 2 * 2;
  static constexpr int NSLOT = (STG + 63) / 64;
  // per-team arrays (doubles)
  static constexpr int tX = 0, tU = tX + NX, tTc = tU + 8, tTm = tTc + 8, tR = tTm + 8, tM = tR + (NJ * 9 + 1) / 2 * 2,
                       tMi = tM + 8 * 8, TEAM_LDS = tMi + 8 * 8;
};

template <int NJ, bool FDDP>
__global__ void __launch_bounds__(128) rollout_team_kernel(KArgs a, SolverDev sp, ModelLimits lim) {
  using C = FwdTeam<NJ>;
  constexpr int NX = C::NX, NU = C::NU, NSLOT = C::NSLOT;
  __shared__ double sm[2 * C::STG + 16 * C::TEAM_LDS];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, team = tid >> 3, c = tid & 7;
  const int B = a.B, T = a.T, b = a.b0 + blockIdx.x;
  int32_t *TI = a.traj_i;
  double *TF = a.traj_f;
  if (!sp.standalone && TI[ASLR_TI_DONE * B + b]) return;
  double *stg = sm + wv * C::STG, *tm_ = sm + 2 * C::STG + team * C::TEAM_LDS;
  double *xT = tm_ + C::tX, *uT = tm_ + C::tU, *tcL = tm_ + C::tTc, *tmL = tm_ + C::tTm, *RL = tm_ + C::tR,
         *ML = tm_ + C::tM, *MiL = tm_ + C::tMi;
  const bool team_on = team < ASLR_NALPHA;
  const int ai = team_on ? team : ASLR_NALPHA - 1;
  const double alpha = 1.0 / (double)(1 << ai);
  const int feasible = TI[ASLR_TI_FEASIBLE * B + b];
  const bool fddp = FDDP && sp.solver == ASLR_SOLVER_FDDP, box = sp.solver == ASLR_SOLVER_BOXDDP;
  const bool use_gaps = fddp && !(feasible || alpha == 1.0);
  const bool need_dv = fddp && !feasible;
  const bool jl = c < NJ;            // this lane owns joint / row cj
  const int cj = jl ? c : NJ - 1;
  const DevDesc &D = *a.desc;
  const aslr_chain_t &ch = D.chain;
  const chain_cp chc = chain_const(&D.chain);
  const size_t TB1 = (size_t)(T + 1) * B, TB = (size_t)T * B;
  const unsigned long long team_bits = 0xffull << ((lane >> 3) * 8);

  // ---- stage slots: element idx = lane + 64 q of [K | us | k | xs | gaps | Vxx f] at knot t ----
  const double *sp0[NSLOT];
  size_t sstr[NSLOT];
  bool son[NSLOT], sctl[NSLOT];
  ASLR_UNROLL for (int q = 0; q < NSLOT; ++q) {
    const int idx = lane + 64 * q;
    sp0[q] = a.xs; sstr[q] = 0; son[q] = false; sctl[q] = false;
    if (idx < C::oU) { sp0[q] = a.kgain + (size_t)b * NU * NX + idx; sstr[q] = (size_t)B * NU * NX; son[q] = true; sctl[q] = true; }
    else if (idx < C::oKf) { sp0[q] = a.us + (size_t)b * NU + (idx - C::oU); sstr[q] = (size_t)B * NU; son[q] = true; sctl[q] = true; }
    else if (idx < C::oXr) { sp0[q] = a.kff + (size_t)b * NU + (idx - C::oKf); sstr[q] = (size_t)B * NU; son[q] = true; sctl[q] = true; }
    else if (idx < C::oFg) { sp0[q] = a.xs + (size_t)b * NX + (idx - C::oXr); sstr[q] = (size_t)B * NX; son[q] = true; }
    else if (idx < C::oVf) { sp0[q] = a.gaps + (size_t)b * NX + (idx - C::oFg); sstr[q] = (size_t)B * NX; son[q] = need_dv; }
    else if (idx < C::oVf + NX) { sp0[q] = a.vxxf + (size_t)b * NX + (idx - C::oVf); sstr[q] = (size_t)B * NX; son[q] = need_dv; }
  }
  double pf[NSLOT];
  auto prefetch = [&](int t) {
    ASLR_UNROLL for (int q = 0; q < NSLOT; ++q) {
      pf[q] = 0.0;
      if (son[q] && (!sctl[q] || t < T)) pf[q] = sp0[q][(size_t)t * sstr[q]];
    }
  };

  // model rows of this lane (reloaded when the node's model changes; wave-uniform)
  double Krow[NJ], Srow[NJ], Brow[NJ], dt = 0.0, lb_c = 0.0, ub_c = 0.0;
  bool has_lim = false;
  int m_loaded = -1;
  // joint placements and axes: constants of the chain, staged once in LDS (as global loads inside the knot loop they
  // cost a vmcnt(0) per knot, which also waits for the prefetch issued just before; in registers they spill)
  __shared__ double jtab[8][12];
  if (threadIdx.x < NJ) {
    ASLR_UNROLL for (int k = 0; k < 9; ++k) jtab[threadIdx.x][k] = ch.joint_R[threadIdx.x][k];
    ASLR_UNROLL for (int k = 0; k < 3; ++k) jtab[threadIdx.x][9 + k] = ch.axis[threadIdx.x][k];
  }
  __syncthreads();

  // x0 into the team's state
  ASLR_UNROLL for (int k = 0; k < 4; ++k) {
    const int e = c + 8 * k;
    if (e < NX) xT[e] = a.x0[(size_t)b * NX + e];
  }
  double dv = 0.0;
  bool fail = false;
  ASLR_PROF_DECL;
  prefetch(0);
  for (int t = 0; t <= T; ++t) {
    const size_t tb = (size_t)t * B + b;
    ASLR_PROF(7);
    ASLR_PROF_COUNT(15);
    wave_sync(); // the previous knot's readers of the stage are done
    ASLR_UNROLL for (int q = 0; q < NSLOT; ++q) { if (lane + 64 * q < C::STG) stg[lane + 64 * q] = pf[q]; }
    const int mi = node_model_at(a, t);
    wave_sync();
    if (t < T) prefetch(t + 1);
    if (use_gaps) {
      ASLR_UNROLL for (int k = 0; k < 4; ++k) {
        const int e = c + 8 * k;
        if (e < NX) xT[e] = xT[e] + stg[C::oFg + e] * (alpha - 1.0);
      }
    }
    if (fddp) wave_sync();
    if (team_on && jl) { // candidate state, 4 contiguous entries per lane
      double *o = a.xs_try + ((size_t)ai * TB1 + tb) * NX + 4 * c;
      ASLR_UNROLL for (int k = 0; k < 4; ++k) o[k] = xT[4 * c + k];
    }
    if (need_dv && c == 0) { // dv -= fs . Vxx (xs - xs_try), summed in state order
      double s = 0.0;
      for (int i = 0; i < NX; ++i) s += stg[C::oVf + i] * (stg[C::oXr + i] - xT[i]);
      dv -= s;
    }
    if (t == T) break;
    const DevModel &dm = D.models[mi];
    if (mi != m_loaded) {
      dt = dm.m.dt;
      ASLR_UNROLL for (int j = 0; j < NJ; ++j) {
        Krow[j] = dm.m.K[cj * NJ + j]; Srow[j] = dm.m.S[cj * NU + j]; Brow[j] = dm.Binv[cj * NJ + j];
      }
      has_lim = lim.has[mi] != 0;
      lb_c = lim.lb[mi][cj];
      ub_c = lim.ub[mi][cj];
      // the values are consumed HERE, so the wait for these loads sits inside this rarely-taken branch and not at
      // the join, where it would drain the prefetch of every knot
      ASLR_UNROLL for (int j = 0; j < NJ; ++j) asm volatile("" : "+v"(Krow[j]), "+v"(Srow[j]), "+v"(Brow[j]));
      asm volatile("" : "+v"(dt), "+v"(lb_c), "+v"(ub_c));
      m_loaded = mi;
    }
    ASLR_PROF(0);
    // ---- control law, row cj: u = us - alpha k - K (x - xs), box clamp ----
    {
      double s = stg[C::oU + cj] - stg[C::oKf + cj] * alpha;
      ASLR_UNROLL for (int jx = 0; jx < NX; ++jx) s -= stg[C::oK + cj * NX + jx] * (xT[jx] - stg[C::oXr + jx]);
      if (box && has_lim) s = fmin(fmax(s, lb_c), ub_c);
      if (jl) uT[c] = s;
      if (team_on && jl) a.us_try[((size_t)ai * TB + tb) * NU + c] = s;
    }
    ASLR_PROF(1);
    // ---- rotation of joint cj ----
    if (jl) {
      const M3 R = mul(m3(jtab[cj]), axis_angle(v3(jtab[cj] + 9), xT[cj]));
      ASLR_UNROLL for (int k = 0; k < 9; ++k) RL[9 * c + k] = R.a[k];
    }
    wave_sync();
    ASLR_PROF(2);
    // ---- coupling and motor torques, entry cj ----
    if (jl) {
      double s = 0.0, s2 = 0.0;
      ASLR_UNROLL for (int j = 0; j < NJ; ++j) { s += Krow[j] * (xT[j] - xT[NJ + j]); s2 += Srow[j] * uT[j]; }
      tcL[c] = s;
      tmL[c] = s2;
    }
    ASLR_PROF(3);
    // ---- one RNEA per lane: column c of M (lanes c < NJ), nonlinear effects (lane NJ) ----
    if (c <= NJ) {
      double vv[NJ], aa[NJ], tau[NJ];
      const bool nl = c == NJ;
      ASLR_UNROLL for (int i = 0; i < NJ; ++i) { vv[i] = nl ? xT[2 * NJ + i] : 0.0; aa[i] = (!nl && i == c) ? 1.0 : 0.0; }
      const V3 g = v3(chc->gravity);
      const V3 grav = nl ? g : V3{0.0, 0.0, 0.0};
      rnea_lds<NJ>(chc, RL, vv, aa, grav, tau);
      ASLR_UNROLL for (int i = 0; i < NJ; ++i) ML[8 * c + i] = tau[i];
    }
    wave_sync();
    ASLR_PROF(4);
    // ---- column cj of M^-1 (M symmetrised like Chain3D::mass) ----
    {
      double Ms[NJ][NJ], e[NJ];
      ASLR_UNROLL for (int i = 0; i < NJ; ++i)
        ASLR_UNROLL for (int j = 0; j <= i; ++j) {
          Ms[i][j] = (i == j) ? ML[8 * j + i] : 0.5 * (ML[8 * j + i] + ML[8 * i + j]);
          Ms[j][i] = Ms[i][j];
        }
      spd_inverse_col<NJ>(Ms, cj, e);
      if (jl) { ASLR_UNROLL for (int i = 0; i < NJ; ++i) MiL[8 * c + i] = e[i]; }
    }
    wave_sync();
    ASLR_PROF(5);
    // ---- accelerations and semi-implicit Euler, entries cj of q_l, q_m, v_l, v_m ----
    {
      double al = 0.0, am = 0.0;
      ASLR_UNROLL for (int j = 0; j < NJ; ++j) {
        al += MiL[8 * j + cj] * (-ML[8 * NJ + j] - tcL[j]);
        am += Brow[j] * (tmL[j] + tcL[j]);
      }
      const double ql = xT[cj], qm = xT[NJ + cj], vl = xT[2 * NJ + cj], vm = xT[3 * NJ + cj];
      const double nql = ql + (vl * dt + al * dt * dt), nvl = vl + al * dt;
      const double nqm = qm + (vm * dt + am * dt * dt), nvm = vm + am * dt;
      const double nxt[4] = {nql, nvl, nqm, nvm}; // this lane's entries of xnext: |xnext|_inf test, entry by entry
      const bool bad = jl && inf_norm_bad<4>(fabs(nql) + fabs(nvl) + fabs(nqm) + fabs(nvm), nxt);
      if (__ballot(bad) & team_bits) fail = true; // NaN / Inf / >= 1e30 in the state ("forward_error")
      wave_sync(); // every lane of the team has read the old state
      if (jl) { xT[c] = nql; xT[NJ + c] = nqm; xT[2 * NJ + c] = nvl; xT[3 * NJ + c] = nvm; }
    }
  }
  ASLR_PROF(6);
  ASLR_PROF_FLUSH;
  if (team_on && c == 0) {
    TI[(ASLR_TI_TRYFAIL0 + ai) * B + b] = fail ? 1 : 0;
    TF[(ASLR_TF_DVTRY0 + ai) * B + b] = dv;
  }
}

} // namespace aslr
