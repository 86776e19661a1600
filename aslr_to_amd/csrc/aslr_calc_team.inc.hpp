// aslr_calc_team.inc.hpp -- rigid-body part of calcDiff for the larger chain (7-DoF SEA), A TEAM OF 8 LANES PER KNOT.
//
// calc_kernel evaluates a knot in one lane; for a 7-joint chain that is 8 Newton-Euler sweeps, a 7 x 7 inverse, one
// more sweep that keeps its intermediates and 14 tangent sweeps (computeRNEADerivatives) back to back, ~250 live
// doubles: 10 KB of scratch per lane.  This kernel does exactly that arithmetic (Chain3D::nle / mass /
// rnea_derivatives and the accelerations of free_fwddyn_asr.py:20-56, same operation order per entry) with the
// sweeps side by side:
//
//   lanes 0..NJ-1 : column c of M (unit-acceleration RNEA), column c of M^-1, entry c of the accelerations,
//                   column c of dtau/dq, then column c of dtau/dv (tangent sweeps with a runtime direction);
//   lane  NJ      : the nonlinear effects.
//
// Intermediates shared by the team (joint rotations, M, M^-1, the kept sweep) live in LDS.  The results --
// [xout (2 nj) | M^-1 | the link rows of da_dx = M^-1 [-dtau/dq - K | K | -dtau/dv]] per knot, region DYN (layout:
// dyn_len_c) -- are consumed by calc_kernel<..., PRE>, which then only does the cost stack and the record assembly and
// reads each da_dx entry where a record line needs it (nothing of the 3 nj^2 block is held in registers).
#pragma once
#include "aslr_forward_team.inc.hpp"

namespace aslr {

// forward-mode tangent of RNEA along q_j (KIND 0) or v_j (KIND 1), Chain3D::rnea_derivatives with a RUNTIME j:
// the joints upstream of j carry exact zeros, the special values at i == j are selected.
// ws (LDS): per joint [v | h | F | Xv | Xa], 6 doubles each (lin, ang).
template <int NJ, int KIND>
ASLR_DEV void rnea_tangent_lds(chain_cp c, const double *Rl, const double *ws, const double *velL, int j,
                                              double *colL) {
  double vv[NJ], col[NJ];
  ASLR_UNROLL for (int i = 0; i < NJ; ++i) vv[i] = velL[i];
  auto sv_at = [&](int i, int which) {
    const double *p = ws + 30 * i + 6 * which;
    return SV{V3{p[0], p[1], p[2]}, V3{p[3], p[4], p[5]}};
  };
  auto sel = [](bool at, SV a, SV b) {
    return SV{V3{at ? a.lin.x : b.lin.x, at ? a.lin.y : b.lin.y, at ? a.lin.z : b.lin.z},
              V3{at ? a.ang.x : b.ang.x, at ? a.ang.y : b.ang.y, at ? a.ang.z : b.ang.z}};
  };
  SV dvp = sv_zero(), dap = sv_zero();
  SV df[NJ];
  ASLR_UNROLL for (int i = 0; i < NJ; ++i) {
    SE3d X;
    X.R = m3(Rl + 9 * i);
    X.p = v3(c->joint_p[i]);
    const V3 ax = v3(c->axis[i]);
    const SV Si = SV{V3{0, 0, 0}, ax};
    const SV vi = sv_at(i, 0), hi = sv_at(i, 1);
    const bool at = i == j;
    SV sv, sa;
    if (KIND == 0) { sv = crm(sv_at(i, 3), Si); sa = crm(sv_at(i, 4), Si); }
    else { sv = Si; sa = crm(vi, Si); }
    const SV dvi = sel(at, sv, motion_actinv(X, dvp));
    SV dai = sel(at, sa, motion_actinv(X, dap));
    const SV vJ = SV{V3{0, 0, 0}, vv[i] * ax};
    dai = dai + crm(dvi, vJ);
    const V3 com = v3(c->com[i]);
    const M3 I = m3(c->inertia[i]);
    df[i] = inertia_mul(c->mass[i], com, I, dai) + crf(dvi, hi) + crf(vi, inertia_mul(c->mass[i], com, I, dvi));
    dvp = dvi;
    dap = dai;
  }
  ASLR_UNROLL for (int i = NJ - 1; i >= 0; --i) {
    const V3 ax = v3(c->axis[i]);
    col[i] = dot(ax, df[i].ang);
    if (i > 0) {
      SE3d X;
      X.R = m3(Rl + 9 * i);
      X.p = v3(c->joint_p[i]);
      df[i - 1] = df[i - 1] + force_act(X, df[i]);
      if (KIND == 0) {
        const SV ex = force_act(X, crf(SV{V3{0, 0, 0}, ax}, sv_at(i, 2)));
        df[i - 1] = df[i - 1] + sel(i == j, ex, sv_zero());
      }
    }
  }
  ASLR_UNROLL for (int i = 0; i < NJ; ++i) colL[i] = col[i];
}

// rnea<NJ, true>() with the rotations in LDS; lane `writer` stores what the tangent sweeps need
template <int NJ>
ASLR_DEV void rnea_keep_lds(chain_cp c, const double *Rl, const double *velL, const double *accL,
                                           double *ws, bool writer) {
  double vv[NJ], aa[NJ];
  ASLR_UNROLL for (int i = 0; i < NJ; ++i) { vv[i] = velL[i]; aa[i] = accL[i]; }
  const V3 grav = v3(c->gravity);
  auto put = [&](int i, int which, SV s) {
    double *p = ws + 30 * i + 6 * which;
    p[0] = s.lin.x; p[1] = s.lin.y; p[2] = s.lin.z; p[3] = s.ang.x; p[4] = s.ang.y; p[5] = s.ang.z;
  };
  SV vp = sv_zero(), ap = SV{neg(grav), V3{0, 0, 0}};
  SV f[NJ];
  ASLR_UNROLL for (int i = 0; i < NJ; ++i) {
    SE3d X;
    X.R = m3(Rl + 9 * i);
    X.p = v3(c->joint_p[i]);
    const V3 ax = v3(c->axis[i]);
    const SV vJ = SV{V3{0, 0, 0}, vv[i] * ax};
    const SV Xv = motion_actinv(X, vp);
    const SV vi = Xv + vJ;
    const SV Xa = motion_actinv(X, ap);
    SV ai = Xa + crm(vi, vJ);
    ai.ang = ai.ang + aa[i] * ax;
    const V3 com = v3(c->com[i]);
    const M3 I = m3(c->inertia[i]);
    const SV h = inertia_mul(c->mass[i], com, I, vi);
    f[i] = inertia_mul(c->mass[i], com, I, ai) + crf(vi, h);
    if (writer) { put(i, 0, vi); put(i, 1, h); put(i, 3, Xv); put(i, 4, Xa); }
    vp = vi;
    ap = ai;
  }
  ASLR_UNROLL for (int i = NJ - 1; i >= 0; --i) {
    if (i > 0) {
      SE3d X;
      X.R = m3(Rl + 9 * i);
      X.p = v3(c->joint_p[i]);
      f[i - 1] = f[i - 1] + force_act(X, f[i]);
    }
    if (writer) put(i, 2, f[i]);
  }
}

template <int NJ>
struct CalcTeam {
  static constexpr int NX = 4 * NJ, NU = NJ;
  static constexpr int tX = 0, tU = tX + NX, tTc = tU + 8, tTm = tTc + 8, tR = tTm + 8, tM = tR + (NJ * 9 + 1) / 2 * 2,
                       tMi = tM + 64, tXo = tMi + 64, tWs = tXo + 16, TEAM_LDS = tWs + (NJ * 30 + 1) / 2 * 2;
};

// mode: calc_kernel's (kModeCommit: read the accepted candidate; kModeSolver: honour RECALC / DONE).
// PHASE 0: M, nonlinear effects, M^-1, accelerations; PHASE 1: the kept sweep, dtau/dq and dtau/dv.  Two launches of
// the same body: each phase gets its own register allocation (262 and 436 VGPRs, no scratch; in one kernel the first
// part would run at the occupancy of the second: measured 706 vs 684 us for the whole calcDiff).
template <int NJ, int PHASE>
__global__ void __launch_bounds__(64) dyn_team_kernel(KArgs a, int mode) {
  using C = CalcTeam<NJ>;
  constexpr int NX = C::NX, NU = C::NU, DL = dyn_len_c(NJ);
  __shared__ double sm[8 * C::TEAM_LDS];
  const int tid = threadIdx.x, team = tid >> 3, c = tid & 7;
  const int B = a.B, T = a.T, t = blockIdx.y;
  const int bq = a.b0 + blockIdx.x * 8 + team;
  const bool valid = bq < a.b1;
  const int b = valid ? bq : a.b1 - 1;
  const int32_t *TI = a.traj_i;
  int acc = -1, recalc = 1, done = 0;
  if (mode & kModeCommit) acc = TI[ASLR_TI_ACCEPTED * B + b];
  if (mode & kModeSolver) { recalc = TI[ASLR_TI_RECALC * B + b]; done = TI[ASLR_TI_DONE * B + b]; }
  const bool compute = valid && recalc && !done && !(mode & kModeNoCompute);
  if (__ballot(compute) == 0ull) return;
  double *tm_ = sm + team * C::TEAM_LDS;
  double *xT = tm_ + C::tX, *uT = tm_ + C::tU, *tcL = tm_ + C::tTc, *tmL = tm_ + C::tTm, *RL = tm_ + C::tR,
         *ML = tm_ + C::tM, *MiL = tm_ + C::tMi, *xoL = tm_ + C::tXo, *WS = tm_ + C::tWs;
  const size_t tb = (size_t)t * B + b, TB1 = (size_t)(T + 1) * B, TB = (size_t)T * B;
  const bool jl = c < NJ;
  const int cj = jl ? c : NJ - 1;
  const DevDesc &D = *a.desc;
  const aslr_chain_t &ch = D.chain;
  const chain_cp chc = chain_const(&D.chain);
  const DevModel &dm = D.models[node_model_at(a, t)];

  // ---- x, u of this knot (from the accepted candidate when there is one; the terminal node takes u = 0) ----
  {
    const double *src = acc >= 0 ? a.xs_try + ((size_t)acc * TB1 + tb) * NX : a.xs + tb * NX;
    ASLR_UNROLL for (int k = 0; k < 4; ++k) {
      const int e = c + 8 * k;
      if (e < NX) xT[e] = src[e];
    }
    if (jl) {
      double uv = 0.0;
      if (t < T) uv = (acc >= 0 ? a.us_try + ((size_t)acc * TB + tb) * NU : a.us + tb * NU)[c];
      uT[c] = uv;
    }
  }
  wave_sync();
  double Krow[NJ], Srow[NJ], Brow[NJ];
  ASLR_UNROLL for (int j = 0; j < NJ; ++j) {
    Krow[j] = dm.m.K[cj * NJ + j]; Srow[j] = dm.m.S[cj * NU + j]; Brow[j] = dm.Binv[cj * NJ + j];
  }
  if (jl) {
    const M3 R = mul(m3(ch.joint_R[cj]), axis_angle(v3(ch.axis[cj]), xT[cj]));
    ASLR_UNROLL for (int k = 0; k < 9; ++k) RL[9 * c + k] = R.a[k];
    double s = 0.0, s2 = 0.0;
    ASLR_UNROLL for (int j = 0; j < NJ; ++j) { s += Krow[j] * (xT[j] - xT[NJ + j]); s2 += Srow[j] * uT[j]; }
    tcL[c] = s;
    tmL[c] = s2;
  }
  wave_sync();
  double *out = a.dyn + tb * DL;
  if constexpr (PHASE == 0) {
    const V3 g = v3(chc->gravity);
    // ---- M columns and nonlinear effects: one RNEA per lane ----
    {
      double vv[NJ], aa[NJ], tau[NJ];
      const bool nl = c == NJ;
      ASLR_UNROLL for (int i = 0; i < NJ; ++i) { vv[i] = nl ? xT[2 * NJ + i] : 0.0; aa[i] = (!nl && i == c) ? 1.0 : 0.0; }
      rnea_lds<NJ>(chc, RL, vv, aa, nl ? g : V3{0.0, 0.0, 0.0}, tau);
      if (c <= NJ) { ASLR_UNROLL for (int i = 0; i < NJ; ++i) ML[8 * c + i] = tau[i]; }
    }
    wave_sync();
    // ---- column cj of M^-1 ----
    {
      double Ms[NJ][NJ], e[NJ];
      ASLR_UNROLL for (int i = 0; i < NJ; ++i)
        ASLR_UNROLL for (int j = 0; j <= i; ++j) {
          Ms[i][j] = (i == j) ? ML[8 * j + i] : 0.5 * (ML[8 * j + i] + ML[8 * i + j]);
          Ms[j][i] = Ms[i][j];
        }
      spd_inverse_col<NJ>(Ms, cj, e);
      if (jl) {
        ASLR_UNROLL for (int i = 0; i < NJ; ++i) MiL[8 * c + i] = e[i];
        if (compute) { ASLR_UNROLL for (int i = 0; i < NJ; ++i) out[2 * NJ + i * NJ + c] = e[i]; } // Minv[i][c]
      }
    }
    wave_sync();
    // ---- accelerations, entry cj ----
    {
      double al = 0.0, am = 0.0;
      ASLR_UNROLL for (int j = 0; j < NJ; ++j) {
        al += MiL[8 * j + cj] * (-ML[8 * NJ + j] - tcL[j]);
        am += Brow[j] * (tmL[j] + tcL[j]);
      }
      if (jl && compute) { out[c] = al; out[NJ + c] = am; }
    }
  } else {
    // ---- RNEA(q, v, a_link) keeping its intermediates, then the two tangent sweeps of this lane ----
    if (jl) {
      xoL[c] = out[c]; // link accelerations and column c of M^-1 from phase 0
      ASLR_UNROLL for (int i = 0; i < NJ; ++i) MiL[8 * c + i] = out[2 * NJ + i * NJ + c];
    }
    wave_sync();
    rnea_keep_lds<NJ>(chc, RL, xT + 2 * NJ, xoL, WS, c == 0);
    wave_sync();
    // ---- column cj of dtau/dq, of dtau/dv, and with each the same column of the link rows of da_dx =
    //      M^-1 [-dtau/dq - K | K | -dtau/dv] (free_fwddyn_asr.py:76-81): the sums of the per-lane evaluation (knot_eval)
    //      entry by entry, l ascending; operands from LDS, results into the row layout of DYN ----
    double *colL = ML + 8 * c;
    double *arow = out + dyn_oa_c(NJ) + c;
    rnea_tangent_lds<NJ, 0>(chc, RL, WS, xT + 2 * NJ, cj, colL);
    {
      double Kcol[NJ];
      ASLR_UNROLL for (int l = 0; l < NJ; ++l) Kcol[l] = dm.m.K[l * NJ + cj];
      ASLR_UNROLL for (int i = 0; i < NJ; ++i) {
        double sq = 0.0, sk = 0.0;
        ASLR_UNROLL for (int l = 0; l < NJ; ++l) {
          const double mi = MiL[8 * l + i];
          sq += mi * (-colL[l] - Kcol[l]);
          sk += mi * Kcol[l];
        }
        if (jl && compute) { arow[i * dyn_row_c(NJ)] = sq; arow[i * dyn_row_c(NJ) + NJ] = sk; }
      }
    }
    rnea_tangent_lds<NJ, 1>(chc, RL, WS, xT + 2 * NJ, cj, colL);
    ASLR_UNROLL for (int i = 0; i < NJ; ++i) {
      double sv = 0.0;
      ASLR_UNROLL for (int l = 0; l < NJ; ++l) sv += MiL[8 * l + i] * (-colL[l]);
      if (jl && compute) arow[i * dyn_row_c(NJ) + 2 * NJ] = sv;
    }
  }
}

} // namespace aslr
