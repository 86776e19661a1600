// aslr_backward.inc.hpp -- Riccati backward pass kernel (SolverDDP.backwardPass + computeGains,
// SolverBoxDDP.computeGains + BoxQP)
//
// One TEAM of lanes per trajectory; lane (h, j) owns rows [h*RPL, (h+1)*RPL) of column j of every
// nx-column matrix.  Vxx / Vx of the next knot live in registers (each lane holds the whole column j
// of the symmetric Vxx, which is also its row j); products stream one operand from LDS as a
// broadcast row and keep the other in registers.  The per-knot DERIV record, the box-QP inputs and
// the gap vector of knot t-1 arrive while knot t computes: at nx = 8 with two lane sets per column straight
// into LDS (global_load_lds, issued once this knot's record is consumed), otherwise through prefetch registers.
//
// The kernel is bound by the instruction stream of ONE wave per SIMD (B = 4096 gives 1024 waves of
// 4 teams), so the code avoids selects / clamps / divisions in the loop: rows divide evenly for
// nx = 8, pivots use rsqrt, the NaN test is one sum per lane.
#pragma once
#include <utility>
#include "aslr_common.hpp"
#include "aslr_team_ops.hpp"

namespace aslr {

// Cholesky with reciprocal pivots from rsqrt: L (lower, in place), rinv[i] = 1 / L[i][i].
// Returns true on a non-positive (or NaN) pivot, like Eigen::LLT info() != Success.
template <int N>
ASLR_DEV bool chol_rs(double (&A)[N][N], double (&rinv)[N]) {
  bool bad = false;
  ASLR_UNROLL for (int j = 0; j < N; ++j) {
    double d = A[j][j];
    ASLR_UNROLL for (int k = 0; k < j; ++k) d -= A[j][k] * A[j][k];
    if (!(d > 0.0)) bad = true;
    const double ri = rsqrt(d);
    rinv[j] = ri;
    A[j][j] = d * ri;
    ASLR_UNROLL for (int i = j + 1; i < N; ++i) {
      double s = A[i][j];
      ASLR_UNROLL for (int k = 0; k < j; ++k) s -= A[i][k] * A[j][k];
      A[i][j] = s * ri;
    }
  }
  return bad;
}

// BoxQP on register arrays, evaluated redundantly by every lane of a team.
//
// Branch-free formulation: the free / clamped split is carried by multiplicative masks mk[i] in {1, 0}
// (free / clamped).  The masked matrix  mk_i mk_j H_ij + (1 - mk_i) delta_ij  has the Cholesky factor
// of Hff in its free block and identity rows elsewhere; multiplying by exact 0 / 1 and adding exact
// zeros changes no bits, so every quantity equals what the index-set formulation computes.  All teams of
// a wave step together (wave-uniform loop control by ballot); a team that has converged keeps its
// state through selects and its factor is simply rebuilt from its (final) mask.
// A team whose line search rejects all step lengths stops at once (its state can no longer change).
// In: x = the warm start already clamped to the box, g = q + H x there (the caller has both from its replay of
// the first active-set test).  The objective value travels with x: the accepted trial's f is the next
// iteration's f(x), so H x is formed once per call.
// On exit: x, the clamped flags of the final active set, and kcol <- Quu_inv kcol where Quu_inv is
// Hff^-1 on the free block and zero elsewhere (Crocoddyl forms Hff^-1 explicitly and multiplies; solving
// with the same factor differs by rounding only).
template <int NU>
ASLR_DEV bool boxqp(const double (&H)[NU][NU], const double (&q)[NU], const double (&lb)[NU],
                    const double (&ub)[NU], double (&x)[NU], double (&g)[NU], bool (&cm)[NU], double (&kcol)[NU],
                    int boxqp_maxiter, double th_acceptstep, double th_grad, double reg
#ifdef ASLR_BWD_PROFILE
                    , long long (&prof_acc)[20], long long &prof_last
#endif
                    ) {
  bool bad = false, finished = false;
  double mk[NU], mkL[NU], L[NU][NU], rinv[NU];
  ASLR_UNROLL for (int i = 0; i < NU; ++i) {
    mk[i] = 1.0; mkL[i] = -1.0; rinv[i] = 1.0;
    ASLR_UNROLL for (int j = 0; j < NU; ++j) L[i][j] = 0.0;
  }
  auto factor = [&]() {
    ASLR_UNROLL for (int i = 0; i < NU; ++i) {
      mkL[i] = mk[i];
      ASLR_UNROLL for (int j = 0; j < NU; ++j)
        L[i][j] = (mk[i] * mk[j]) * H[i][j] + (i == j ? (mk[i] * reg + (1.0 - mk[i])) : 0.0);
    }
    return chol_rs<NU>(L, rinv);
  };
  // f(x) = 1/2 x^T H x + q^T x
  double fold = 0.0;
  ASLR_UNROLL for (int i = 0; i < NU; ++i) {
    double s = 0.0;
    ASLR_UNROLL for (int j = 0; j < NU; ++j) s += H[i][j] * x[j];
    fold += 0.5 * x[i] * s + q[i] * x[i];
  }
  ASLR_PROF(6);
  for (int it = 0; it < boxqp_maxiter; ++it) {
    ASLR_PROF_COUNT(12);
    double gnorm = 0.0, nfree = 0.0;
    ASLR_UNROLL for (int j = 0; j < NU; ++j) {
      // (plain & / |: four compares and three mask operations, no short-circuit branches)
      const bool at_lb = (x[j] == lb[j]) & (g[j] > 0.0), at_ub = (x[j] == ub[j]) & (g[j] < 0.0);
      const double mj = (at_lb | at_ub) ? 0.0 : 1.0;
      gnorm = fmax(gnorm, mj * fabs(g[j]));
      nfree += mj;
      mk[j] = finished ? mk[j] : mj;
    }
    const bool fin_now = finished || (gnorm <= th_grad) || (nfree == 0.0);
    if (__ballot(!fin_now) == 0ull) { finished = true; break; }
    // Newton step on the free subspace (teams already finished just rebuild their own factor)
    const bool cbad = factor();
    if (cbad && !fin_now) bad = true;
    const bool stepping = !fin_now && !cbad;
    double rhs[NU], dx[NU];
    ASLR_UNROLL for (int i = 0; i < NU; ++i) {
      double s = -q[i];
      ASLR_UNROLL for (int j = 0; j < NU; ++j) s -= H[i][j] * ((1.0 - mk[j]) * x[j]);
      rhs[i] = mk[i] * s;
    }
    chol_solve_r<NU>(L, rinv, rhs);
    ASLR_UNROLL for (int i = 0; i < NU; ++i) dx[i] = mk[i] * (rhs[i] - x[i]);
    ASLR_PROF(7);
    double alpha = 1.0, fnext = fold;
    bool found = !stepping;
    for (int al = 0; al < ASLR_NALPHA; ++al, alpha *= 0.5) {
      double xn[NU], fnew = 0.0, gd = 0.0;
      ASLR_UNROLL for (int i = 0; i < NU; ++i) xn[i] = fmax(fmin(x[i] + alpha * dx[i], ub[i]), lb[i]);
      ASLR_UNROLL for (int i = 0; i < NU; ++i) {
        double s = 0.0;
        ASLR_UNROLL for (int j = 0; j < NU; ++j) s += H[i][j] * xn[j];
        fnew += 0.5 * xn[i] * s + q[i] * xn[i];
        gd += g[i] * (x[i] - xn[i]);
      }
      const bool take = !found && (fold - fnew > th_acceptstep * gd);
      ASLR_UNROLL for (int i = 0; i < NU; ++i) x[i] = take ? xn[i] : x[i];
      fnext = take ? fnew : fnext;
      found = found || take;
      if (__ballot(!found) == 0ull) break;
    }
    fold = fnext;
    // No step length accepted: x is unchanged, so every remaining iteration would recompute the same
    // gradient, active set and rejected steps and return this x.  Stop here with that result.
    finished = fin_now || cbad || !found;
    // gradient at the new point, accumulated in the oracle's order (q first)
    ASLR_UNROLL for (int i = 0; i < NU; ++i) {
      double s = q[i];
      ASLR_UNROLL for (int j = 0; j < NU; ++j) s += H[i][j] * x[j];
      g[i] = s;
    }
    ASLR_PROF(8);
  }
  ASLR_PROF(7);
  // factor of the final free block: the one at hand unless the active set changed in the last step
  bool stale = false;
  ASLR_UNROLL for (int i = 0; i < NU; ++i) stale = stale || (mkL[i] != mk[i]);
  if (__ballot(stale) != 0ull) {
    double nfree = 0.0;
    ASLR_UNROLL for (int i = 0; i < NU; ++i) nfree += mk[i];
    if (factor() && nfree > 0.0) bad = true;
  }
  ASLR_UNROLL for (int i = 0; i < NU; ++i) { cm[i] = (mk[i] == 0.0); kcol[i] *= mk[i]; }
  chol_solve_r<NU>(L, rinv, kcol);
  ASLR_UNROLL for (int i = 0; i < NU; ++i) kcol[i] *= mk[i];
  ASLR_PROF(9);
  return bad;
}

// SolverDDP / SolverBoxDDP computeGains of one knot on register arrays, every lane of the wave running the same problem
// or its team's (the per-lane form the nu != 4 kernels use; aslr_team_gains.hpp is the lane-distributed form for nu = 4).
// In: Quu (regularised), qu, the lane's column of Qux in Kc; for a boxed node lb / ub (bounds minus u) and the stored k
// as warm start k0.  Out: kv = k, Kc = the lane's column of K, qu with clamped entries zeroed.  Returns "backward_error".
// Box nodes replay BoxQP's first iteration (clamped warm start, gradient, active set); with nothing clamped there the
// two cheap outcomes -- (a) |g|_inf <= th_grad: x0 itself, (b) the Newton point strictly inside the box -- need only the
// plain gains; everything else runs boxqp<NU>.
template <int NU>
ASLR_DEV bool lane_gains(const double (&Quu)[NU][NU], double (&qu)[NU], double (&Kc)[NU], double (&kv)[NU], bool boxed,
                         const double (&lb)[NU], const double (&ub)[NU], const double (&k0)[NU], const SolverDev &sp
#ifdef ASLR_BWD_PROFILE
                         , long long (&prof_acc)[20], long long &prof_last
#endif
                         ) {
  bool failed = false;
  double x0[NU], g0[NU], Qux[NU];
  bool any_clamped = false;
  double gnorm0 = 0.0;
  ASLR_UNROLL for (int c = 0; c < NU; ++c) { Qux[c] = Kc[c]; x0[c] = 0.0; g0[c] = 0.0; }
  if (boxed) {
    ASLR_UNROLL for (int c = 0; c < NU; ++c) x0[c] = fmax(fmin(k0[c], ub[c]), lb[c]);
    ASLR_UNROLL for (int c = 0; c < NU; ++c) {
      double sg = qu[c];
      ASLR_UNROLL for (int e = 0; e < NU; ++e) sg += Quu[c][e] * x0[e];
      g0[c] = sg;
      any_clamped = any_clamped | ((x0[c] == lb[c]) & (sg > 0.0)) | ((x0[c] == ub[c]) & (sg < 0.0));
      gnorm0 = fmax(gnorm0, fabs(sg));
    }
  }
  const bool need_plain = !boxed || !any_clamped;
  bool plain_bad = false;
  ASLR_UNROLL for (int c = 0; c < NU; ++c) { kv[c] = 0.0; Kc[c] = 0.0; }
  if (__ballot(need_plain) != 0ull) {
    double L[NU][NU], rinv[NU];
    ASLR_UNROLL for (int c = 0; c < NU; ++c)
      ASLR_UNROLL for (int e = 0; e < NU; ++e) L[c][e] = Quu[c][e];
    plain_bad = chol_rs<NU>(L, rinv);
    ASLR_UNROLL for (int c = 0; c < NU; ++c) { kv[c] = qu[c]; Kc[c] = Qux[c]; }
    chol_solve_r<NU>(L, rinv, kv);
    chol_solve_r<NU>(L, rinv, Kc);
  }
  if (!boxed) {
    if (plain_bad) failed = true;
  } else {
    bool interior = !any_clamped && !plain_bad && sp.boxqp_reg == 0.0;
    ASLR_UNROLL for (int c = 0; c < NU; ++c) {
      const double dlt = -kv[c], mrg = 1e-9 * (1.0 + fabs(dlt));
      interior = interior & (dlt > lb[c] + mrg) & (dlt < ub[c] - mrg);
    }
    if (!any_clamped && !plain_bad && sp.boxqp_reg == 0.0 && gnorm0 <= sp.boxqp_th_grad) {
      ASLR_UNROLL for (int c = 0; c < NU; ++c) kv[c] = -x0[c]; // (a)
    } else if (interior) {
      // (b): kv, Kc already hold the result
    } else {
      double xq[NU];
      bool cm[NU];
      ASLR_UNROLL for (int c = 0; c < NU; ++c) { xq[c] = x0[c]; Kc[c] = Qux[c]; }
      if (boxqp<NU>(Quu, qu, lb, ub, xq, g0, cm, Kc, sp.boxqp_maxiter, sp.boxqp_th_acceptstep, sp.boxqp_th_grad, sp.boxqp_reg
#ifdef ASLR_BWD_PROFILE
                    , prof_acc, prof_last
#endif
                    )) failed = true;
      ASLR_UNROLL for (int c = 0; c < NU; ++c) {
        kv[c] = -xq[c];
        if (cm[c]) qu[c] = 0.0;
      }
    }
  }
  return failed;
}

// TPWA: teams actually used per wave (<= 64 / TEAM).  Fewer teams per wave means more waves (idle issue
// slots are plentiful at 4096 trajectories per GPU) and less lock-step waste in the per-team BoxQP loops.
template <int NX, int NU, int HS, int TPWA = 0>
struct BwdCfg {
  static constexpr int NXP = NX <= 8 ? 8 : 32;
  static constexpr int TEAM = NXP * HS;
  static constexpr int TPW = (TPWA > 0 && TPWA < 64 / TEAM) ? TPWA : 64 / TEAM;
  static constexpr int RPL = (NX + HS - 1) / HS;
  static constexpr bool EXACT = (RPL * HS == NX); // rows divide evenly: no clamping of row indices
  static constexpr int REC = rec_len_c(NX, NU);
  static constexpr int oFx = 0, oFu = oFx + NX * NX, oLxx = oFu + NX * NU, oLxu = oLxx + NX * NX,
                       oLuu = oLxu + NX * NU, oLx = oLuu + NU * NU, oLu = oLx + NX;
  // LDS arrays per team (doubles)
  static constexpr int sRec = 0, sAT = sRec + REC, sBT = sAT + NX * NX, sQux = sBT + NX * NU,
                       sVT = sQux + NU * NX, sQuu = sVT + NX * NX, sQu = sQuu + NU * NU, sVx = sQu + NU,
                       sEnd = sVx + NX;
  static constexpr int LDS_TEAM = (sEnd + 1) / 2 * 2;
  static constexpr int NPRE = (REC / 2 + TEAM - 1) / TEAM; // double2 prefetch registers per lane
  // the Fu products (Fu^T P, (Fu^T P) Fx, (Fu^T P) Fu) need all nu rows of a column: with HS lanes per column each
  // lane takes NU / HS of them (when that divides) instead of every lane computing all of them
  static constexpr bool SPLITU = HS > 1 && NU % HS == 0;
  static constexpr int NUH = SPLITU ? NU / HS : NU;
  // DMA: the record of the next knot (and its [us | k | gap] inputs) goes from HBM straight into LDS
  // (global_load_lds_dwordx4: every lane fetches 16 bytes, the wave's 64 pieces land contiguously), issued once
  // this knot's record has been consumed, so no prefetch registers stay live across the gains / box-QP phase.
  // One instruction covers a block of BS doubles of each team's record: element e of team tm sits at
  // (e / BS) * DMAW + tm * BS + e % BS.
  static constexpr bool DMA = (NX == 8 && HS >= 2 && TPW == 64 / TEAM && NU % 2 == 0);
  // nu = 4 gains and box QP spread over the lanes of a 16-lane DPP row (aslr_team_gains.hpp) instead of every lane
  // running the whole 4x4 problem on register arrays
  static constexpr bool TEAMQP = DMA && NU == 4 && TEAM % 16 == 0;
  static constexpr int BS = 2 * TEAM, DMAW = 128;
  static constexpr int AUXL = NU + NX / 2; // lanes of a team that fetch the small inputs
  static constexpr int ridx(int e) { return (e / BS) * DMAW + e % BS; }
  // every (constant + per-lane offset) access of the kernel must stay inside one block
  static constexpr bool in_block(int cst, int rtmax) { return cst % BS + rtmax < BS; }
  static constexpr bool dma_layout_ok() {
    bool ok = (REC % 2 == 0) && (AUXL <= TEAM); // (the Lxx rows are read through per-lane offsets: lxx_idx)
    for (int l = 0; l < NX; ++l) {
      for (int i = 0; i < RPL; ++i) ok = ok && in_block(oFx + l * NX + i, (HS - 1) * RPL);
      for (int c = 0; c < NUH; ++c) ok = ok && in_block(oFu + l * NU + c, (HS - 1) * NUH);
      ok = ok && in_block(oFx + l * NX, NX - 1) && in_block(oFu + l * NU, NU - 1);
    }
    ok = ok && in_block(oLx, NX - 1) && in_block(oLu, NU - 1);
    for (int i = 0; i < RPL; ++i) ok = ok && in_block(oLxx + i * NX, NX - 1);
    for (int c = 0; c < NUH; ++c)
      ok = ok && in_block(oLxu + c, (NX - 1) * NU + (HS - 1) * NUH) && in_block(oLuu + c * NU, (HS - 1) * NUH * NU + NU - 1);
    return ok;
  }
  static_assert(!DMA || dma_layout_ok(), "record layout does not fit the LDS-DMA blocks");
};

// One record: piece I of every team with instruction offset I * BS * 8, which moves the global address by I blocks
// and the LDS address by the same number of bytes (hence the DMAW - BS stride of the bases).
template <class C, int... I>
ASLR_DEV void dma_record(const char *g, unsigned recD_addr, int lt, std::integer_sequence<int, I...>) {
  ((C::NPRE * C::TEAM == C::REC / 2 || lt + C::TEAM * I < C::REC / 2 ? dma16<I * C::BS * 8>(g, recD_addr + I * (C::DMAW - C::BS) * 8) : (void)0), ...);
}

// BOX: SolverBoxDDP gains may be needed (solver is BoxDDP); GAPS: infeasible candidates may be present
// (gap terms, FDDP expected-improvement terms).  A wave whose trajectories need neither runs the lean variant.
#ifndef ASLR_BWD_WAVES_TEAMQP
#define ASLR_BWD_WAVES_TEAMQP 1 // register budget of the 16-lane-team nu = 4 kernels as waves per SIMD (2: <= 256, so that calc / cost waves of other sub-shards fit next to a sweep wave)
#endif
#ifndef ASLR_BWD_WAVES
#define ASLR_BWD_WAVES 2 // waves per SIMD the HS = 4 variant (two 32-lane teams per wave) must fit
#endif
template <int NX, int NU, int HS, int TPWA, bool BOX, bool GAPS>
__global__ void __launch_bounds__(64, (HS >= 4 ? ASLR_BWD_WAVES : (BwdCfg<NX, NU, HS, TPWA>::TEAMQP ? ASLR_BWD_WAVES_TEAMQP : 1))) backward_kernel(KArgs a, SolverDev sp, ModelLimits lim) {
  using C = BwdCfg<NX, NU, HS, TPWA>;
  constexpr int NXP = C::NXP, TEAM = C::TEAM, TPW = C::TPW, RPL = C::RPL, REC = C::REC;
  // The work arrays are a static allocation in the DMA configuration: the compiler then knows they cannot overlap
  // the DMA targets and does not drain vmcnt before reading them while the next record is in flight.
  extern __shared__ __attribute__((aligned(16))) double smem_dyn[];
  __shared__ __attribute__((aligned(16))) double smem_sta[C::DMA ? C::TPW * C::LDS_TEAM : 2];
  double *smem = C::DMA ? smem_sta : smem_dyn;

  const int lane = threadIdx.x, team = lane / TEAM, lt = lane % TEAM, j = lt % NXP, h = lt / NXP;
  const int B = a.B, T = a.T;
  const int bq = a.b0 + blockIdx.x * TPW + team;
  const bool team_valid = team < TPW && bq < a.b1;
  const int b = team_valid ? bq : a.b1 - 1;
  const bool col_valid = j < NX;
  const int jj = col_valid ? j : NX - 1;
  const bool writer = team_valid && col_valid && h == 0;
  const int r0 = h * RPL;
  const int ju = jj < NU ? jj : NU - 1;
  double *sm = smem + (team < TPW ? team : 0) * C::LDS_TEAM;
  double *rec = sm + C::sRec, *AT = sm + C::sAT, *BT = sm + C::sBT, *QuxL = sm + C::sQux, *VT = sm + C::sVT,
         *QuuL = sm + C::sQuu, *QuL = sm + C::sQu, *VxL = sm + C::sVx;
  // two record buffers (by parity of the knot): the DMA of knot t - 1 is issued at the TOP of knot t, a whole knot ahead
  __shared__ __attribute__((aligned(16))) double recD[C::DMA ? 2 * C::NPRE * C::DMAW : 2];
  __shared__ __attribute__((aligned(16))) double auxD[C::DMA ? 2 * C::DMAW : 2];
  const double *recT = recD + (C::DMA ? team * C::BS : 0); // (re-pointed to the knot's buffer at the top of every knot)
  // record element (compile-time part cst, per-lane part rt)
  auto R = [&](int cst, int rt) -> double { return C::DMA ? recT[C::ridx(cst) + rt] : rec[cst + rt]; };
  auto row = [&](int i) { return C::EXACT ? r0 + i : (r0 + i < NX ? r0 + i : NX - 1); };
  auto row_ok = [&](int i) { return C::EXACT ? true : (r0 + i < NX); };

  int32_t *TI = a.traj_i;
  double *TF = a.traj_f;
  ASLR_STAMP_BEGIN(a, 1);
#ifdef ASLR_EXP_WAVETIME
  const long long exp_t0 = wall_clock64();
#endif

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
    wave_sync();
    if (lt == 0 && team_valid) {
      TI[ASLR_TI_FEASIBLE * B + b] = feasible;
      TI[ASLR_TI_ACCEPTED * B + b] = -1;
      TI[ASLR_TI_GAPFLAG * B + b] = 0;
    }
  }
  bool need = team_valid && !done;
  if (__ballot(need) == 0ull) return;
  double xreg = TF[ASLR_TF_XREG * B + b];
  const bool fddp = GAPS && sp.solver == ASLR_SOLVER_FDDP;
  const bool box = BOX && sp.solver == ASLR_SOLVER_BOXDDP && feasible;
  const bool gaps_on = GAPS && !feasible;
  const unsigned long long team_mask = (TEAM == 64 ? ~0ull : ((1ull << TEAM) - 1ull)) << (team * TEAM);

  // one-hot selectors of this lane's diagonal / control row (FMA instead of compare + select in the loop)
  constexpr int NUH = C::NUH;
  const int c0 = C::SPLITU ? h * NUH : 0; // first of this lane's control rows in the Fu products
  double oh_row[RPL], oh_u[NU], oh_u0[NUH];
  ASLR_UNROLL for (int c = 0; c < NUH; ++c) oh_u0[c] = (c0 + c == j) ? 1.0 : 0.0;
  ASLR_UNROLL for (int i = 0; i < RPL; ++i) oh_row[i] = (r0 + i == jj) ? 1.0 : 0.0;
  ASLR_UNROLL for (int c = 0; c < NU; ++c) oh_u[c] = (c == j) ? 1.0 : 0.0;

  // LDS index of Lxx(r0 + i, jj) inside this team's DMA blocks
  int lxx_idx[RPL];
  ASLR_UNROLL for (int i = 0; i < RPL; ++i) {
    const int e = C::oLxx + (r0 + i) * NX + jj;
    lxx_idx[i] = C::DMA ? (e / C::BS) * C::DMAW + e % C::BS : 0;
  }
  // team-distributed gains (nu = 4): this lane is row qr of the 4x4 problem; control limits of every model in LDS so
  // that a lane reads ITS bound with one load ([model][lb 0..3 | ub 0..3])
  const int qr = lt & 3;
  double oh4[4];
  ASLR_UNROLL for (int c = 0; c < 4; ++c) oh4[c] = (c == qr) ? 1.0 : 0.0;
  __shared__ double limL[C::TEAMQP && BOX ? ASLR_MAX_MODELS * 8 : 1];
  if (C::TEAMQP && BOX) {
    ASLR_UNROLL for (int m = 0; m < ASLR_MAX_MODELS; ++m)
      ASLR_UNROLL for (int c = 0; c < 4; ++c) {
        limL[m * 8 + c] = lim.lb[m][c]; // (every lane writes the same values)
        limL[m * 8 + 4 + c] = lim.ub[m][c];
      }
    wave_sync();
  }
  const TeamQPParams qpp{sp.boxqp_maxiter, sp.boxqp_th_acceptstep, sp.boxqp_th_grad, sp.boxqp_reg, ASLR_NALPHA};

  double d1 = 0.0, d2 = 0.0, stop = 0.0, dgf = 0.0, dqf = 0.0;
  ASLR_PROF_DECL;
#ifdef ASLR_BWD_PROFILE
  if (C::TEAMQP && threadIdx.x == 0) { for (int i = 0; i < 16; ++i) tg_prof()[i] = 0; }
#endif
  while (__ballot(need) != 0ull) {
    bool failed = false;
    d1 = d2 = stop = dgf = dqf = 0.0;
    const double xr = isnan(xreg) ? 0.0 : xreg;
    double Pcol[NX], pvec[NX], Vx_own;
    // ---- terminal node: Vxx = Lxx (+xreg), Vx = Lx (+ Vxx f) ----
    {
      const double *rT = a.deriv + ((size_t)T * B + b) * REC;
      ASLR_UNROLL for (int r = 0; r < NX; ++r) Pcol[r] = rT[C::oLxx + r * NX + jj] + (r == jj ? xr : 0.0);
      Vx_own = rT[C::oLx + jj];
      if (gaps_on) {
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
      if (!C::TEAMQP) { // (there Vx stays spread over lanes 0..7 of the row and reaches the products through DPP)
        wave_sync();
        VxL[jj] = Vx_own;
        wave_sync();
        ASLR_UNROLL for (int l = 0; l < NX; ++l) pvec[l] = VxL[l];
      }
    }
    // ---- prefetch for knot T-1: record, model index, box-QP inputs, gap ----
    double prx[C::NPRE], pry[C::NPRE];
    double pre_u[NU], pre_k[NU], pre_f[NX];
    int pre_m = 0;
    ASLR_UNROLL for (int c = 0; c < NU; ++c) { pre_u[c] = 0.0; pre_k[c] = 0.0; }
    ASLR_UNROLL for (int r = 0; r < NX; ++r) pre_f[r] = 0.0;
    // (a macro, not a by-reference lambda: the closure kept `pre` in scratch memory)
#define ASLR_BWD_PREFETCH(tt)                                                                          \
    do {                                                                                               \
      const size_t tbp = (size_t)(tt) * B + b;                                                         \
      typedef double nt_double2 __attribute__((ext_vector_type(2)));                                   \
      const nt_double2 *src = reinterpret_cast<const nt_double2 *>(a.deriv + tbp * REC);               \
      ASLR_UNROLL for (int i = 0; i < C::NPRE; ++i) {                                                  \
        const int idx = lt + TEAM * i;                                                                 \
        if (C::NPRE * TEAM == REC / 2 || idx < REC / 2) { const nt_double2 v2 = __builtin_nontemporal_load(src + idx); prx[i] = v2.x; pry[i] = v2.y; } \
      }                                                                                                \
      pre_m = node_model_at(a, tt);                                                                        \
      if (box) {                                                                                       \
        ASLR_UNROLL for (int c = 0; c < NU; ++c) { pre_u[c] = a.us[tbp * NU + c]; pre_k[c] = a.kff[tbp * NU + c]; } \
      }                                                                                                \
      if (gaps_on) {                                                                                   \
        ASLR_UNROLL for (int r = 0; r < NX; ++r) pre_f[r] = a.gaps[tbp * NX + r];                      \
      }                                                                                                \
    } while (0)
    // DMA variant: record pieces and the small inputs of knot tt straight into LDS (aux slot by parity of tt)
#define ASLR_BWD_DMA(tt)                                                                               \
    do {                                                                                               \
      const char *kb = reinterpret_cast<const char *>(a.deriv + (size_t)(tt) * B * REC); /* uniform */ \
      dma_record<C>(kb + rec_voff, lds_address(recD) + ((tt) & 1) * (C::NPRE * C::DMAW * 8), lt, std::make_integer_sequence<int, C::NPRE>()); \
      if (aux_on) dma16<0>(reinterpret_cast<const char *>(aux_src + (size_t)(tt) * aux_step), lds_address(auxD) + ((tt) & 1) * C::DMAW * 8); \
      pre_m = node_model_at(a, tt);                                                                        \
    } while (0)
    // per-lane byte offset of this lane's 16-byte piece inside a knot's slab of records
    const size_t rec_voff = (size_t)b * (size_t)(REC * 8) + (size_t)lt * 16u;
    // which of the small inputs this lane fetches: lanes [0, nu/2) us, [nu/2, nu) k, [nu, nu + nx/2) the gap
    const size_t aux_stride = lt < NU ? NU : NX;
    const double *aux_src = (lt < NU / 2 ? a.us + 2 * lt : (lt < NU ? a.kff + 2 * (lt - NU / 2) : a.gaps + 2 * (lt - NU))) + (size_t)b * aux_stride;
    const size_t aux_step = (size_t)B * aux_stride; // doubles per knot
    const bool aux_on = lt < NU ? box : (gaps_on && lt < C::AUXL);
    // results of a knot: K / k / Qu (state of the gains phase), Vx / Vxx / Vxx f (state after step 4).  In the DMA
    // configurations they are issued at the top of the NEXT knot of the sweep (and after the last one).
    bool st_k = false, st_v = false, st_f = false;
    double Kc_st[NU], kj_st = 0.0, qj_st = 0.0, vf_st = 0.0;
    ASLR_UNROLL for (int c = 0; c < NU; ++c) Kc_st[c] = 0.0;
#define ASLR_BWD_STORES_K(tbs)                                                                         \
    do {                                                                                               \
      if (st_k) {                                                                                      \
        double *Kout = a.kgain + (tbs) * NU * NX;                                                      \
        ASLR_UNROLL for (int c = 0; c < NU; ++c) Kout[c * NX + jj] = Kc_st[c];                         \
        if (j < NU) {                                                                                  \
          a.kff[(tbs) * NU + j] = kj_st;                                                               \
          a.qu[(tbs) * NU + j] = qj_st;                                                                \
        }                                                                                              \
      }                                                                                                \
    } while (0)
#define ASLR_BWD_STORES_V(tbs)                                                                         \
    do {                                                                                               \
      if (st_f) a.vxxf[(tbs) * NX + jj] = vf_st;                                                       \
      if (st_v) {                                                                                      \
        a.vx[(tbs) * NX + jj] = Vx_own;                                                                \
        double *o = a.vxx + (tbs) * NX * NX;                                                           \
        ASLR_UNROLL for (int r = 0; r < NX; ++r) o[r * NX + jj] = Pcol[r];                             \
      }                                                                                                \
    } while (0)
#define ASLR_BWD_STORES(tbs) do { ASLR_BWD_STORES_K(tbs); ASLR_BWD_STORES_V(tbs); } while (0)
    if (C::DMA) ASLR_BWD_DMA(T - 1); else ASLR_BWD_PREFETCH(T - 1);
    for (int t = T - 1; t >= 0; --t) {
      const size_t tb = (size_t)t * B + b;
      ASLR_PROF(11);
      ASLR_PROF_COUNT(15);
      const double *auxT = auxD + (C::DMA ? (t & 1) * C::DMAW + team * C::BS : 0); // [us | k | gap] of this knot
      double ut[NU], k0[NU], fg[NX];
      const int mi = pre_m; // (before the next knot's loads are issued: they fetch ITS model index into pre_m)
      if (C::DMA) {
        // the record of this knot (issued at the top of the previous one) has landed, and the stores issued there are done
        wait_vmcnt<0>();
        wave_sync();
        recT = recD + (t & 1) * (C::NPRE * C::DMAW) + team * C::BS;
        // results of knot t + 1, then the loads of knot t - 1: every vector-memory operation of the sweep gets a whole
        // knot before anything waits for it
        ASLR_BWD_STORES(tb + B);
        if (t > 0) ASLR_BWD_DMA(t - 1);
      } else {
        // stage the record in LDS, take this knot's small inputs, start the next loads
        double2 *dst = reinterpret_cast<double2 *>(rec);
        ASLR_UNROLL for (int i = 0; i < C::NPRE; ++i) {
          const int idx = lt + TEAM * i;
          if (C::NPRE * TEAM == REC / 2 || idx < REC / 2) { double2 v2; v2.x = prx[i]; v2.y = pry[i]; dst[idx] = v2; }
        }
        ASLR_UNROLL for (int c = 0; c < NU; ++c) { ut[c] = pre_u[c]; k0[c] = pre_k[c]; }
        ASLR_UNROLL for (int r = 0; r < NX; ++r) fg[r] = pre_f[r];
      }
      if (!C::DMA) {
        wave_sync();
        if (t > 0) ASLR_BWD_PREFETCH(t - 1);
      }

      ASLR_PROF(0);
      // ---- step 1: A = Fx^T P (my rows of column jj), Bc = Fu^T P (column jj), Qx, Qu ----
      double Fxcol[NX], Fucol[NX];
      ASLR_UNROLL for (int l = 0; l < NX; ++l) { Fxcol[l] = R(C::oFx + l * NX, jj); Fucol[l] = R(C::oFu + l * NU, ju); }
      {
        double Arow[RPL], Bc[NUH];
        ASLR_UNROLL for (int i = 0; i < RPL; ++i) Arow[i] = 0.0;
        ASLR_UNROLL for (int c = 0; c < NUH; ++c) Bc[c] = 0.0;
        ASLR_UNROLL for (int l = 0; l < NX; ++l) {
          ASLR_UNROLL for (int i = 0; i < RPL; ++i) Arow[i] += (C::DMA ? R(C::oFx + l * NX + i, r0) : rec[C::oFx + l * NX + row(i)]) * Pcol[l];
          ASLR_UNROLL for (int c = 0; c < NUH; ++c) Bc[c] += R(C::oFu + l * NU + c, c0) * Pcol[l];
        }
        ASLR_UNROLL for (int i = 0; i < RPL; ++i) if (row_ok(i)) AT[jj * NX + r0 + i] = Arow[i];
        ASLR_UNROLL for (int c = 0; c < NUH; ++c) BT[jj * NU + c0 + c] = Bc[c];
      }
      double Qx, Qu_own;
      {
        double s = 0.0, s2 = 0.0;
        if constexpr (C::TEAMQP) {
          DevTeamOps::dot8_acc(s, Vx_own, Fxcol); // lane l of the row holds Vx[l]
          DevTeamOps::dot8_acc(s2, Vx_own, Fucol);
        } else {
          ASLR_UNROLL for (int l = 0; l < NX; ++l) { s += Fxcol[l] * pvec[l]; s2 += Fucol[l] * pvec[l]; }
        }
        Qx = R(C::oLx, jj) + s;
        Qu_own = R(C::oLu, ju) + s2;
      }
      wave_sync();
      ASLR_PROF(1);
      // ---- step 2: Qxx (my rows), Qux (column jj), Quu (column jj < NU) ----
      double Qxx[RPL], Qux[NU];
      {
        double acc[RPL], accu[NUH], accq[NUH];
        ASLR_UNROLL for (int i = 0; i < RPL; ++i) acc[i] = 0.0;
        ASLR_UNROLL for (int c = 0; c < NUH; ++c) { accu[c] = 0.0; accq[c] = 0.0; }
        ASLR_UNROLL for (int l = 0; l < NX; ++l) {
          ASLR_UNROLL for (int i = 0; i < RPL; ++i) acc[i] += AT[l * NX + row(i)] * Fxcol[l];
          ASLR_UNROLL for (int c = 0; c < NUH; ++c) {
            const double bt = BT[l * NU + c0 + c];
            accu[c] += bt * Fxcol[l];
            accq[c] += bt * Fucol[l];
          }
        }
        ASLR_UNROLL for (int i = 0; i < RPL; ++i) Qxx[i] = (C::DMA ? recT[lxx_idx[i]] : rec[C::oLxx + row(i) * NX + jj]) + acc[i];
        ASLR_UNROLL for (int c = 0; c < NUH; ++c) QuxL[(c0 + c) * NX + jj] = R(C::oLxu + c, jj * NU + c0) + accu[c];
        if (j < NU) {
          ASLR_UNROLL for (int c = 0; c < NUH; ++c)
            QuuL[(c0 + c) * NU + j] = R(C::oLuu + c * NU, c0 * NU + j) + accq[c] + oh_u0[c] * xr;
          QuL[j] = Qu_own;
        }
      }
      wave_sync();
      ASLR_UNROLL for (int c = 0; c < NU; ++c) Qux[c] = QuxL[c * NX + jj];
      ASLR_PROF(2);
      if (C::DMA && !C::TEAMQP) {
        ASLR_UNROLL for (int c = 0; c < NU; ++c) { ut[c] = auxT[c]; k0[c] = auxT[NU + c]; } // (used by box nodes only)
      }
      ASLR_PROF(3);
      // ---- step 3: gains ----
      double Kc[NU];
      double kj_out = 0.0, qj_out = 0.0; // this lane's entries of k and Qu (lanes j < NU of the writer set)
      if constexpr (C::TEAMQP) {
        // one entry of every 4-vector and one row of Quu per lane; broadcasts through DPP (aslr_team_gains.hpp)
        double Hr[4];
        {
          const double2 h01 = *reinterpret_cast<const double2 *>(QuuL + qr * NU), h23 = *reinterpret_cast<const double2 *>(QuuL + qr * NU + 2);
          Hr[0] = h01.x; Hr[1] = h01.y; Hr[2] = h23.x; Hr[3] = h23.y;
        }
        const double q_own = QuL[qr];
        const bool boxed = box && lim.has[mi];
        double lbr = 0.0, ubr = 0.0, k0r = 0.0;
        if (BOX) {
          const double utr = auxT[qr];
          k0r = auxT[NU + qr];
          lbr = limL[mi * 8 + qr] - utr;
          ubr = limL[mi * 8 + 4 + qr] - utr;
        }
        ASLR_PROF(4);
        TeamFactor<DevTeamOps> F;
        double kv_own, qz_own;
        bool gbad;
        team_gains4<DevTeamOps, BOX>(Hr, q_own, boxed, lbr, ubr, k0r, oh4, qpp, kv_own, qz_own, F, gbad);
        if (gbad) failed = true;
        ASLR_PROF(5);
        ASLR_UNROLL for (int c = 0; c < NU; ++c) Kc[c] = Qux[c];
        team_gain_column<DevTeamOps>(F, Kc);
        ASLR_PROF(16);
        // Quu k (row qr), expected-improvement terms, Vx
        double Quuk = 0.0;
        DevTeamOps::matvec_acc<false>(Quuk, kv_own, Hr);
        {
          const double one = 1.0, t1 = qz_own * kv_own, t2 = kv_own * Quuk, t3 = qz_own * qz_own;
          DevTeamOps::acc3(d1, t1, d2, t2, stop, t3, one);
        }
        {
          double s = 0.0, s2 = 0.0;
          DevTeamOps::matvec_acc<false>(s, Quuk, Kc);
          DevTeamOps::matvec_acc<false>(s2, qz_own, Kc);
          Vx_own = Qx + s - 2.0 * s2;
        }
        kj_out = kv_own; // (lanes j < 4 have qr = j)
        qj_out = qz_own;
      } else {
      double Quu[NU][NU], qu[NU], kv[NU];
      ASLR_UNROLL for (int c = 0; c < NU; ++c) {
        qu[c] = QuL[c];
        ASLR_UNROLL for (int e = 0; e < NU; ++e) Quu[c][e] = QuuL[c * NU + e];
      }
      // SolverBoxDDP::computeGains starts by replaying BoxQP's first iteration exactly (clamped warm start
      // x0, gradient, active set): cheap, and it tells which teams need the plain gains at all.
      const bool boxed = box && lim.has[mi];
      double lb[NU], ub[NU], x0[NU];
      bool any_clamped = false;
      double gnorm0 = 0.0, g0[NU];
      if (boxed) {
        ASLR_UNROLL for (int c = 0; c < NU; ++c) {
          lb[c] = lim.lb[mi][c] - ut[c];
          ub[c] = lim.ub[mi][c] - ut[c];
          x0[c] = fmax(fmin(k0[c], ub[c]), lb[c]);
        }
        ASLR_UNROLL for (int c = 0; c < NU; ++c) {
          double sg = qu[c];
          ASLR_UNROLL for (int e = 0; e < NU; ++e) sg += Quu[c][e] * x0[e];
          g0[c] = sg;
          any_clamped = any_clamped | ((x0[c] == lb[c]) & (sg > 0.0)) | ((x0[c] == ub[c]) & (sg < 0.0));
          gnorm0 = fmax(gnorm0, fabs(sg));
        }
      }
      // plain DDP gains K = Quu^-1 Qux, k = Quu^-1 Qu: needed by unconstrained nodes and by box nodes whose
      // first active set is empty; skipped (wave-uniformly) when every team of the wave runs the QP
      ASLR_PROF(4);
      const bool need_plain = !boxed || !any_clamped;
      bool plain_bad = false;
      ASLR_UNROLL for (int c = 0; c < NU; ++c) { kv[c] = 0.0; Kc[c] = 0.0; }
      if (__ballot(need_plain) != 0ull) {
        ASLR_PROF_COUNT(14);
        double L[NU][NU], rinv[NU];
        ASLR_UNROLL for (int c = 0; c < NU; ++c)
          ASLR_UNROLL for (int e = 0; e < NU; ++e) L[c][e] = Quu[c][e];
        plain_bad = chol_rs<NU>(L, rinv);
        ASLR_UNROLL for (int c = 0; c < NU; ++c) { kv[c] = qu[c]; Kc[c] = Qux[c]; }
        chol_solve_r<NU>(L, rinv, kv);
        chol_solve_r<NU>(L, rinv, Kc);
      }
      ASLR_PROF(5);
      if (!boxed) {
        if (plain_bad) failed = true;
      } else {
        // When no index is clamped at x0, the QP iteration has two cheap outcomes that need no
        // projected-Newton loop:
        //   (a) |g(x0)|_inf <= th_grad: BoxQP returns x0 itself with every index free;
        //   (b) otherwise it takes the full Newton step to the unconstrained minimiser -Quu^-1 Qu; when
        //       that point is strictly inside the box the step is accepted at alpha = 1 and the next
        //       gradient test passes, again with every index free.
        // In both, Hff^-1 = Quu^-1, so K is the plain gain above.  Everything else runs BoxQP.
        bool interior = !any_clamped && !plain_bad && sp.boxqp_reg == 0.0;
        ASLR_UNROLL for (int c = 0; c < NU; ++c) {
          const double dlt = -kv[c], mrg = 1e-9 * (1.0 + fabs(dlt));
          interior = interior & (dlt > lb[c] + mrg) & (dlt < ub[c] - mrg); // (plain &: no short-circuit branches)
        }
        if (!any_clamped && !plain_bad && sp.boxqp_reg == 0.0 && gnorm0 <= sp.boxqp_th_grad) {
          ASLR_UNROLL for (int c = 0; c < NU; ++c) kv[c] = -x0[c]; // (a)
        } else if (interior) {
          // (b): kv, Kc already hold the result
        } else {
          double xq[NU];
          bool cm[NU];
          ASLR_UNROLL for (int c = 0; c < NU; ++c) { xq[c] = x0[c]; Kc[c] = Qux[c]; }
          ASLR_PROF_COUNT(13);
          if (boxqp<NU>(Quu, qu, lb, ub, xq, g0, cm, Kc, sp.boxqp_maxiter, sp.boxqp_th_acceptstep, sp.boxqp_th_grad,
                        sp.boxqp_reg
#ifdef ASLR_BWD_PROFILE
                        , prof_acc, prof_last
#endif
                        )) failed = true;
          ASLR_UNROLL for (int c = 0; c < NU; ++c) {
            kv[c] = -xq[c];
            if (cm[c]) qu[c] = 0.0;
          }
        }
      }
      ASLR_PROF(16); // (a team that sits out the QP of its wave mates spends their QP time here)
      if (C::DMA) {
        ASLR_UNROLL for (int r = 0; r < NX; ++r) fg[r] = 0.0;
        if (gaps_on) {
          ASLR_UNROLL for (int r = 0; r < NX; ++r) fg[r] = auxT[2 * NU + r];
        }
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
        if (j < NU) {
          ASLR_UNROLL for (int c = 0; c < NU; ++c) { kj_out += oh_u[c] * kv[c]; qj_out += oh_u[c] * qu[c]; }
        }
      }
      {
        double acc[RPL];
        ASLR_UNROLL for (int i = 0; i < RPL; ++i) acc[i] = 0.0;
        ASLR_UNROLL for (int c = 0; c < NU; ++c)
          ASLR_UNROLL for (int i = 0; i < RPL; ++i) acc[i] += QuxL[c * NX + row(i)] * Kc[c];
        // Vxx (unsymmetrised) with the state regularisation already on its diagonal: the average
        // below leaves a diagonal entry d + xreg untouched
        ASLR_UNROLL for (int i = 0; i < RPL; ++i)
          if (row_ok(i)) VT[jj * NX + r0 + i] = (Qxx[i] - acc[i]) + oh_row[i] * xr;
      }
      const bool st_ok = writer && need && !failed;
      st_k = st_ok;
      ASLR_UNROLL for (int c = 0; c < NU; ++c) Kc_st[c] = Kc[c];
      kj_st = kj_out; qj_st = qj_out;
      if (!C::DMA) ASLR_BWD_STORES_K(tb);
      wave_sync();
      ASLR_PROF(10);
      // ---- step 4: symmetrise (column jj of the symmetric Vxx = its row jj), gap term, publish Vx ----
      double chk = 0.0;
      ASLR_UNROLL for (int r = 0; r < NX; ++r) {
        Pcol[r] = 0.5 * (VT[jj * NX + r] + VT[r * NX + jj]);
        chk += fabs(Pcol[r]);
      }
      if (gaps_on) {
        if (C::TEAMQP) { // (the gap of this knot is still in its aux slot: the next DMA writes the other one)
          ASLR_UNROLL for (int r = 0; r < NX; ++r) fg[r] = auxT[2 * NU + r];
        }
        double vf = 0.0;
        ASLR_UNROLL for (int r = 0; r < NX; ++r) vf += Pcol[r] * fg[r];
        Vx_own += vf;
        if (fddp) {
          double fj = fg[0];
          ASLR_UNROLL for (int r = 1; r < NX; ++r) if (r == jj) fj = fg[r];
          dgf -= Vx_own * fj;
          dqf += fj * vf;
          vf_st = vf;
        }
      }
      chk += fabs(Vx_own);
      // NaN / Inf / >= 1e30 anywhere in Vx, Vxx -> "backward_error" (Crocoddyl's inf-norm test: the 1-norm of the
      // column clears the common case, the entries decide otherwise)
      {
        double ent[NX + 1];
        ASLR_UNROLL for (int r = 0; r < NX; ++r) ent[r] = Pcol[r];
        ent[NX] = Vx_own;
        if (__ballot(inf_norm_bad<NX + 1>(chk, ent)) & team_mask) failed = true;
      }
      st_v = sp.store_v && st_ok && !failed;
      st_f = gaps_on && fddp && st_ok;
      if (!C::DMA) ASLR_BWD_STORES_V(tb);
      if (!C::TEAMQP) {
        VxL[jj] = Vx_own;
        wave_sync();
        ASLR_UNROLL for (int l = 0; l < NX; ++l) pvec[l] = VxL[l];
      }
    }
    if (C::DMA) ASLR_BWD_STORES((size_t)b); // knot 0
    ASLR_PROF(11);
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
          // (a regularisation that cannot grow -- zero, NaN, a factor <= 1 -- would redo the sweep for ever: Crocoddyl's own
          //  loop has the same property; here it counts as the ceiling, so that the kernel always terminates)
          const double grown = xreg * sp.reg_incfactor;
          xreg = (grown > xreg) ? grown : sp.reg_max;
          if (xreg > sp.reg_max) xreg = sp.reg_max;
          if (!(xreg < sp.reg_max)) { // (== reg_max; also a NaN ceiling ends the retries)
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
#ifdef ASLR_BWD_PROFILE
  if (C::TEAMQP && threadIdx.x == 0) { // team-gains regions 1..5 -> table entries 6..9 (+ 5 stays the total), counters 10..12 -> 13, 12, 14
    long long *g = tg_prof();
    prof_acc[6] += g[1]; prof_acc[7] += g[2]; prof_acc[8] += g[3]; prof_acc[9] += g[4] + g[5];
    prof_acc[13] += g[10]; prof_acc[12] += g[11]; prof_acc[14] += g[12]; prof_acc[17] += g[13];
  }
#endif
  ASLR_PROF_FLUSH;
  ASLR_STAMP_END(1, true);
#ifdef ASLR_EXP_WAVETIME
  // (experiment: how long each wave of the sweep ran, 100 MHz ticks, into the unused head of VX -- tools/wave_times.py)
  if (threadIdx.x == 0) a.vx[blockIdx.x] = (double)(wall_clock64() - exp_t0);
#endif
}

// `all_feasible`: the caller knows every trajectory of the shard is feasible (no gap terms needed)
template <int NX, int NU, int HS, int TPWA = 0>
int launch_backward_t(const KArgs &k, const SolverDev &sd, const ModelLimits &lim, bool all_feasible, hipStream_t st) {
  using C = BwdCfg<NX, NU, HS, TPWA>;
  const int blocks = (k.b1 - k.b0 + C::TPW - 1) / C::TPW;
  const size_t lds = C::DMA ? 0 : (size_t)C::TPW * C::LDS_TEAM * sizeof(double); // (DMA: allocated statically)
  const bool box = sd.solver == ASLR_SOLVER_BOXDDP;
  if (box && !all_feasible) hipLaunchKernelGGL((backward_kernel<NX, NU, HS, TPWA, true, true>), dim3(blocks), dim3(64), lds, st, k, sd, lim);
  else if (box) hipLaunchKernelGGL((backward_kernel<NX, NU, HS, TPWA, true, false>), dim3(blocks), dim3(64), lds, st, k, sd, lim);
  else if (!all_feasible) hipLaunchKernelGGL((backward_kernel<NX, NU, HS, TPWA, false, true>), dim3(blocks), dim3(64), lds, st, k, sd, lim);
  else hipLaunchKernelGGL((backward_kernel<NX, NU, HS, TPWA, false, false>), dim3(blocks), dim3(64), lds, st, k, sd, lim);
  HIP_TRY(hipGetLastError());
  return ASLR_OK;
}

} // namespace aslr
