// aslr_team_gains.hpp -- control gains of ONE knot for nu = 4, spread over the lanes of a 16-lane DPP row
// (SolverDDP::computeGains, SolverBoxDDP::computeGains + BoxQP::solve).
//
// Why this exists: in the backward sweep every lane of a trajectory's team used to run the whole 4x4 problem
// (Cholesky, solves, the projected-Newton box QP) redundantly on register arrays -- ~100 doubles of state and
// ~320 instructions per QP iteration per wave, half of the sweep.  Here lane L of a 16-lane row acts as ROW
// r = L & 3 of the 4x4 problem (the four quads of a row hold identical copies): it keeps row r of Quu and element
// r of every vector (x, g, q, lb, ub, ...).  A matrix-vector product is 4 `v_fmac_f64_dpp ... row_newbcast:c`
// (gfx90a+ "DP ALU DPP": the multiplicand comes from lane c of the row inside the FMA, no shuffle instruction),
// element-wise work is 1 instruction instead of 4, the Cholesky factor is built one column per step with the
// pivot row broadcast the same way.  ~3x fewer instructions and ~1/4 of the registers.
//
// The arithmetic per ENTRY is the arithmetic of the per-lane version it replaces (`boxqp<NU>`, `chol_rs`,
// `chol_solve_r` in aslr_backward.inc.hpp) and of the CPU restatement the tests check against (its BoxQP, chol, chol_solve): same
// operations in the same order; the free / clamped split is carried by exact 0/1 masks as before.
//
// The code is written against an `Ops` policy so that the SAME source runs (a) on the GPU with real = double and
// DPP broadcasts, (b) on the host with real = 16 emulated lanes (tests/host/team_gains_emul.cpp), which is how the
// lane choreography is checked against the oracle's BoxQP without a GPU.
#pragma once

#ifndef ASLR_TG_FN
#define ASLR_TG_FN inline
#endif
#ifndef ASLR_TG_MARK
#define ASLR_TG_MARK(i)  // region-timing hook of the profile build (aslr_team_ops.hpp)
#define ASLR_TG_COUNT(i)
#endif
#if defined(__clang__)
#define ASLR_TG_UNROLL _Pragma("unroll")
#else
#define ASLR_TG_UNROLL _Pragma("GCC unroll 4")
#endif

namespace aslr {

struct TeamQPParams {
  int maxiter;
  double th_acceptstep, th_grad, reg;
  int nalpha;
};

// State of the factorisation of the masked matrix, row-distributed.
template <class O>
struct TeamFactor {
  typename O::real Lr[4];   // row r of L (entries right of the diagonal are don't-cares)
  typename O::real Lc[4];   // column r of L below the diagonal: Lc[k] = L[k][r], k > r (Lc[0] unused)
  typename O::real rinv[4]; // reciprocal pivots, the same in every lane
  typename O::real mkk[4];  // the mask it was built for, the same in every lane: mkk[c] = mk of row c
  typename O::real mkL;     // ... and this lane's own entry of it
};

// masked matrix  mk_r mk_c H_rc + (1 - mk_r) delta_rc  (+ reg on the free diagonal), its Cholesky factor with rsqrt
// pivots (chol_rs), and the column layout the back substitution needs.  Returns "a pivot was not positive".
template <class O>
ASLR_TG_FN typename O::mask team_factor(TeamFactor<O> &F, const typename O::real (&Hr)[4], typename O::real mk,
                                       typename O::real reg, const typename O::real (&oh)[4]) {
  using real = typename O::real;
  using mask = typename O::mask;
  const real one = O::cst(1.0);
  F.mkL = mk;
  O::bc4(mk, F.mkk); // mkk[c] = mk of row c
  const real dterm = mk * reg + (one - mk);
  ASLR_TG_UNROLL for (int c = 0; c < 4; ++c) F.Lr[c] = (mk * F.mkk[c]) * Hr[c] + oh[c] * dterm;
  mask bad = O::mfalse();
  // column j: s_r = A[r][j] - sum_{k<j} L[r][k] L[j][k]  (the pivot row j comes through the broadcast);
  // for r = j this is the pivot d, for r > j the entry before scaling
#define ASLR_TG_COL(j, SUBTRACT)                                                          \
  {                                                                                       \
    real s = F.Lr[j];                                                                     \
    SUBTRACT                                                                              \
    const real d = O::template bc<j>(s);                                                  \
    bad = bad | (!(d > O::cst(0.0)));                                                       \
    const real ri = O::rsqrt(d);                                                          \
    F.rinv[j] = ri;                                                                       \
    F.Lr[j] = s * ri;                                                                     \
  }
#define ASLR_TG_SUB(j, k) O::template fmac_bc<j, true>(s, F.Lr[k], F.Lr[k]);
  ASLR_TG_COL(0, )
  ASLR_TG_COL(1, ASLR_TG_SUB(1, 0))
  ASLR_TG_COL(2, ASLR_TG_SUB(2, 0) ASLR_TG_SUB(2, 1))
  ASLR_TG_COL(3, ASLR_TG_SUB(3, 0) ASLR_TG_SUB(3, 1) ASLR_TG_SUB(3, 2))
#undef ASLR_TG_SUB
#undef ASLR_TG_COL
  // transpose of the strictly lower part: Lc[k] = L[k][r] = sum_c oh[c] * (L[k][c] from lane k), exact
  O::transpose_lower(F.Lr, oh, F.Lc);
  return bad;
}

// L L^T z = b with b distributed (element r in lane r); returns z_r.  Entry-wise the operations of chol_solve_r:
// forward  y_i = (b_i - sum_{k<i} L[i][k] y_k) rinv_i  with k ascending, backward  z_i = (y_i - sum_{k>i} L[k][i] z_k) rinv_i
// with k ascending.  y and z travel as row-uniform values (one broadcast per element).
template <class O>
ASLR_TG_FN typename O::real team_solve(const TeamFactor<O> &F, typename O::real b, const typename O::real (&oh)[4]) {
  using real = typename O::real;
  real y[4], z[4];
  real s = b;
#define ASLR_TG_FWD(k)                                  \
  {                                                     \
    const real t = s * F.rinv[k];                       \
    y[k] = O::template bc<k>(t);                        \
    if (k < 3) s = s - F.Lr[k] * y[k];                  \
  }
  ASLR_TG_FWD(0) ASLR_TG_FWD(1) ASLR_TG_FWD(2) ASLR_TG_FWD(3)
#undef ASLR_TG_FWD
  z[3] = y[3] * F.rinv[3]; // (y is row-uniform, and row 3 has nothing to subtract)
  {
    real s2 = y[2];
    s2 = s2 - F.Lc[3] * z[3];
    z[2] = O::template bc<2>(s2 * F.rinv[2]);
  }
  {
    real s1 = y[1];
    s1 = s1 - F.Lc[2] * z[2];
    s1 = s1 - F.Lc[3] * z[3];
    z[1] = O::template bc<1>(s1 * F.rinv[1]);
  }
  {
    real s0 = y[0];
    s0 = s0 - F.Lc[1] * z[1];
    s0 = s0 - F.Lc[2] * z[2];
    s0 = s0 - F.Lc[3] * z[3];
    z[0] = O::template bc<0>(s0 * F.rinv[0]);
  }
  return ((oh[0] * z[0] + oh[1] * z[1]) + oh[2] * z[2]) + oh[3] * z[3]; // (exact: one term is non-zero)
}

// sum of the four elements of a distributed vector, in element order, the same in every lane
template <class O>
ASLR_TG_FN typename O::real team_sum(typename O::real t, typename O::real one) {
  return O::sum4(t, one);
}

// Gains of one knot.  In (per lane, r = lane & 3): Hr = row r of Quu (regularised), q = Qu_r, `boxed` (the same in
// the whole row): the node takes SolverBoxDDP's QP -- then lb = u_lb_r - u_r, ub = u_ub_r - u_r, k0 = the stored k_r
// (warm start); oh = one-hot of r.  Out: kv = k_r, qz = Qu_r with clamped entries zeroed, F = the factor of the
// final free block (the caller solves its columns of Qux with it: K = Quu_inv Qux), bad = "backward_error".
//
// One flow for every node: the first Newton step on the first active set IS the plain DDP gain when nothing is
// boxed (x0 = 0, every index free: z = -Quu^-1 Qu), and BoxQP's first iteration otherwise.  BoxQP outcomes that need
// no line search leave at once: (a) |g_free(x0)|_inf <= th_grad or nothing free: x0 itself; (b) nothing clamped and
// the Newton point strictly inside the box: it is accepted with alpha = 1 and passes the next gradient test with
// every index free.  Everything else iterates (BoxQP::solve), all rows of the wave together.
template <class O, bool BOX>
ASLR_TG_FN void team_gains4(const typename O::real (&Hr)[4], typename O::real q, typename O::mask boxed,
                            typename O::real lb, typename O::real ub, typename O::real k0,
                            const typename O::real (&oh)[4], const TeamQPParams &P, typename O::real &kv,
                            typename O::real &qz, TeamFactor<O> &F, typename O::mask &bad) {
  using real = typename O::real;
  using mask = typename O::mask;
  const real zero = O::cst(0.0), one = O::cst(1.0);
  real x = zero, g = q;
  mask cl = O::mfalse();
  if (BOX) {
    x = O::sel(boxed, O::fmax(O::fmin(k0, ub), lb), zero);
    O::template matvec_acc<false>(g, x, Hr); // g = q + H x
    cl = boxed & (((x == lb) & (g > zero)) | ((x == ub) & (g < zero)));
  }
  real mk = O::sel(cl, zero, one);
  const real reg = BOX ? O::sel(boxed, O::cst(P.reg), zero) : zero;
  mask allcl = O::mfalse();
  if (BOX) allcl = O::team_all(cl);
  ASLR_TG_MARK(0);
  mask cbad = team_factor<O>(F, Hr, mk, reg, oh);
  ASLR_TG_MARK(1);
  bad = cbad & (!allcl); // (a factor of a non-empty free block is needed in every outcome)
  real z;
  {
    const real xc = (one - mk) * x;
    real s = -q;
    O::template matvec_acc<true>(s, xc, Hr); // -q - H ((1 - mk) x)
    z = team_solve<O>(F, mk * s, oh);
  }
  mask finished = O::mtrue();
  if (BOX) {
    const mask fin0 = (!O::team_any((mk * O::fabs(g)) > O::cst(P.th_grad))) | allcl;
    const real mrg = O::cst(1e-9) * (one + O::fabs(z));
    const mask inside = (z > lb + mrg) & (z < ub - mrg);
    const mask interior = (!O::team_any(cl)) & (!cbad) & O::team_all(inside) & O::uniform(P.reg == 0.0);
    x = O::sel((!boxed) | (interior & (!fin0)), z, x);
    finished = (!boxed) | fin0 | interior;
  } else {
    x = z;
  }
  ASLR_TG_MARK(2);
  if (BOX && O::wave_any(!finished)) {
    ASLR_TG_COUNT(10);
    // f(x) = 1/2 x^T H x + q^T x
    real fold;
    {
      real s = zero;
      O::template matvec_acc<false>(s, x, Hr);
      fold = team_sum<O>(O::cst(0.5) * x * s + q * x, one);
    }
    for (int it = 0;;) {
      ASLR_TG_COUNT(11);
      // ---- line search along the projected Newton direction ----
      const real dx = mk * (z - x);
      mask found = finished | cbad;
      real alpha = one, fnext = fold;
      for (int al = 0; al < P.nalpha; ++al) {
        ASLR_TG_COUNT(13);
        const real xn = O::fmax(O::fmin(x + alpha * dx, ub), lb);
        real s = zero;
        O::template matvec_acc<false>(s, xn, Hr);
        const real fnew = team_sum<O>(O::cst(0.5) * xn * s + q * xn, one);
        const real gd = team_sum<O>(g * (x - xn), one);
        const mask take = (!found) & ((fold - fnew) > O::cst(P.th_acceptstep) * gd);
        x = O::sel(take, xn, x);
        fnext = O::sel(take, fnew, fnext);
        found = found | take;
        if (!O::wave_any(!found)) break;
        alpha = alpha * O::cst(0.5);
      }
      fold = fnext;
      // no step length accepted: x is unchanged, every further iteration would repeat this one
      finished = finished | cbad | (!found);
      g = q;
      O::template matvec_acc<false>(g, x, Hr);
      ASLR_TG_MARK(3);
      if (++it >= P.maxiter) break;
      // ---- next iteration: active set, convergence, factor, Newton point ----
      cl = ((x == lb) & (g > zero)) | ((x == ub) & (g < zero));
      const real mj = O::sel(cl, zero, one);
      const mask fin_now = finished | (!O::team_any((mj * O::fabs(g)) > O::cst(P.th_grad))) | O::team_all(cl);
      mk = O::sel(finished, mk, mj);
      if (!O::wave_any(!fin_now)) { finished = O::mtrue(); break; }
      cbad = team_factor<O>(F, Hr, mk, reg, oh);
      bad = bad | (cbad & (!fin_now));
      finished = fin_now;
      {
        const real xc = (one - mk) * x;
        real s = -q;
        O::template matvec_acc<true>(s, xc, Hr);
        z = team_solve<O>(F, mk * s, oh);
      }
      ASLR_TG_MARK(4);
    }
    ASLR_TG_MARK(4);
    // factor of the final free block: the one at hand unless the active set changed in the last step
    const mask stale = O::team_any(!(F.mkL == mk));
    if (O::wave_any(stale)) {
      ASLR_TG_COUNT(12);
      const mask cb = team_factor<O>(F, Hr, mk, reg, oh);
      bad = bad | (cb & O::team_any(mk > zero));
    }
  }
  ASLR_TG_MARK(5);
  kv = -x;
  qz = q * mk; // (mk = 0 on clamped entries of boxed nodes, 1 otherwise; exact)
}

// Column `col` of Quu_inv Qux with the final factor: Quu_inv = Hff^-1 on the free block, zero elsewhere.
// F's row-distributed factor is first made row-uniform (every lane then solves ITS column of Qux).
template <class O>
ASLR_TG_FN void team_gain_column(const TeamFactor<O> &F, typename O::real (&col)[4]) {
  using real = typename O::real;
  real L10, L20, L21, L30, L31, L32; // the strictly lower part of L, row-uniform
  O::bc_lower(F.Lr, L10, L20, L21, L30, L31, L32);
  real b0 = col[0] * F.mkk[0], b1 = col[1] * F.mkk[1], b2 = col[2] * F.mkk[2], b3 = col[3] * F.mkk[3];
  // chol_solve_r, unrolled
  b0 = b0 * F.rinv[0];
  b1 = (b1 - L10 * b0) * F.rinv[1];
  b2 = ((b2 - L20 * b0) - L21 * b1) * F.rinv[2];
  b3 = (((b3 - L30 * b0) - L31 * b1) - L32 * b2) * F.rinv[3];
  b3 = b3 * F.rinv[3];
  b2 = (b2 - L32 * b3) * F.rinv[2];
  b1 = ((b1 - L21 * b2) - L31 * b3) * F.rinv[1];
  b0 = (((b0 - L10 * b1) - L20 * b2) - L30 * b3) * F.rinv[0];
  col[0] = b0 * F.mkk[0]; col[1] = b1 * F.mkk[1]; col[2] = b2 * F.mkk[2]; col[3] = b3 * F.mkk[3];
}

} // namespace aslr
