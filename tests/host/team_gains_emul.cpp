// Host emulation of the lane choreography of aslr_to_amd/csrc/aslr_team_gains.hpp: the SAME template source the
// backward kernel instantiates with DPP broadcasts runs here on 64 emulated lanes (4 rows of 16, i.e. the four
// trajectories of a wavefront in lock-step), so that the distribution of the 4x4 gains / box-QP over the lanes can be
// checked against the oracle's BoxQP (oracle/aslr_oracle.c:865-957) without a GPU.  Test infrastructure only
// (tests/test_team_gains_emul.py builds it with g++).
#include <cmath>
#include <cstring>

namespace emu {
constexpr int W = 64;
struct M { bool m[W]; };
struct V { double v[W]; };
#define EMU_BIN(op)                                                                                         \
  inline V operator op(const V &a, const V &b) { V r; for (int i = 0; i < W; ++i) r.v[i] = a.v[i] op b.v[i]; return r; } \
  inline V operator op(double a, const V &b) { V r; for (int i = 0; i < W; ++i) r.v[i] = a op b.v[i]; return r; }        \
  inline V operator op(const V &a, double b) { V r; for (int i = 0; i < W; ++i) r.v[i] = a.v[i] op b; return r; }
EMU_BIN(+) EMU_BIN(-) EMU_BIN(*)
#undef EMU_BIN
inline V operator-(const V &a) { V r; for (int i = 0; i < W; ++i) r.v[i] = -a.v[i]; return r; }
#define EMU_CMP(op)                                                                                         \
  inline M operator op(const V &a, const V &b) { M r; for (int i = 0; i < W; ++i) r.m[i] = a.v[i] op b.v[i]; return r; }
EMU_CMP(==) EMU_CMP(>) EMU_CMP(<)
#undef EMU_CMP
inline M operator&(const M &a, const M &b) { M r; for (int i = 0; i < W; ++i) r.m[i] = a.m[i] && b.m[i]; return r; }
inline M operator|(const M &a, const M &b) { M r; for (int i = 0; i < W; ++i) r.m[i] = a.m[i] || b.m[i]; return r; }
inline M operator!(const M &a) { M r; for (int i = 0; i < W; ++i) r.m[i] = !a.m[i]; return r; }

struct Ops {
  using real = V;
  using mask = M;
  static V cst(double c) { V r; for (int i = 0; i < W; ++i) r.v[i] = c; return r; }
  static M mfalse() { M r; for (int i = 0; i < W; ++i) r.m[i] = false; return r; }
  static M mtrue() { M r; for (int i = 0; i < W; ++i) r.m[i] = true; return r; }
  static M uniform(bool b) { return b ? mtrue() : mfalse(); }
  static V sel(const M &m, const V &a, const V &b) { V r; for (int i = 0; i < W; ++i) r.v[i] = m.m[i] ? a.v[i] : b.v[i]; return r; }
  static V fmin(const V &a, const V &b) { V r; for (int i = 0; i < W; ++i) r.v[i] = std::fmin(a.v[i], b.v[i]); return r; }
  static V fmax(const V &a, const V &b) { V r; for (int i = 0; i < W; ++i) r.v[i] = std::fmax(a.v[i], b.v[i]); return r; }
  static V fabs(const V &a) { V r; for (int i = 0; i < W; ++i) r.v[i] = std::fabs(a.v[i]); return r; }
  static V rsqrt(const V &a) { V r; for (int i = 0; i < W; ++i) r.v[i] = 1.0 / std::sqrt(a.v[i]); return r; }
  // row_newbcast:C -- every lane of a 16-lane row reads lane C of that row
  template <int C> static V bc(const V &a) { V r; for (int i = 0; i < W; ++i) r.v[i] = a.v[(i & ~15) + C]; return r; }
  template <int C, bool NEG> static void fmac_bc(V &acc, const V &v, const V &h) {
    const V b = bc<C>(v);
    for (int i = 0; i < W; ++i) acc.v[i] = NEG ? acc.v[i] - b.v[i] * h.v[i] : acc.v[i] + b.v[i] * h.v[i];
  }
  template <bool NEG> static void matvec_acc(V &acc, const V &v, const V (&h)[4]) {
    fmac_bc<0, NEG>(acc, v, h[0]); fmac_bc<1, NEG>(acc, v, h[1]); fmac_bc<2, NEG>(acc, v, h[2]); fmac_bc<3, NEG>(acc, v, h[3]);
  }
  // the multi-instruction blocks of the device version (one wait state in front of several DPP instructions)
  static void bc4(const V &v, V (&o)[4]) { o[0] = bc<0>(v); o[1] = bc<1>(v); o[2] = bc<2>(v); o[3] = bc<3>(v); }
  static V sum4(const V &t, const V &one) {
    V s = bc<0>(t);
    fmac_bc<1, false>(s, t, one); fmac_bc<2, false>(s, t, one); fmac_bc<3, false>(s, t, one);
    return s;
  }
  // Lc[k] = L[k][r] for k > r: sum_c oh[c] * (L[k][c] from lane k), exact
  static void transpose_lower(const V (&Lr)[4], const V (&oh)[4], V (&Lc)[4]) {
    for (int k = 0; k < 4; ++k) Lc[k] = cst(0.0);
    fmac_bc<1, false>(Lc[1], Lr[0], oh[0]);
    fmac_bc<2, false>(Lc[2], Lr[0], oh[0]); fmac_bc<2, false>(Lc[2], Lr[1], oh[1]);
    fmac_bc<3, false>(Lc[3], Lr[0], oh[0]); fmac_bc<3, false>(Lc[3], Lr[1], oh[1]); fmac_bc<3, false>(Lc[3], Lr[2], oh[2]);
  }
  static void bc_lower(const V (&Lr)[4], V &L10, V &L20, V &L21, V &L30, V &L31, V &L32) {
    L10 = bc<1>(Lr[0]); L20 = bc<2>(Lr[0]); L21 = bc<2>(Lr[1]); L30 = bc<3>(Lr[0]); L31 = bc<3>(Lr[1]); L32 = bc<3>(Lr[2]);
  }
  // the device version looks at lanes 0..3 of the row only (the other quads hold copies)
  static M team_any(const M &p) {
    M r;
    for (int i = 0; i < W; ++i) { const int b = i & ~15; r.m[i] = p.m[b] || p.m[b + 1] || p.m[b + 2] || p.m[b + 3]; }
    return r;
  }
  static M team_all(const M &p) { return !team_any(!p); }
  static bool wave_any(const M &p) { for (int i = 0; i < W; ++i) if (p.m[i]) return true; return false; }
};
} // namespace emu

#include "../../aslr_to_amd/csrc/aslr_team_gains.hpp"

// Four problems (one per 16-lane row).  H [4][16], q / lb / ub / k0 [4][4], boxed [4], Qux [4][4][8] (control row, state
// column).  Out: k [4][4], qz [4][4], K [4][4][8], bad [4].  box = 0 runs the SolverDDP instantiation.
extern "C" void emul_team_gains(int box, const double *H, const double *q, const double *lb, const double *ub,
                                const double *k0, const int *boxed, const double *Qux, int maxiter, double th_acceptstep,
                                double th_grad, double reg, double *k, double *qz, double *K, int *bad) {
  using namespace emu;
  V Hr[4], qv, lbv, ubv, k0v, oh[4], kv, qzv;
  M bx, bd;
  for (int i = 0; i < W; ++i) {
    const int row = i / 16, r = i & 3;
    for (int c = 0; c < 4; ++c) { Hr[c].v[i] = H[row * 16 + r * 4 + c]; oh[c].v[i] = (c == r) ? 1.0 : 0.0; }
    qv.v[i] = q[row * 4 + r]; lbv.v[i] = lb[row * 4 + r]; ubv.v[i] = ub[row * 4 + r]; k0v.v[i] = k0[row * 4 + r];
    bx.m[i] = boxed[row] != 0;
  }
  aslr::TeamQPParams P{maxiter, th_acceptstep, th_grad, reg, 10};
  aslr::TeamFactor<Ops> F;
  if (box) aslr::team_gains4<Ops, true>(Hr, qv, bx, lbv, ubv, k0v, oh, P, kv, qzv, F, bd);
  else aslr::team_gains4<Ops, false>(Hr, qv, bx, lbv, ubv, k0v, oh, P, kv, qzv, F, bd);
  V col[4];
  for (int i = 0; i < W; ++i) {
    const int row = i / 16, jj = i & 7;
    for (int c = 0; c < 4; ++c) col[c].v[i] = Qux[(row * 4 + c) * 8 + jj];
  }
  aslr::team_gain_column<Ops>(F, col);
  for (int row = 0; row < 4; ++row) {
    for (int r = 0; r < 4; ++r) { k[row * 4 + r] = kv.v[row * 16 + r]; qz[row * 4 + r] = qzv.v[row * 16 + r]; }
    for (int c = 0; c < 4; ++c)
      for (int jj = 0; jj < 8; ++jj) K[(row * 4 + c) * 8 + jj] = col[c].v[row * 16 + jj];
    bad[row] = bd.m[row * 16];
    // every replica (quad) of a row must agree
    for (int i = 1; i < 16; ++i)
      if (kv.v[row * 16 + i] != kv.v[row * 16 + (i & 3)] && !(std::isnan(kv.v[row * 16 + i]))) bad[row] |= 2;
  }
}
