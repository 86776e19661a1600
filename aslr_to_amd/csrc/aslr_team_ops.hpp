// aslr_team_ops.hpp -- the GPU instantiation of the `Ops` policy of aslr_team_gains.hpp: real = double, one lane of
// a 16-lane DPP row; broadcasts are gfx90a+ "DP ALU DPP" operations with `row_newbcast:c` (lane c of the row feeds
// every lane of the row), available for the 64-bit VOP1 / VOP2 encodings: v_mov_b64 and v_fmac_f64 are the two used.
//
// They are issued through inline assembly (this compiler has no 64-bit update_dpp builtin and does not fold a 64-bit
// DPP move into the FMA).  The hazard the compiler would otherwise pad -- a VALU write of a VGPR followed within two
// instructions by a DPP read of it (2 wait states; a trans-unit result one more) -- is covered by the `s_nop 1` that
// opens every block; inside a block only the accumulator is rewritten, which the FMA reads as an ordinary operand.
// EXEC is full wherever these run (the backward kernel keeps all 64 lanes alive and carries inactivity as flags),
// so every broadcast source lane is valid.
#pragma once
#include "aslr_device.hpp"
#define ASLR_TG_FN __device__ __forceinline__
#ifdef ASLR_BWD_PROFILE
// region timing inside team_gains4 (profile build only): lane 0's shader clock between marks, kept in LDS;
// [0..5] regions, [10..12] counters, [15] the last clock value.  backward_kernel zeroes and flushes the table.
namespace aslr { __device__ __forceinline__ long long *tg_prof() { static __shared__ long long a[16]; return a; } }
#define ASLR_TG_MARK(i) do { if (threadIdx.x == 0) { long long *p_ = aslr::tg_prof(); const long long n_ = clock64(); p_[i] += n_ - p_[15]; p_[15] = n_; } } while (0)
#define ASLR_TG_COUNT(i) do { if (threadIdx.x == 0) aslr::tg_prof()[i] += 1; } while (0)
#endif
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wbitwise-instead-of-logical" // (masks are plain bools here; & and | on purpose: no branches)
#include "aslr_team_gains.hpp"
#pragma clang diagnostic pop

namespace aslr {

#ifndef ASLR_DPP_NOP
#define ASLR_DPP_NOP "s_nop 1\n\t"
#endif
#ifndef ASLR_DPP_NOP_CHAIN
#define ASLR_DPP_NOP_CHAIN "" // between the dependent FMAs of one block (tools/ubench/dpp_hazard.hip decides)
#endif

struct DevTeamOps {
  using real = double;
  using mask = bool;
  static ASLR_DEV real cst(double c) { return c; }
  static ASLR_DEV mask mfalse() { return false; }
  static ASLR_DEV mask mtrue() { return true; }
  static ASLR_DEV mask uniform(bool b) { return b; }
  static ASLR_DEV real sel(mask m, real a, real b) { return m ? a : b; }
  static ASLR_DEV real fmin(real a, real b) { return ::fmin(a, b); }
  static ASLR_DEV real fmax(real a, real b) { return ::fmax(a, b); }
  static ASLR_DEV real fabs(real a) { return ::fabs(a); }
  // 1 / sqrt(a) for the Cholesky pivots: v_rsq_f64 and the library's own refinement step (the same six operations, so
  // the same bits for every positive finite argument) WITHOUT its class test for 0 / inf / NaN inputs (3 instructions
  // per pivot): a pivot that is not positive is flagged from `d > 0` by the caller and the sweep is redone anyway.
  static ASLR_DEV real rsqrt(real a) {
    const real y = __builtin_amdgcn_rsq(a);
    const real e = fma(y * -a, y, 1.0);
    return fma(y * e, fma(e, 0.375, 0.5), y);
  }
  template <int C>
  static ASLR_DEV real bc(real v) {
    real r;
    asm(ASLR_DPP_NOP "v_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v), "n"(C));
    return r;
  }
  // acc (+/-)= (v of lane C) * h
  template <int C, bool NEG>
  static ASLR_DEV void fmac_bc(real &acc, real v, real h) {
    if (NEG) asm(ASLR_DPP_NOP "v_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(v), "v"(h), "n"(C));
    else asm(ASLR_DPP_NOP "v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(v), "v"(h), "n"(C));
  }
  // acc (+/-)= sum_c (v of lane c) * h[c], c ascending: one row of a 4x4 matrix-vector product
  template <bool NEG>
  static ASLR_DEV void matvec_acc(real &acc, real v, const real (&h)[4]) {
    if (NEG)
      asm(ASLR_DPP_NOP "v_fmac_f64_dpp %0, -%1, %2 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t" ASLR_DPP_NOP_CHAIN
          "v_fmac_f64_dpp %0, -%1, %3 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t" ASLR_DPP_NOP_CHAIN
          "v_fmac_f64_dpp %0, -%1, %4 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t" ASLR_DPP_NOP_CHAIN
          "v_fmac_f64_dpp %0, -%1, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf"
          : "+v"(acc) : "v"(v), "v"(h[0]), "v"(h[1]), "v"(h[2]), "v"(h[3]));
    else
      asm(ASLR_DPP_NOP "v_fmac_f64_dpp %0, %1, %2 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t" ASLR_DPP_NOP_CHAIN
          "v_fmac_f64_dpp %0, %1, %3 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t" ASLR_DPP_NOP_CHAIN
          "v_fmac_f64_dpp %0, %1, %4 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t" ASLR_DPP_NOP_CHAIN
          "v_fmac_f64_dpp %0, %1, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf"
          : "+v"(acc) : "v"(v), "v"(h[0]), "v"(h[1]), "v"(h[2]), "v"(h[3]));
  }
#define ASLR_NB(c) " row_newbcast:" #c " row_mask:0xf bank_mask:0xf\n\t"
  // multi-instruction blocks: the DPP sources are not rewritten inside a block, so one wait state in front suffices
  static ASLR_DEV void bc4(real v, real (&o)[4]) {
    asm(ASLR_DPP_NOP "v_mov_b64_dpp %0, %4" ASLR_NB(0) "v_mov_b64_dpp %1, %4" ASLR_NB(1) "v_mov_b64_dpp %2, %4" ASLR_NB(2) "v_mov_b64_dpp %3, %4" ASLR_NB(3)
        : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]) : "v"(v));
  }
  static ASLR_DEV real sum4(real t, real one) {
    real s;
    asm(ASLR_DPP_NOP "v_mov_b64_dpp %0, %1" ASLR_NB(0) "v_fmac_f64_dpp %0, %1, %2" ASLR_NB(1) "v_fmac_f64_dpp %0, %1, %2" ASLR_NB(2) "v_fmac_f64_dpp %0, %1, %2" ASLR_NB(3)
        : "=&v"(s) : "v"(t), "v"(one));
    return s;
  }
  static ASLR_DEV void transpose_lower(const real (&Lr)[4], const real (&oh)[4], real (&Lc)[4]) {
    real c1 = 0.0, c2 = 0.0, c3 = 0.0;
    asm(ASLR_DPP_NOP "v_fmac_f64_dpp %0, %3, %6" ASLR_NB(1) "v_fmac_f64_dpp %1, %3, %6" ASLR_NB(2) "v_fmac_f64_dpp %1, %4, %7" ASLR_NB(2)
        "v_fmac_f64_dpp %2, %3, %6" ASLR_NB(3) "v_fmac_f64_dpp %2, %4, %7" ASLR_NB(3) "v_fmac_f64_dpp %2, %5, %8" ASLR_NB(3)
        : "+v"(c1), "+v"(c2), "+v"(c3) : "v"(Lr[0]), "v"(Lr[1]), "v"(Lr[2]), "v"(oh[0]), "v"(oh[1]), "v"(oh[2]));
    Lc[0] = 0.0; Lc[1] = c1; Lc[2] = c2; Lc[3] = c3;
  }
  static ASLR_DEV void bc_lower(const real (&Lr)[4], real &L10, real &L20, real &L21, real &L30, real &L31, real &L32) {
    asm(ASLR_DPP_NOP "v_mov_b64_dpp %0, %6" ASLR_NB(1) "v_mov_b64_dpp %1, %6" ASLR_NB(2) "v_mov_b64_dpp %2, %7" ASLR_NB(2)
        "v_mov_b64_dpp %3, %6" ASLR_NB(3) "v_mov_b64_dpp %4, %7" ASLR_NB(3) "v_mov_b64_dpp %5, %8" ASLR_NB(3)
        : "=&v"(L10), "=&v"(L20), "=&v"(L21), "=&v"(L30), "=&v"(L31), "=&v"(L32) : "v"(Lr[0]), "v"(Lr[1]), "v"(Lr[2]));
  }
  // acc += sum_l (v of lane l) * h[l], l = 0..7 ascending: a dot product with a vector spread over lanes 0..7 of the row
  static ASLR_DEV void dot8_acc(real &acc, real v, const real (&h)[8]) {
    asm(ASLR_DPP_NOP "v_fmac_f64_dpp %0, %1, %2" ASLR_NB(0) "v_fmac_f64_dpp %0, %1, %3" ASLR_NB(1) "v_fmac_f64_dpp %0, %1, %4" ASLR_NB(2)
        "v_fmac_f64_dpp %0, %1, %5" ASLR_NB(3) "v_fmac_f64_dpp %0, %1, %6" ASLR_NB(4) "v_fmac_f64_dpp %0, %1, %7" ASLR_NB(5)
        "v_fmac_f64_dpp %0, %1, %8" ASLR_NB(6) "v_fmac_f64_dpp %0, %1, %9" ASLR_NB(7)
        : "+v"(acc) : "v"(v), "v"(h[0]), "v"(h[1]), "v"(h[2]), "v"(h[3]), "v"(h[4]), "v"(h[5]), "v"(h[6]), "v"(h[7]));
  }
  // three running sums over the row at once: a += sum_c ta_c, b -= sum_c tb_c, c += sum_c tc_c (c ascending)
  static ASLR_DEV void acc3(real &a, real ta, real &b, real tb, real &c, real tc, real one) {
    asm(ASLR_DPP_NOP "v_fmac_f64_dpp %0, %3, %6" ASLR_NB(0) "v_fmac_f64_dpp %1, -%4, %6" ASLR_NB(0) "v_fmac_f64_dpp %2, %5, %6" ASLR_NB(0)
        "v_fmac_f64_dpp %0, %3, %6" ASLR_NB(1) "v_fmac_f64_dpp %1, -%4, %6" ASLR_NB(1) "v_fmac_f64_dpp %2, %5, %6" ASLR_NB(1)
        "v_fmac_f64_dpp %0, %3, %6" ASLR_NB(2) "v_fmac_f64_dpp %1, -%4, %6" ASLR_NB(2) "v_fmac_f64_dpp %2, %5, %6" ASLR_NB(2)
        "v_fmac_f64_dpp %0, %3, %6" ASLR_NB(3) "v_fmac_f64_dpp %1, -%4, %6" ASLR_NB(3) "v_fmac_f64_dpp %2, %5, %6" ASLR_NB(3)
        : "+v"(a), "+v"(b), "+v"(c) : "v"(ta), "v"(tb), "v"(tc), "v"(one));
  }
  // "some lane of my 16-lane row has p": the four quads of a row hold copies, so lanes 0..3 of the row decide
  static ASLR_DEV mask team_any(mask p) {
    const unsigned long long m = __ballot(p);
    const unsigned lane = threadIdx.x;
    const unsigned w = (lane & 32u) ? (unsigned)(m >> 32) : (unsigned)m;
    return ((w >> (lane & 16u)) & 0xFu) != 0u;
  }
  static ASLR_DEV mask team_all(mask p) { return !team_any(!p); }
  static ASLR_DEV bool wave_any(mask p) { return __ballot(p) != 0ull; }
};

} // namespace aslr
