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
  static ASLR_DEV real rsqrt(real a) { return ::rsqrt(a); }
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
