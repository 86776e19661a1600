// aslr_device.hpp -- per-knot model arithmetic for gfx950, one LANE per (trajectory, knot).
//
// Everything here is thread-local and fully unrolled over the compile-time joint count NJ, so the
// small vectors/matrices live in VGPRs.  It implements, for a fixed-base serial revolute chain:
//   * SEA / VSA free-forward-dynamics calc   (python/aslr_to/free_fwddyn_asr.py:20-56,
//                                            python/aslr_to/free_fwddyn_vsa.py:20-57)
//   * their calcDiff                         (free_fwddyn_asr.py:58-92, free_fwddyn_vsa.py:59-94)
//   * the semi-implicit Euler step           (python/aslr_to/integrated_action.py:13-42)
//   * the cost stack of the example scripts  
// The rigid-body pieces Pinocchio provides to the reference (computeAllTerms, computeRNEADerivatives,
// frame placement / LOCAL frame Jacobian, log6 / Jlog6) are written here directly: RNEA for nle,
// unit-acceleration RNEA passes for M, forward-mode (tangent) RNEA for dtau_dq / dtau_dv.
#pragma once

#include <hip/hip_runtime.h>

#include <type_traits>

#include "../../include/aslr_to_amd.h"

namespace aslr {

#define ASLR_DEV __device__ __forceinline__
#define ASLR_UNROLL _Pragma("unroll")

// Ordering point for LDS traffic inside ONE wavefront (every block of these kernels is a single
// wave): the hardware executes a wave's LDS instructions in order, so all that is needed is that the
// compiler does not move LDS accesses across this point.  Unlike __syncthreads() it does not drain
// vmcnt, so global prefetches and streaming stores stay in flight across it.
ASLR_DEV void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Region timing of the serial sweeps (build with -DASLR_BWD_PROFILE; tools/bwd_regions*.py, tools/fwd_regions_c5.py): every wave adds the shader-clock
// cycles it spent between consecutive marks to a device-side table.  Compiled out of the product library.
#ifdef ASLR_BWD_PROFILE
static __device__ unsigned long long aslr_bwd_prof_dev[32];
#define ASLR_PROF_DECL long long prof_acc[20] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, prof_last = clock64()
#define ASLR_PROF(i) do { const long long now_ = clock64(); prof_acc[i] += now_ - prof_last; prof_last = now_; } while (0)
#define ASLR_PROF_COUNT(i) do { prof_acc[i] += 1; } while (0)
#define ASLR_PROF_FLUSH do { if (threadIdx.x == 0) { for (int i_ = 0; i_ < 20; ++i_) atomicAdd(&aslr_bwd_prof_dev[i_], (unsigned long long)prof_acc[i_]); } } while (0)
#else
#define ASLR_PROF_DECL
#define ASLR_PROF(i)
#define ASLR_PROF_COUNT(i)
#define ASLR_PROF_FLUSH
#endif

typedef __attribute__((address_space(3))) void *lds_void_p;

// One 16-byte piece per lane from global memory straight into LDS (global_load_lds_dwordx4): lane L's piece lands at
// LDS address lds_addr + OFF + 16 L, and OFF also advances the global address.  Issued through inline assembly on
// purpose: for the builtin the compiler drains vmcnt before EVERY later LDS read whose memory operand has lost its
// alias scope (all merged ds_read_b128 have), i.e. right after the issue, which exposes the whole HBM latency.  The
// kernels order these loads by hand instead: an explicit s_waitcnt vmcnt(n) before the first read of the target (n =
// the number of vector-memory instructions issued after the loads that may still be in flight), and an lgkmcnt(0) +
// wave barrier before the issue so that no earlier read of the target is still pending.  (m0 has no other user in
// these kernels.)  NT: non-temporal (data read once).
template <int OFF, bool NT = true>
ASLR_DEV void dma16(const char *g, unsigned lds_addr) {
  if (NT) asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off offset:%2 nt" : : "v"(g), "s"(lds_addr), "n"(OFF) : "memory");
  else asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off offset:%2" : : "v"(g), "s"(lds_addr), "n"(OFF) : "memory");
}
ASLR_DEV unsigned lds_address(const void *p) { return (unsigned)(size_t)(lds_void_p)p; }
// s_waitcnt with only one counter constrained (gfx9 encoding: vmcnt in bits [3:0] and [15:14], expcnt [6:4],
// lgkmcnt [11:8]; the unconstrained fields hold their maxima)
template <int N>
ASLR_DEV void wait_vmcnt() {
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
  __builtin_amdgcn_s_waitcnt(0x0F70 | (N & 15) | ((N >> 4) << 14));
}
ASLR_DEV void wait_lgkmcnt0() { __builtin_amdgcn_s_waitcnt(0xC07F); }

// sin/cos for joint angles: Cody-Waite reduction by pi/2 (33 + 53 bits of pi/2, exact for
// |x| < ~1e5) and the fdlibm kernel polynomials; < 1 ulp there.  Larger arguments take the library
// path.  (ocml's sincos carries the full Payne-Hanek reduction inline: ~6x the instructions.)
ASLR_DEV void sincos_fast(double x, double *sn, double *cs) {
  if (!(fabs(x) < 1.0e5)) { sincos(x, sn, cs); return; }
  const double fn = rint(x * 6.36619772367581382433e-01);
  const double r = fma(-fn, 1.57079632673412561417e+00, x);
  const double y = r - fn * 6.07710050650619224932e-11;
  const double z = y * y;
  // kernel sin
  const double rs = 8.33333333332248946124e-03 + z * (-1.98412698298579493134e-04 + z * (2.75573137070700676789e-06 +
                    z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10)));
  const double ks = y + (z * y) * (-1.66666666666666324348e-01 + z * rs);
  // kernel cos
  const double rc = z * (4.16666666666666019037e-02 + z * (-1.38888888888741095749e-03 + z * (2.48015872894767294178e-05 +
                    z * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11)))));
  const double hz = 0.5 * z, w = 1.0 - hz;
  const double kc = w + (((1.0 - w) - hz) + z * rc);
  const int n = (int)fn & 3;
  const double s0 = (n & 1) ? kc : ks, c0 = (n & 1) ? ks : kc;
  *sn = (n & 2) ? -s0 : s0;
  *cs = ((n + 1) & 2) ? -c0 : c0;
}

// sincos_fast with its 16 constants passed in (VGPR-resident copies made once per kernel by SinCosK): in a loop over
// knots the compiler otherwise re-materialises every 64-bit literal with two scalar moves per use and iteration,
// 64 SALU instructions per knot of the planar rollout -- which runs at the one-instruction-per-8-cycles issue limit.
struct SinCosK {
  double k[16];
  // opaque = true: VGPR-resident copies (kernels that loop over knots); false: plain literals, folded as before
  ASLR_DEV explicit SinCosK(bool opaque) {
    const double v[16] = {6.36619772367581382433e-01, 1.57079632673412561417e+00, 6.07710050650619224932e-11,
                          8.33333333332248946124e-03, -1.98412698298579493134e-04, 2.75573137070700676789e-06,
                          -2.50507602534068634195e-08, 1.58969099521155010221e-10, -1.66666666666666324348e-01,
                          4.16666666666666019037e-02, -1.38888888888741095749e-03, 2.48015872894767294178e-05,
                          -2.75573143513906633035e-07, 2.08757232129817482790e-09, -1.13596475577881948265e-11, 0.5};
    ASLR_UNROLL for (int i = 0; i < 16; ++i) {
      double c = v[i];
      if (opaque) asm volatile("" : "+v"(c)); // an opaque register value from here on
      k[i] = c;
    }
  }
};
ASLR_DEV void sincos_fast(const SinCosK &K, double x, double *sn, double *cs) {
  if (!(fabs(x) < 1.0e5)) { sincos(x, sn, cs); return; }
  const double *k = K.k;
  const double fn = rint(x * k[0]);
  const double r = fma(-fn, k[1], x);
  const double y = r - fn * k[2];
  const double z = y * y;
  const double rs = k[3] + z * (k[4] + z * (k[5] + z * (k[6] + z * k[7])));
  const double ks = y + (z * y) * (k[8] + z * rs);
  const double rc = z * (k[9] + z * (k[10] + z * (k[11] + z * (k[12] + z * (k[13] + z * k[14])))));
  const double hz = k[15] * z, w = 1.0 - hz;
  const double kc = w + (((1.0 - w) - hz) + z * rc);
  const int n = (int)fn & 3;
  const double s0 = (n & 1) ? kc : ks, c0 = (n & 1) ? ks : kc;
  *sn = (n & 2) ? -s0 : s0;
  *cs = ((n + 1) & 2) ? -c0 : c0;
}

// Device-side model: the ABI struct plus host-precomputed inverse of the motor inertia.
struct DevModel {
  aslr_model_t m;
  double Binv[ASLR_MAX_NJ * ASLR_MAX_NJ];
  // frame-placement costs whose frame is turned about z on its joint (planar fast path, PlanarChain::reach_ok):
  // cos, sin and angle of that rotation, per cost term; filled on the host
  double fr_c[ASLR_MAX_COSTS], fr_s[ASLR_MAX_COSTS], fr_phi[ASLR_MAX_COSTS];
};
// Planar restatement of the chain (valid when `ok`): every joint axis is +z and every joint
// placement rotates about z.  Filled on the host at problem creation.
struct PlanarChain {
  int32_t ok;
  // the frame-placement residuals take the closed form of ChainPlanar::reach_residual: every cost frame is turned
  // about z on its joint and every reference rotation (cost defaults and per-trajectory overrides) is the identity
  int32_t reach_ok;
  double phi[ASLR_MAX_NJ]; // angle of the joint placement rotation (atan2(sphi, cphi))
  double gx, gy;
  double cphi[ASLR_MAX_NJ], sphi[ASLR_MAX_NJ], px[ASLR_MAX_NJ], py[ASLR_MAX_NJ], pz[ASLR_MAX_NJ];
  double m[ASLR_MAX_NJ], cx[ASLR_MAX_NJ], cy[ASLR_MAX_NJ], izz[ASLR_MAX_NJ];
  // two-link chains: constants of the closed-form joint-space inertia and nonlinear effects (ChainPlanar<2>)
  //   [K1, J2, mA, mB, d1x, d1y, d2x, d2y], filled on the host
  double two[8];
};
struct DevDesc {
  aslr_chain_t chain;
  DevModel models[ASLR_MAX_MODELS];
  PlanarChain planar;
};

// ---------------------------------------------------------------------------------------------
// 3-D / spatial algebra (Pinocchio conventions: motion and force = [linear; angular])
// ---------------------------------------------------------------------------------------------
struct V3 { double x, y, z; };
struct SV { V3 lin, ang; };
struct M3 { double a[9]; };
struct SE3d { M3 R; V3 p; };

ASLR_DEV V3 v3(double x, double y, double z) { return V3{x, y, z}; }
ASLR_DEV V3 v3(const double *p) { return V3{p[0], p[1], p[2]}; }
ASLR_DEV V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
ASLR_DEV V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
ASLR_DEV V3 operator*(double s, V3 a) { return V3{s * a.x, s * a.y, s * a.z}; }
ASLR_DEV V3 neg(V3 a) { return V3{-a.x, -a.y, -a.z}; }
ASLR_DEV double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
ASLR_DEV V3 cross(V3 a, V3 b) {
  return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
ASLR_DEV V3 mul(const M3 &R, V3 v) {
  return V3{R.a[0] * v.x + R.a[1] * v.y + R.a[2] * v.z, R.a[3] * v.x + R.a[4] * v.y + R.a[5] * v.z,
            R.a[6] * v.x + R.a[7] * v.y + R.a[8] * v.z};
}
ASLR_DEV V3 mulT(const M3 &R, V3 v) {
  return V3{R.a[0] * v.x + R.a[3] * v.y + R.a[6] * v.z, R.a[1] * v.x + R.a[4] * v.y + R.a[7] * v.z,
            R.a[2] * v.x + R.a[5] * v.y + R.a[8] * v.z};
}
ASLR_DEV M3 mul(const M3 &A, const M3 &B) {
  M3 C;
  ASLR_UNROLL for (int i = 0; i < 3; ++i)
    ASLR_UNROLL for (int j = 0; j < 3; ++j)
      C.a[3 * i + j] = A.a[3 * i] * B.a[j] + A.a[3 * i + 1] * B.a[3 + j] + A.a[3 * i + 2] * B.a[6 + j];
  return C;
}
ASLR_DEV M3 mulTN(const M3 &A, const M3 &B) { // A^T B
  M3 C;
  ASLR_UNROLL for (int i = 0; i < 3; ++i)
    ASLR_UNROLL for (int j = 0; j < 3; ++j)
      C.a[3 * i + j] = A.a[i] * B.a[j] + A.a[3 + i] * B.a[3 + j] + A.a[6 + i] * B.a[6 + j];
  return C;
}
ASLR_DEV M3 m3(const double *p) {
  M3 R;
  ASLR_UNROLL for (int i = 0; i < 9; ++i) R.a[i] = p[i];
  return R;
}
// The chain table is read-only for the lifetime of the problem.  Read through the constant address space, its
// wave-uniform entries are fetched by scalar loads into SGPRs (s_load) instead of one vector load per lane that
// the compiler must otherwise keep ordered with the kernel's global stores.
typedef const aslr_chain_t __attribute__((address_space(4))) *chain_cp;
typedef const double __attribute__((address_space(4))) *cdp;
ASLR_DEV chain_cp chain_const(const aslr_chain_t *c) { return (chain_cp)c; }
ASLR_DEV V3 v3(cdp p) { return V3{p[0], p[1], p[2]}; }
ASLR_DEV M3 m3(cdp p) {
  M3 m;
  ASLR_UNROLL for (int i = 0; i < 9; ++i) m.a[i] = p[i];
  return m;
}
ASLR_DEV SV operator+(SV a, SV b) { return SV{a.lin + b.lin, a.ang + b.ang}; }
ASLR_DEV SV sv_zero() { return SV{V3{0, 0, 0}, V3{0, 0, 0}}; }
// Rodrigues rotation about a unit axis (JointModelRevoluteUnaligned)
ASLR_DEV M3 axis_angle_sc(V3 ax, double s, double c) {
  const double v = 1.0 - c;
  M3 R;
  R.a[0] = ax.x * ax.x * v + c;        R.a[1] = ax.x * ax.y * v - ax.z * s; R.a[2] = ax.x * ax.z * v + ax.y * s;
  R.a[3] = ax.y * ax.x * v + ax.z * s; R.a[4] = ax.y * ax.y * v + c;        R.a[5] = ax.y * ax.z * v - ax.x * s;
  R.a[6] = ax.z * ax.x * v - ax.y * s; R.a[7] = ax.z * ax.y * v + ax.x * s; R.a[8] = ax.z * ax.z * v + c;
  return R;
}
ASLR_DEV M3 axis_angle(V3 ax, double q) {
  double s, c;
  sincos_fast(q, &s, &c);
  return axis_angle_sc(ax, s, c);
}
ASLR_DEV SE3d se3_mul(const SE3d &A, const SE3d &B) { return SE3d{mul(A.R, B.R), mul(A.R, B.p) + A.p}; }
// motion: child -> parent
ASLR_DEV SV motion_act(const SE3d &M, SV m) {
  V3 w = mul(M.R, m.ang);
  return SV{mul(M.R, m.lin) + cross(M.p, w), w};
}
// motion: parent -> child
ASLR_DEV SV motion_actinv(const SE3d &M, SV m) {
  return SV{mulT(M.R, m.lin - cross(M.p, m.ang)), mulT(M.R, m.ang)};
}
// force: child -> parent
ASLR_DEV SV force_act(const SE3d &M, SV f) {
  V3 l = mul(M.R, f.lin);
  return SV{l, mul(M.R, f.ang) + cross(M.p, l)};
}
ASLR_DEV SV crm(SV a, SV b) { return SV{cross(a.ang, b.lin) + cross(a.lin, b.ang), cross(a.ang, b.ang)}; }
ASLR_DEV SV crf(SV a, SV f) { return SV{cross(a.ang, f.lin), cross(a.ang, f.ang) + cross(a.lin, f.lin)}; }
ASLR_DEV SV inertia_mul(double mass, V3 c, const M3 &I, SV m) {
  V3 l = mass * (m.lin - cross(c, m.ang));
  return SV{l, mul(I, m.ang) + cross(c, l)};
}

// ---------------------------------------------------------------------------------------------
// SE(3) log map and its Jacobian (Pinocchio 2.6 explog: acos on the trace, branch near pi)
// ---------------------------------------------------------------------------------------------
constexpr double kTaylorPrec = 1.220703125e-04; // eps^(1/4)
constexpr double kPi = 3.14159265358979323846;

ASLR_DEV double log3(const M3 &R, V3 &w) {
  double tr = R.a[0] + R.a[4] + R.a[8], theta;
  if (tr >= 3.0) { tr = 3.0; theta = 0.0; }
  else if (tr <= -1.0) { tr = -1.0; theta = kPi; }
  else theta = acos((tr - 1.0) / 2.0);
  if (theta >= kPi - 1e-2) {
    const double cphi = -(tr - 1.0) / 2.0;
    const double beta = theta * theta / (1.0 + cphi);
    const double t0 = (R.a[0] + cphi) * beta, t1 = (R.a[4] + cphi) * beta, t2 = (R.a[8] + cphi) * beta;
    w.x = (R.a[7] > R.a[5] ? 1.0 : -1.0) * (t0 > 0.0 ? sqrt(t0) : 0.0);
    w.y = (R.a[2] > R.a[6] ? 1.0 : -1.0) * (t1 > 0.0 ? sqrt(t1) : 0.0);
    w.z = (R.a[3] > R.a[1] ? 1.0 : -1.0) * (t2 > 0.0 ? sqrt(t2) : 0.0);
  } else {
    const double t = ((theta > kTaylorPrec) ? theta / sin(theta) : 1.0) / 2.0;
    w.x = t * (R.a[7] - R.a[5]);
    w.y = t * (R.a[2] - R.a[6]);
    w.z = t * (R.a[3] - R.a[1]);
  }
  return theta;
}

// r = log6(M).vector = [v; w] (residual_frame_placement.py:14-15); also returns theta and w.
ASLR_DEV void log6(const SE3d &M, double *r, double &t, V3 &w) {
  t = log3(M.R, w);
  const double t2 = t * t;
  double alpha, beta;
  if (t < kTaylorPrec) {
    alpha = 1.0 - t2 / 12.0 - t2 * t2 / 720.0;
    beta = 1.0 / 12.0 + t2 / 720.0;
  } else {
    double st, ct;
    sincos(t, &st, &ct);
    alpha = t * st / (2.0 * (1.0 - ct));
    beta = 1.0 / t2 - st / (2.0 * t * (1.0 - ct));
  }
  const V3 wxp = cross(w, M.p);
  const double wp = dot(w, M.p);
  r[0] = alpha * M.p.x - 0.5 * wxp.x + beta * wp * w.x;
  r[1] = alpha * M.p.y - 0.5 * wxp.y + beta * wp * w.y;
  r[2] = alpha * M.p.z - 0.5 * wxp.z + beta * wp * w.z;
  r[3] = w.x; r[4] = w.y; r[5] = w.z;
}

ASLR_DEV M3 jlog3(double theta, V3 w) {
  const double t2 = theta * theta;
  double alpha, diag;
  if (theta < kTaylorPrec) {
    alpha = 1.0 / 12.0 + t2 / 720.0;
    diag = 0.5 * (2.0 - t2 / 6.0);
  } else {
    double st, ct;
    sincos(theta, &st, &ct);
    const double st_1mct = st / (1.0 - ct);
    alpha = 1.0 / t2 - st_1mct / (2.0 * theta);
    diag = 0.5 * (theta * st_1mct);
  }
  const double wv[3] = {w.x, w.y, w.z};
  M3 J;
  ASLR_UNROLL for (int i = 0; i < 3; ++i)
    ASLR_UNROLL for (int j = 0; j < 3; ++j) J.a[3 * i + j] = alpha * wv[i] * wv[j];
  J.a[0] += diag; J.a[4] += diag; J.a[8] += diag;
  J.a[1] -= 0.5 * w.z; J.a[2] += 0.5 * w.y;
  J.a[3] += 0.5 * w.z; J.a[5] -= 0.5 * w.x;
  J.a[6] -= 0.5 * w.y; J.a[7] += 0.5 * w.x;
  return J;
}

// Jlog6(M) = [[A, B], [0, A]] (residual_frame_placement.py:19); theta, w from log6
ASLR_DEV void jlog6(const SE3d &M, double t, V3 w, M3 &A, M3 &Bm) {
  const double t2 = t * t;
  double beta, bdot;
  if (t < kTaylorPrec) {
    beta = 1.0 / 12.0 + t2 / 720.0;
    bdot = 1.0 / 360.0;
  } else {
    const double tinv = 1.0 / t, t2inv = tinv * tinv;
    double st, ct;
    sincos(t, &st, &ct);
    const double inv_2_2ct = 1.0 / (2.0 * (1.0 - ct));
    beta = t2inv - st * tinv * inv_2_2ct;
    bdot = -2.0 * t2inv * t2inv + (1.0 + st * tinv) * t2inv * inv_2_2ct;
  }
  A = jlog3(t, w);
  const V3 p = M.p;
  const double wTp = dot(w, p);
  const V3 v3t = (bdot * wTp) * w - (t2 * bdot + 2.0 * beta) * p;
  const double vv[3] = {v3t.x, v3t.y, v3t.z}, wv[3] = {w.x, w.y, w.z}, pv[3] = {p.x, p.y, p.z};
  M3 Cm;
  ASLR_UNROLL for (int i = 0; i < 3; ++i)
    ASLR_UNROLL for (int j = 0; j < 3; ++j) Cm.a[3 * i + j] = vv[i] * wv[j] + beta * wv[i] * pv[j];
  Cm.a[0] += wTp * beta; Cm.a[4] += wTp * beta; Cm.a[8] += wTp * beta;
  Cm.a[1] -= 0.5 * p.z; Cm.a[2] += 0.5 * p.y;
  Cm.a[3] += 0.5 * p.z; Cm.a[5] -= 0.5 * p.x;
  Cm.a[6] -= 0.5 * p.y; Cm.a[7] += 0.5 * p.x;
  Bm = mul(Cm, A);
}

// ---------------------------------------------------------------------------------------------
// small dense helpers on register arrays
// ---------------------------------------------------------------------------------------------
// Cholesky LL^T in place (lower); returns true on a non-positive pivot
template <int N>
ASLR_DEV bool chol(double (&A)[N][N]) {
  bool bad = false;
  ASLR_UNROLL for (int j = 0; j < N; ++j) {
    double d = A[j][j];
    ASLR_UNROLL for (int k = 0; k < j; ++k) d -= A[j][k] * A[j][k];
    if (!(d > 0.0)) bad = true;
    d = sqrt(d);
    A[j][j] = d;
    ASLR_UNROLL for (int i = j + 1; i < N; ++i) {
      double s = A[i][j];
      ASLR_UNROLL for (int k = 0; k < j; ++k) s -= A[i][k] * A[j][k];
      A[i][j] = s / d;
    }
  }
  return bad;
}
template <int N>
ASLR_DEV void chol_solve(const double (&L)[N][N], double (&b)[N]) {
  ASLR_UNROLL for (int i = 0; i < N; ++i) {
    double s = b[i];
    ASLR_UNROLL for (int k = 0; k < i; ++k) s -= L[i][k] * b[k];
    b[i] = s / L[i][i];
  }
  ASLR_UNROLL for (int i = N - 1; i >= 0; --i) {
    double s = b[i];
    ASLR_UNROLL for (int k = i + 1; k < N; ++k) s -= L[k][i] * b[k];
    b[i] = s / L[i][i];
  }
}
// Cholesky with reciprocal pivots: L (lower, in place) and rinv[i] = 1 / L[i][i]; true on a
// non-positive pivot.  The triangular solves below multiply by rinv instead of dividing.
template <int N>
ASLR_DEV bool chol_r(double (&A)[N][N], double (&rinv)[N]) {
  bool bad = false;
  ASLR_UNROLL for (int j = 0; j < N; ++j) {
    double d = A[j][j];
    ASLR_UNROLL for (int k = 0; k < j; ++k) d -= A[j][k] * A[j][k];
    if (!(d > 0.0)) bad = true;
    d = sqrt(d);
    A[j][j] = d;
    rinv[j] = 1.0 / d;
    ASLR_UNROLL for (int i = j + 1; i < N; ++i) {
      double s = A[i][j];
      ASLR_UNROLL for (int k = 0; k < j; ++k) s -= A[i][k] * A[j][k];
      A[i][j] = s * rinv[j];
    }
  }
  return bad;
}
template <int N>
ASLR_DEV void chol_solve_r(const double (&L)[N][N], const double (&rinv)[N], double (&b)[N]) {
  ASLR_UNROLL for (int i = 0; i < N; ++i) {
    double s = b[i];
    ASLR_UNROLL for (int k = 0; k < i; ++k) s -= L[i][k] * b[k];
    b[i] = s * rinv[i];
  }
  ASLR_UNROLL for (int i = N - 1; i >= 0; --i) {
    double s = b[i];
    ASLR_UNROLL for (int k = i + 1; k < N; ++k) s -= L[k][i] * b[k];
    b[i] = s * rinv[i];
  }
}

template <int N>
ASLR_DEV void spd_inverse(const double (&A)[N][N], double (&Ainv)[N][N]) {
  double L[N][N];
  ASLR_UNROLL for (int i = 0; i < N; ++i)
    ASLR_UNROLL for (int j = 0; j < N; ++j) L[i][j] = A[i][j];
  chol<N>(L);
  ASLR_UNROLL for (int j = 0; j < N; ++j) {
    double e[N];
    ASLR_UNROLL for (int i = 0; i < N; ++i) e[i] = (i == j) ? 1.0 : 0.0;
    chol_solve<N>(L, e);
    ASLR_UNROLL for (int i = 0; i < N; ++i) Ainv[i][j] = e[i];
  }
}

// ---------------------------------------------------------------------------------------------
// chain dynamics, generic 3-D path
// ---------------------------------------------------------------------------------------------
template <int NJ>
struct Kin { // forward kinematics shared by everything at one q
  SE3d liMi[NJ];
};

// RNEA(q, v, a) with gravity; keeps what the tangent passes need when KEEP is set.
template <int NJ>
struct RneaWs {
  SV v[NJ], h[NJ], F[NJ], vJ[NJ], Xv[NJ], Xa[NJ];
};

template <int NJ, bool KEEP>
ASLR_DEV void rnea(chain_cp c, const Kin<NJ> &k, const double *v, const double *a, V3 grav,
                   double *tau, RneaWs<NJ> &w) {
  SV vp = sv_zero(), ap = SV{neg(grav), V3{0, 0, 0}};
  SV f[NJ];
  ASLR_UNROLL for (int i = 0; i < NJ; ++i) {
    const V3 ax = v3(c->axis[i]);
    const SV vJ = SV{V3{0, 0, 0}, v[i] * ax};
    const SV Xv = motion_actinv(k.liMi[i], vp);
    const SV vi = Xv + vJ;
    const SV Xa = motion_actinv(k.liMi[i], ap);
    SV ai = Xa + crm(vi, vJ);
    ai.ang = ai.ang + a[i] * ax;
    const V3 com = v3(c->com[i]);
    const M3 I = m3(c->inertia[i]);
    const SV h = inertia_mul(c->mass[i], com, I, vi);
    f[i] = inertia_mul(c->mass[i], com, I, ai) + crf(vi, h);
    if (KEEP) { w.v[i] = vi; w.h[i] = h; w.vJ[i] = vJ; w.Xv[i] = Xv; w.Xa[i] = Xa; }
    vp = vi;
    ap = ai;
  }
  ASLR_UNROLL for (int i = NJ - 1; i >= 0; --i) {
    tau[i] = dot(v3(c->axis[i]), f[i].ang);
    if (i > 0) f[i - 1] = f[i - 1] + force_act(k.liMi[i], f[i]);
    if (KEEP) w.F[i] = f[i];
  }
}

// Chain policy: generic fixed-base revolute chain in 3-D.
template <int NJ>
struct Chain3D {
  // chain constants: the generic path reads the (large) table in place
  struct Consts {
    chain_cp c;
    ASLR_DEV explicit Consts(const DevDesc &D, bool = false) : c(chain_const(&D.chain)) {}
  };
  const chain_cp c;
  Kin<NJ> kin;
  RneaWs<NJ> ws;
  SE3d oMi[NJ];
  ASLR_DEV explicit Chain3D(const Consts &cc) : c(cc.c) {}

  ASLR_DEV void setup(const double *q) {
    ASLR_UNROLL for (int i = 0; i < NJ; ++i) {
      const M3 Rj = axis_angle(v3(c->axis[i]), q[i]);
      kin.liMi[i].R = mul(m3(c->joint_R[i]), Rj);
      kin.liMi[i].p = v3(c->joint_p[i]);
    }
  }
  // data.nle = RNEA(q, v, 0)
  ASLR_DEV void nle(const double *v, double *out) {
    double zero[NJ];
    ASLR_UNROLL for (int i = 0; i < NJ; ++i) zero[i] = 0.0;
    rnea<NJ, false>(c, kin, v, zero, v3(c->gravity), out, ws);
  }
  // joint-space inertia: column j = RNEA(q, 0, e_j) without gravity; symmetrised like the Python
  // binding's data.M
  ASLR_DEV void mass(double (&M)[NJ][NJ]) {
    ASLR_UNROLL for (int j = 0; j < NJ; ++j) {
      SV ap = sv_zero();
      SV f[NJ];
      ASLR_UNROLL for (int i = 0; i < NJ; ++i) {
        if (i < j) { f[i] = sv_zero(); continue; }
        SV ai = (i == j) ? sv_zero() : motion_actinv(kin.liMi[i], ap);
        if (i == j) ai.ang = v3(c->axis[i]);
        f[i] = inertia_mul(c->mass[i], v3(c->com[i]), m3(c->inertia[i]), ai);
        ap = ai;
      }
      ASLR_UNROLL for (int i = NJ - 1; i >= 0; --i) {
        M[i][j] = dot(v3(c->axis[i]), f[i].ang);
        if (i > 0) f[i - 1] = f[i - 1] + force_act(kin.liMi[i], f[i]);
      }
    }
    ASLR_UNROLL for (int i = 0; i < NJ; ++i)
      ASLR_UNROLL for (int j = i + 1; j < NJ; ++j) {
        const double s = 0.5 * (M[i][j] + M[j][i]);
        M[i][j] = s;
        M[j][i] = s;
      }
  }
  // computeRNEADerivatives(q, v, a) by forward-mode differentiation of the recursion (one
  // direction per column): dq[i][j] = dtau_i/dq_j, dv[i][j] = dtau_i/dv_j.
  ASLR_DEV void rnea_derivatives(const double *v, const double *a, double (&dq)[NJ][NJ], double (&dv)[NJ][NJ]) {
    double tau[NJ];
    rnea<NJ, true>(c, kin, v, a, v3(c->gravity), tau, ws);
    ASLR_UNROLL for (int kind = 0; kind < 2; ++kind) {
      ASLR_UNROLL for (int j = 0; j < NJ; ++j) {
        const SV Sj = SV{V3{0, 0, 0}, v3(c->axis[j])};
        SV dvp = sv_zero(), dap = sv_zero();
        SV df[NJ];
        ASLR_UNROLL for (int i = 0; i < NJ; ++i) {
          if (i < j) { df[i] = sv_zero(); continue; } // nothing upstream of joint j moves
          SV dvi = (i == j) ? sv_zero() : motion_actinv(kin.liMi[i], dvp);
          SV dai = (i == j) ? sv_zero() : motion_actinv(kin.liMi[i], dap);
          if (i == j) {
            if (kind == 0) { dvi = crm(ws.Xv[i], Sj); dai = crm(ws.Xa[i], Sj); }
            else { dvi = Sj; dai = crm(ws.v[i], Sj); }
          }
          dai = dai + crm(dvi, ws.vJ[i]);
          const V3 com = v3(c->com[i]);
          const M3 I = m3(c->inertia[i]);
          df[i] = inertia_mul(c->mass[i], com, I, dai) + crf(dvi, ws.h[i]) +
                  crf(ws.v[i], inertia_mul(c->mass[i], com, I, dvi));
          dvp = dvi;
          dap = dai;
        }
        ASLR_UNROLL for (int i = NJ - 1; i >= 0; --i) {
          const double val = dot(v3(c->axis[i]), df[i].ang);
          if (kind == 0) dq[i][j] = val; else dv[i][j] = val;
          if (i > 0) {
            df[i - 1] = df[i - 1] + force_act(kin.liMi[i], df[i]);
            if (kind == 0 && i == j) df[i - 1] = df[i - 1] + force_act(kin.liMi[i], crf(Sj, ws.F[i]));
          }
        }
      }
    }
  }
  // world placement of joint fj (runtime index); also fills oMi for jac_col
  ASLR_DEV SE3d joint_world(int fj) {
    oMi[0] = kin.liMi[0];
    ASLR_UNROLL for (int i = 1; i < NJ; ++i) oMi[i] = se3_mul(oMi[i - 1], kin.liMi[i]);
    SE3d r = oMi[0];
    ASLR_UNROLL for (int i = 1; i < NJ; ++i) if (i == fj) r = oMi[i];
    return r;
  }
  // Cost derivatives when the dynamics were evaluated elsewhere: only sin / cos of the joint angles are kept (2 nj
  // doubles); a joint's world placement is rebuilt as a running product whenever it is needed -- the same products in
  // the same order as setup() + joint_world(), so the same bits -- instead of holding all nj placements (12 nj doubles,
  // which with the Jacobian and the cost Hessian of a 7-joint chain did not fit the register file: 4 KB of scratch)
  double sn_[NJ], cs_[NJ];
  SE3d run_;
  ASLR_DEV void setup_world(const double *q) {
    ASLR_UNROLL for (int i = 0; i < NJ; ++i) sincos_fast(q[i], &sn_[i], &cs_[i]);
  }
  ASLR_DEV SE3d local_of(int i) const {
    SE3d li;
    li.R = mul(m3(c->joint_R[i]), axis_angle_sc(v3(c->axis[i]), sn_[i], cs_[i]));
    li.p = v3(c->joint_p[i]);
    return li;
  }
  ASLR_DEV SE3d joint_world_ready(int fj) const {
    SE3d r = local_of(0), o = r;
    ASLR_UNROLL for (int i = 1; i < NJ; ++i) {
      if (i <= fj) { // wave-uniform (fj is a constant of the cost)
        r = se3_mul(r, local_of(i));
        o = r;
      }
    }
    return o;
  }
  // LOCAL frame Jacobian column j after setup_world(): to be called with j = 0, 1, 2, ... in this order (the world
  // placement of joint j is the running product advanced by one joint per call)
  ASLR_DEV SV jac_col_seq(int j, const SE3d &oMf) {
    if (j == 0) run_ = local_of(0); else run_ = se3_mul(run_, local_of(j));
    SE3d fMj;
    fMj.R = mulTN(oMf.R, run_.R);
    fMj.p = mulT(oMf.R, run_.p - oMf.p);
    return motion_act(fMj, SV{V3{0, 0, 0}, v3(c->axis[j])});
  }
  // world placement of joint fj alone (cost-only evaluations): the same products in the same order as setup() +
  // joint_world(), as a running product -- no per-joint arrays (they cost 3.4 KB of scratch per lane at nj = 7)
  ASLR_DEV static SE3d world_of(const Consts &cc, const double *q, int fj) {
    const chain_cp c = cc.c;
    SE3d r;
    ASLR_UNROLL for (int i = 0; i < NJ; ++i) {
      if (i <= fj) { // wave-uniform (fj is a constant of the cost)
        SE3d li;
        li.R = mul(m3(c->joint_R[i]), axis_angle(v3(c->axis[i]), q[i]));
        li.p = v3(c->joint_p[i]);
        if (i == 0) r = li; else r = se3_mul(r, li);
      }
    }
    return r;
  }
  // (planar chains only: ChainPlanar::reach_residual)
  ASLR_DEV static void reach_residual(const Consts &, const double *, int, const double *, double, double, double,
                                      const double *, double (&)[6]) {}
  // LOCAL frame Jacobian column j: (oMf^-1 oMj).act(S_j)
  ASLR_DEV SV jac_col(int j, const SE3d &oMf) const {
    SE3d fMj;
    fMj.R = mulTN(oMf.R, oMi[j].R);
    fMj.p = mulT(oMf.R, oMi[j].p - oMf.p);
    return motion_act(fMj, SV{V3{0, 0, 0}, v3(c->axis[j])});
  }
};

// ---------------------------------------------------------------------------------------------
// chain dynamics, planar path: every joint axis is +z of its frame and every joint placement is a
// rotation about z, so only (omega_z, v_x, v_y) / (n_z, f_x, f_y) carry the dynamics.  Same
// recursions as above on 3-vectors; offsets and gravity along z cannot load the joints.
// ---------------------------------------------------------------------------------------------
struct PV { double w, x, y; };      // planar motion (w: angular z) or force (w: moment z)
struct PX { double c, s, px, py; }; // liMi: x_parent = Rz(c, s) x_child + (px, py)

ASLR_DEV PV operator+(PV a, PV b) { return PV{a.w + b.w, a.x + b.x, a.y + b.y}; }
ASLR_DEV PV pv_zero() { return PV{0.0, 0.0, 0.0}; }
ASLR_DEV PV p_actinv(PX X, PV m) { // motion parent -> child
  const double vx = m.x - X.py * m.w, vy = m.y + X.px * m.w;
  return PV{m.w, X.c * vx + X.s * vy, X.c * vy - X.s * vx};
}
ASLR_DEV PV p_fact(PX X, PV f) { // force child -> parent
  const double lx = X.c * f.x - X.s * f.y, ly = X.s * f.x + X.c * f.y;
  return PV{f.w + X.px * ly - X.py * lx, lx, ly};
}
ASLR_DEV PV p_crm(PV a, PV b) { return PV{0.0, b.w * a.y - a.w * b.y, a.w * b.x - b.w * a.x}; }
ASLR_DEV PV p_crf(PV a, PV f) { return PV{a.x * f.y - a.y * f.x, -a.w * f.y, a.w * f.x}; }
ASLR_DEV PV p_inertia(double m, double cx, double cy, double izz, PV v) {
  const double lx = m * (v.x - cy * v.w), ly = m * (v.y + cx * v.w);
  return PV{izz * v.w + cx * ly - cy * lx, lx, ly};
}

template <int NJ>
struct ChainPlanar {
  // chain constants copied once into registers (uniform values end up in SGPRs), so a kernel that
  // evaluates many knots per lane does not re-fetch them every knot
  struct Consts {
    double gx, gy, cphi[NJ], sphi[NJ], px[NJ], py[NJ], pz[NJ], m[NJ], cx[NJ], cy[NJ], izz[NJ], phi[NJ];
    double two[8];
    SinCosK sck;
    ASLR_DEV explicit Consts(const DevDesc &D, bool loop_kernel = false) : sck(loop_kernel) {
      const PlanarChain &p = D.planar;
      gx = p.gx; gy = p.gy;
      ASLR_UNROLL for (int i = 0; i < NJ; ++i) {
        phi[i] = p.phi[i];
        cphi[i] = p.cphi[i]; sphi[i] = p.sphi[i]; px[i] = p.px[i]; py[i] = p.py[i]; pz[i] = p.pz[i];
        m[i] = p.m[i]; cx[i] = p.cx[i]; cy[i] = p.cy[i]; izz[i] = p.izz[i];
      }
      ASLR_UNROLL for (int i = 0; i < 8; ++i) two[i] = p.two[i];
    }
  };
  const Consts &pc;
  PX X[NJ];
  PV v_[NJ], h_[NJ], F_[NJ], Xv_[NJ], Xa_[NJ];
  double qd_[NJ];
  double cT[NJ], sT[NJ], Px[NJ], Py[NJ]; // world angle / position of each joint frame
  ASLR_DEV explicit ChainPlanar(const Consts &cc) : pc(cc) {}

  ASLR_DEV void setup(const double *q) {
    ASLR_UNROLL for (int i = 0; i < NJ; ++i) {
      double s, c;
      sincos_fast(pc.sck, q[i], &s, &c);
      // Rz(phi_i) Rz(q_i)
      X[i] = PX{pc.cphi[i] * c - pc.sphi[i] * s, pc.sphi[i] * c + pc.cphi[i] * s, pc.px[i], pc.py[i]};
    }
  }
  template <bool KEEP>
  ASLR_DEV void rnea_(const double *v, const double *a, double *tau) {
    PV vp = pv_zero(), ap = PV{0.0, -pc.gx, -pc.gy};
    PV f[NJ];
    ASLR_UNROLL for (int i = 0; i < NJ; ++i) {
      const PV Xv = p_actinv(X[i], vp);
      PV vi = Xv;
      vi.w += v[i];
      const PV Xa = p_actinv(X[i], ap);
      PV ai = Xa + p_crm(vi, PV{v[i], 0.0, 0.0});
      ai.w += a[i];
      const PV h = p_inertia(pc.m[i], pc.cx[i], pc.cy[i], pc.izz[i], vi);
      f[i] = p_inertia(pc.m[i], pc.cx[i], pc.cy[i], pc.izz[i], ai) + p_crf(vi, h);
      if (KEEP) { v_[i] = vi; h_[i] = h; Xv_[i] = Xv; Xa_[i] = Xa; qd_[i] = v[i]; }
      vp = vi;
      ap = ai;
    }
    ASLR_UNROLL for (int i = NJ - 1; i >= 0; --i) {
      tau[i] = f[i].w;
      if (i > 0) f[i - 1] = f[i - 1] + p_fact(X[i], f[i]);
      if (KEEP) F_[i] = f[i];
    }
  }
  // Two links: M(q) and the nonlinear effects in closed form (same quantities as the recursions below; about a
  // quarter of their instructions, which matters on the serial path of the rollout).  With theta_i = phi_i + q_i,
  // p2 / c_i the offset of joint 2 / the centres of mass, E = m2 p2 . R(theta2) c2:
  //   M = [[K1 + 2E, J2 + E], [J2 + E, J2]],  C v = E' [ (2 v1 + v2) v2, -v1^2 ],
  //   G2 = -(R2^T R1^T g) . m2 c2^perp,  G1 = G2 - (R1^T g) . (m1 c1^perp + m2 p2^perp).
  ASLR_DEV void nle2(const double *v, double *out) const {
    const double c1 = X[0].c, s1 = X[0].s, c2 = X[1].c, s2 = X[1].s;
    const double Ep = pc.two[3] * c2 - pc.two[2] * s2;
    const double ux = c1 * pc.gx + s1 * pc.gy, uy = c1 * pc.gy - s1 * pc.gx;
    const double wx = c2 * ux + s2 * uy, wy = c2 * uy - s2 * ux;
    const double G2 = -(wx * pc.two[6] + wy * pc.two[7]);
    const double G1 = G2 - (ux * pc.two[4] + uy * pc.two[5]);
    out[0] = Ep * ((2.0 * v[0] + v[1]) * v[1]) + G1;
    out[1] = G2 - Ep * (v[0] * v[0]);
  }
  ASLR_DEV void mass2(double (&M)[NJ][NJ]) const {
    const double E = pc.two[2] * X[1].c + pc.two[3] * X[1].s;
    M[0][0] = pc.two[0] + 2.0 * E;
    M[0][1] = pc.two[1] + E;
    M[1][0] = M[0][1];
    M[1][1] = pc.two[1];
  }
  ASLR_DEV void nle(const double *v, double *out) {
    if constexpr (NJ == 2) { nle2(v, out); return; }
    double zero[NJ];
    ASLR_UNROLL for (int i = 0; i < NJ; ++i) zero[i] = 0.0;
    rnea_<false>(v, zero, out);
  }
  ASLR_DEV void mass(double (&M)[NJ][NJ]) {
    if constexpr (NJ == 2) { mass2(M); return; }
    ASLR_UNROLL for (int j = 0; j < NJ; ++j) {
      PV ap = pv_zero();
      PV f[NJ];
      ASLR_UNROLL for (int i = 0; i < NJ; ++i) {
        if (i < j) { f[i] = pv_zero(); continue; }
        const PV ai = (i == j) ? PV{1.0, 0.0, 0.0} : p_actinv(X[i], ap);
        f[i] = p_inertia(pc.m[i], pc.cx[i], pc.cy[i], pc.izz[i], ai);
        ap = ai;
      }
      ASLR_UNROLL for (int i = NJ - 1; i >= 0; --i) {
        M[i][j] = f[i].w;
        if (i > 0) f[i - 1] = f[i - 1] + p_fact(X[i], f[i]);
      }
    }
    ASLR_UNROLL for (int i = 0; i < NJ; ++i)
      ASLR_UNROLL for (int j = i + 1; j < NJ; ++j) {
        const double s = 0.5 * (M[i][j] + M[j][i]);
        M[i][j] = s;
        M[j][i] = s;
      }
  }
  ASLR_DEV void rnea_derivatives(const double *v, const double *a, double (&dq)[NJ][NJ], double (&dv)[NJ][NJ]) {
    double tau[NJ];
    rnea_<true>(v, a, tau);
    const PV S = PV{1.0, 0.0, 0.0};
    ASLR_UNROLL for (int kind = 0; kind < 2; ++kind) {
      ASLR_UNROLL for (int j = 0; j < NJ; ++j) {
        PV dvp = pv_zero(), dap = pv_zero();
        PV df[NJ];
        ASLR_UNROLL for (int i = 0; i < NJ; ++i) {
          if (i < j) { df[i] = pv_zero(); continue; }
          PV dvi = (i == j) ? pv_zero() : p_actinv(X[i], dvp);
          PV dai = (i == j) ? pv_zero() : p_actinv(X[i], dap);
          if (i == j) {
            if (kind == 0) { dvi = p_crm(Xv_[i], S); dai = p_crm(Xa_[i], S); }
            else { dvi = S; dai = p_crm(v_[i], S); }
          }
          dai = dai + p_crm(dvi, PV{qd_[i], 0.0, 0.0});
          df[i] = p_inertia(pc.m[i], pc.cx[i], pc.cy[i], pc.izz[i], dai) + p_crf(dvi, h_[i]) +
                  p_crf(v_[i], p_inertia(pc.m[i], pc.cx[i], pc.cy[i], pc.izz[i], dvi));
          dvp = dvi;
          dap = dai;
        }
        ASLR_UNROLL for (int i = NJ - 1; i >= 0; --i) {
          if (kind == 0) dq[i][j] = df[i].w; else dv[i][j] = df[i].w;
          if (i > 0) {
            df[i - 1] = df[i - 1] + p_fact(X[i], df[i]);
            if (kind == 0 && i == j) df[i - 1] = df[i - 1] + p_fact(X[i], p_crf(S, F_[i]));
          }
        }
      }
    }
  }
  ASLR_DEV SE3d joint_world(int fj) {
    cT[0] = X[0].c; sT[0] = X[0].s; Px[0] = X[0].px; Py[0] = X[0].py;
    ASLR_UNROLL for (int i = 1; i < NJ; ++i) {
      cT[i] = cT[i - 1] * X[i].c - sT[i - 1] * X[i].s;
      sT[i] = sT[i - 1] * X[i].c + cT[i - 1] * X[i].s;
      Px[i] = Px[i - 1] + (cT[i - 1] * X[i].px - sT[i - 1] * X[i].py);
      Py[i] = Py[i - 1] + (sT[i - 1] * X[i].px + cT[i - 1] * X[i].py);
    }
    double c = cT[0], s = sT[0], x = Px[0], y = Py[0], z = pc.pz[0];
    ASLR_UNROLL for (int i = 1; i < NJ; ++i) if (i == fj) { c = cT[i]; s = sT[i]; x = Px[i]; y = Py[i]; z = pc.pz[i]; }
    SE3d r;
    r.R.a[0] = c; r.R.a[1] = -s; r.R.a[2] = 0.0;
    r.R.a[3] = s; r.R.a[4] = c;  r.R.a[5] = 0.0;
    r.R.a[6] = 0.0; r.R.a[7] = 0.0; r.R.a[8] = 1.0;
    r.p = V3{x, y, z};
    return r;
  }
  ASLR_DEV void setup_world(const double *q) { setup(q); }
  ASLR_DEV SE3d joint_world_ready(int fj) { return joint_world(fj); }
  ASLR_DEV SV jac_col_seq(int j, const SE3d &oMf) const { return jac_col(j, oMf); }
  ASLR_DEV static SE3d world_of(const Consts &cc, const double *q, int fj) {
    ChainPlanar ch(cc);
    ch.setup(q);
    return ch.joint_world(fj);
  }
  // Frame-placement residual r = log6(Mref^-1 oMf).vector (residual_frame_placement.py:13-15) in closed form, for a
  // frame on joint fj turned about z by phiF (cF, sF = its cos, sin; Fp its offset) and a reference placement with
  // identity rotation at pref: the relative rotation is Rz(psi) with psi = sum_{i <= fj} (phi_i + q_i) + phiF wrapped
  // to (-pi, pi], so the angular part is w = (0, 0, psi) and, with p = oMf.p - pref and t = |psi|,
  //   r = [alpha p_x + psi p_y / 2, alpha p_y - psi p_x / 2, p_z, 0, 0, psi],   alpha = t sin t / (2 (1 - cos t)):
  // log6()'s formulas with w x p = psi (-p_y, p_x, 0); on the z component alpha + beta t^2 = 1.  No acos, no 3-D
  // products, one division -- a cost-only evaluation is otherwise dominated by the general SE(3) log.
  ASLR_DEV static void reach_residual(const Consts &cc, const double *q, int fj, const double *Fp, double cF, double sF,
                                      double phiF, const double *pref, double (&r)[6]) {
    double th = 0.0, Px = 0.0, Py = 0.0, Pz = 0.0, c = 1.0, s = 0.0; // world angle / origin of the frame reached so far
    ASLR_UNROLL for (int i = 0; i < NJ; ++i) {
      if (i <= fj) { // wave-uniform (fj is a constant of the cost)
        Px += c * cc.px[i] - s * cc.py[i];
        Py += s * cc.px[i] + c * cc.py[i];
        Pz = cc.pz[i];
        th += cc.phi[i] + q[i];
        sincos_fast(cc.sck, th, &s, &c);
      }
    }
    const double px_ = (Px + (c * Fp[0] - s * Fp[1])) - pref[0];
    const double py_ = (Py + (s * Fp[0] + c * Fp[1])) - pref[1];
    const double pz_ = (Pz + Fp[2]) - pref[2];
    // psi in (-pi, pi]: two-term reduction by 2 pi
    double psi = th + phiF;
    const double kk = rint(psi * 1.59154943091895335769e-01);
    psi = fma(-kk, 6.28318530717958623200e+00, psi) - kk * 2.44929359829470635445e-16;
    const double cr = c * cF - s * sF, sr = s * cF + c * sF; // (cF = 1, sF = 0 leave c, s unchanged, bit for bit)
    const double t = fabs(psi), st = fabs(sr), ct = cr;
    const double t2 = t * t;
    const double alpha = t < kTaylorPrec ? 1.0 - t2 / 12.0 - t2 * t2 / 720.0 : t * st / (2.0 * (1.0 - ct));
    r[0] = alpha * px_ + 0.5 * psi * py_;
    r[1] = alpha * py_ - 0.5 * psi * px_;
    r[2] = pz_;
    r[3] = 0.0; r[4] = 0.0; r[5] = psi;
  }
  ASLR_DEV SV jac_col(int j, const SE3d &oMf) const {
    // joint j turns the frame about the world z axis through (Px[j], Py[j])
    const V3 lin = V3{-(oMf.p.y - Py[j]), oMf.p.x - Px[j], 0.0};
    return SV{mulT(oMf.R, lin), V3{oMf.R.a[6], oMf.R.a[7], oMf.R.a[8]}};
  }
};

// DYN record of a knot (region DYN, chains with nj > 2: written by dyn_team_kernel, read by calc_kernel<..., PRE>), doubles:
//   [ xout (2 nj) | M^-1 (nj x nj, row-major) | pad to even | nj rows [ Aqq[i][:] | Aqm[i][:] | Aqv[i][:] | pad to even ] ]
// i.e. the link rows of da_dx in the column order of an Fx row, so the record assembly reads them where it needs them
constexpr int dyn_row_c(int nj) { return (3 * nj + 1) / 2 * 2; }
constexpr int dyn_oa_c(int nj) { return (2 * nj + nj * nj + 1) / 2 * 2; }
constexpr int dyn_len_c(int nj) { return nj > 2 ? dyn_oa_c(nj) + nj * dyn_row_c(nj) : 0; }

// ---------------------------------------------------------------------------------------------
// knot-level results
// ---------------------------------------------------------------------------------------------
// Compact derivative set of one knot.  The continuous Fx of the SEA/VSA models has the block
// structure [ddq_dq | Minv K | ddq_dv | 0 ; Binv K | -Binv K | 0 | 0] (free_fwddyn_asr.py:78-86),
// Lxx of the package's cost types is (top-left nj x nj block) + diagonal, Luu diagonal, Lxu = 0.
template <int NJ, int NU>
struct KnotDiff {
  double Aqq[NJ][NJ], Aqm[NJ][NJ], Aqv[NJ][NJ], Bk[NJ][NJ]; // da_dx blocks
  double Ful[NJ][NU], Fum[NJ][NU];                          // da_du link / motor rows
  double Lx[4 * NJ], Lu[NU], Lqq[NJ][NJ], Lxxd[4 * NJ], Luud[NU];
  // kEvalPre: the link rows of da_dx stay in the knot's DYN record (dyn_team_kernel wrote them); rec_elem<..., LAZY> reads
  // an entry there when it assembles the record line that holds it, so the 3 nj^2 values are never live together
  const double *arows;
  // kEvalLqqMem: where Lqq[j][l] lives instead (element j * NJ + l); rec_elem<..., LAZY> reads it there
  double *lqq_mem;
};

template <int NJ, int DAM> struct ModelDims {
  static constexpr int nx = 4 * NJ;
  static constexpr int nu = DAM == ASLR_DAM_VSA ? 2 * NJ : NJ;
};

// SE(3) log of the frame-placement residual with the sin/cos of its angle shared between
// log6, Jlog3 and Jlog6 (same formulas as log6()/jlog6() above).
struct Log6 {
  double r[6], t, st, ct;
  V3 w;
};
ASLR_DEV void log6_shared(const SE3d &M, Log6 &o) {
  // log3 (same branches as log3() above) with sin(theta), cos(theta) taken from the trace:
  // cos(theta) = (tr - 1) / 2 and sin(theta) = sqrt((1 - c)(1 + c)) for theta = acos(c) in [0, pi]
  const M3 &R = M.R;
  double tr = R.a[0] + R.a[4] + R.a[8], theta;
  if (tr >= 3.0) { tr = 3.0; theta = 0.0; }
  else if (tr <= -1.0) { tr = -1.0; theta = kPi; }
  else theta = acos((tr - 1.0) / 2.0);
  o.ct = (tr - 1.0) / 2.0;
  o.st = sqrt((1.0 - o.ct) * (1.0 + o.ct));
  if (theta >= kPi - 1e-2) {
    const double cphi = -(tr - 1.0) / 2.0;
    const double beta = theta * theta / (1.0 + cphi);
    const double t0 = (R.a[0] + cphi) * beta, t1 = (R.a[4] + cphi) * beta, t2 = (R.a[8] + cphi) * beta;
    o.w.x = (R.a[7] > R.a[5] ? 1.0 : -1.0) * (t0 > 0.0 ? sqrt(t0) : 0.0);
    o.w.y = (R.a[2] > R.a[6] ? 1.0 : -1.0) * (t1 > 0.0 ? sqrt(t1) : 0.0);
    o.w.z = (R.a[3] > R.a[1] ? 1.0 : -1.0) * (t2 > 0.0 ? sqrt(t2) : 0.0);
  } else {
    const double t = ((theta > kTaylorPrec) ? theta / o.st : 1.0) / 2.0;
    o.w.x = t * (R.a[7] - R.a[5]);
    o.w.y = t * (R.a[2] - R.a[6]);
    o.w.z = t * (R.a[3] - R.a[1]);
  }
  o.t = theta;
  const double t = o.t, t2 = t * t;
  double alpha, beta;
  if (t < kTaylorPrec) {
    alpha = 1.0 - t2 / 12.0 - t2 * t2 / 720.0;
    beta = 1.0 / 12.0 + t2 / 720.0;
  } else {
    alpha = t * o.st / (2.0 * (1.0 - o.ct));
    beta = 1.0 / t2 - o.st / (2.0 * t * (1.0 - o.ct));
  }
  const V3 wxp = cross(o.w, M.p);
  const double wp = dot(o.w, M.p);
  o.r[0] = alpha * M.p.x - 0.5 * wxp.x + beta * wp * o.w.x;
  o.r[1] = alpha * M.p.y - 0.5 * wxp.y + beta * wp * o.w.y;
  o.r[2] = alpha * M.p.z - 0.5 * wxp.z + beta * wp * o.w.z;
  o.r[3] = o.w.x; o.r[4] = o.w.y; o.r[5] = o.w.z;
}
ASLR_DEV void jlog6_shared(const SE3d &M, const Log6 &L, M3 &A, M3 &Bm) {
  const double t = L.t, t2 = t * t, st = L.st, ct = L.ct;
  const V3 w = L.w;
  double beta, bdot, alpha3, diag3;
  if (t < kTaylorPrec) {
    beta = 1.0 / 12.0 + t2 / 720.0;
    bdot = 1.0 / 360.0;
    alpha3 = 1.0 / 12.0 + t2 / 720.0;
    diag3 = 0.5 * (2.0 - t2 / 6.0);
  } else {
    const double tinv = 1.0 / t, t2inv = tinv * tinv;
    const double inv_2_2ct = 1.0 / (2.0 * (1.0 - ct));
    beta = t2inv - st * tinv * inv_2_2ct;
    bdot = -2.0 * t2inv * t2inv + (1.0 + st * tinv) * t2inv * inv_2_2ct;
    const double st_1mct = st / (1.0 - ct);
    alpha3 = 1.0 / t2 - st_1mct / (2.0 * t);
    diag3 = 0.5 * (t * st_1mct);
  }
  const double wv[3] = {w.x, w.y, w.z};
  ASLR_UNROLL for (int i = 0; i < 3; ++i)
    ASLR_UNROLL for (int j = 0; j < 3; ++j) A.a[3 * i + j] = alpha3 * wv[i] * wv[j];
  A.a[0] += diag3; A.a[4] += diag3; A.a[8] += diag3;
  A.a[1] -= 0.5 * w.z; A.a[2] += 0.5 * w.y;
  A.a[3] += 0.5 * w.z; A.a[5] -= 0.5 * w.x;
  A.a[6] -= 0.5 * w.y; A.a[7] += 0.5 * w.x;
  const V3 p = M.p;
  const double wTp = dot(w, p);
  const V3 v3t = (bdot * wTp) * w - (t2 * bdot + 2.0 * beta) * p;
  const double vv[3] = {v3t.x, v3t.y, v3t.z}, pv[3] = {p.x, p.y, p.z};
  M3 Cm;
  ASLR_UNROLL for (int i = 0; i < 3; ++i)
    ASLR_UNROLL for (int j = 0; j < 3; ++j) Cm.a[3 * i + j] = vv[i] * wv[j] + beta * wv[i] * pv[j];
  Cm.a[0] += wTp * beta; Cm.a[4] += wTp * beta; Cm.a[8] += wTp * beta;
  Cm.a[1] -= 0.5 * p.z; Cm.a[2] += 0.5 * p.y;
  Cm.a[3] += 0.5 * p.z; Cm.a[5] -= 0.5 * p.x;
  Cm.a[6] -= 0.5 * p.y; Cm.a[7] += 0.5 * p.x;
  Bm = mul(Cm, A);
}

// inverse of the SPD joint-space inertia via Cholesky with reciprocal pivots
template <int N>
ASLR_DEV void spd_inverse_fast(const double (&A)[N][N], double (&Ainv)[N][N]) {
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
  ASLR_UNROLL for (int j = 0; j < N; ++j) {
    double e[N];
    ASLR_UNROLL for (int i = 0; i < N; ++i) {
      double s = (i == j) ? 1.0 : 0.0;
      ASLR_UNROLL for (int k = 0; k < i; ++k) if (k >= j) s -= L[i][k] * e[k];
      e[i] = (i < j) ? 0.0 : s * rinv[i];
    }
    ASLR_UNROLL for (int i = N - 1; i >= 0; --i) {
      double s = e[i];
      ASLR_UNROLL for (int k = i + 1; k < N; ++k) s -= L[k][i] * e[k];
      e[i] = s * rinv[i];
    }
    ASLR_UNROLL for (int i = 0; i < N; ++i) Ainv[i][j] = e[i];
  }
}

// x = pinv(F) b from A = F^T F and g = F^T b (thresholded Jacobi eigen-decomposition; the oracle's
// pinv_normal_solve, same operation order): Crocoddyl's pseudoInverse for quasiStatic, rank-deficient F included
template <int N>
ASLR_DEV void pinv_normal_solve(int rows, double (&A)[N][N], const double (&g)[N], double (&x)[N]) {
  double V[N][N];
  ASLR_UNROLL for (int i = 0; i < N; ++i)
    ASLR_UNROLL for (int j = 0; j < N; ++j) V[i][j] = i == j ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 8; ++sweep) {
    ASLR_UNROLL for (int p = 0; p < N - 1; ++p)
      ASLR_UNROLL for (int q = p + 1; q < N; ++q) {
        const double apq = A[p][q];
        if (apq != 0.0) {
          const double theta = (A[q][q] - A[p][p]) / (2.0 * apq);
          const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
          const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
          ASLR_UNROLL for (int k = 0; k < N; ++k) {
            const double akp = A[k][p], akq = A[k][q];
            A[k][p] = c * akp - sn * akq;
            A[k][q] = sn * akp + c * akq;
          }
          ASLR_UNROLL for (int k = 0; k < N; ++k) {
            const double apk = A[p][k], aqk = A[q][k];
            A[p][k] = c * apk - sn * aqk;
            A[q][k] = sn * apk + c * aqk;
          }
          ASLR_UNROLL for (int k = 0; k < N; ++k) {
            const double vkp = V[k][p], vkq = V[k][q];
            V[k][p] = c * vkp - sn * vkq;
            V[k][q] = sn * vkp + c * vkq;
          }
        }
      }
  }
  double lmax = 0.0;
  ASLR_UNROLL for (int i = 0; i < N; ++i) lmax = fmax(lmax, fabs(A[i][i]));
  const double thr = 2.220446049250313e-16 * (double)(rows > N ? rows : N) * lmax;
  double y[N];
  ASLR_UNROLL for (int i = 0; i < N; ++i) {
    double a = 0.0;
    ASLR_UNROLL for (int k = 0; k < N; ++k) a += V[k][i] * g[k];
    y[i] = A[i][i] > thr ? a / A[i][i] : 0.0;
  }
  ASLR_UNROLL for (int k = 0; k < N; ++k) {
    double a = 0.0;
    ASLR_UNROLL for (int i = 0; i < N; ++i) a += V[k][i] * y[i];
    x[k] = a;
  }
}

// dynamics constants of one action model held in registers
template <int NJ, int NU>
struct ModelRegs {
  double dt, K[NJ][NJ], Binv[NJ][NJ], S[NJ][NU];
  ASLR_DEV void load(const DevModel &dm_) {
    // (constant address space: scalar loads, counted by lgkmcnt -- a vector load here made every knot of the
    //  rollout wait for its own outstanding candidate stores)
    const DevModel __attribute__((address_space(4))) &dm = *(const DevModel __attribute__((address_space(4))) *)(&dm_);
    dt = dm.m.dt;
    ASLR_UNROLL for (int i = 0; i < NJ; ++i) {
      ASLR_UNROLL for (int j = 0; j < NJ; ++j) { K[i][j] = dm.m.K[i * NJ + j]; Binv[i][j] = dm.Binv[i * NJ + j]; }
      ASLR_UNROLL for (int j = 0; j < NU; ++j) S[i][j] = dm.m.S[i * NU + j];
    }
  }
};

// what knot_eval computes
constexpr int kEvalDyn = 1;  // xnext (dynamics + Euler step)
constexpr int kEvalCost = 2; // cost
constexpr int kEvalDiff = 4; // compact derivatives (implies both of the above)
constexpr int kEvalPre = 8;  // the rigid-body part (xout, M^-1, the link rows of da_dx) is read from `pre` (DYN region)
constexpr int kEvalSkipCost = 16; // with kEvalDiff: dynamics and its derivatives only (first half of a split evaluation)
constexpr int kEvalSkipDyn = 32;  // with kEvalDiff: cost stack and its derivatives only (second half)
constexpr int kEvalResid = 64;    // also store the stacked cost residuals (data.r) through `resid`
constexpr int kEvalFastReach = 128; // cost-only evaluations on a planar chain with PlanarChain::reach_ok: frame-placement
                                    // residuals in closed form (ChainPlanar::reach_residual)
constexpr int kEvalLqqMem = 256;    // the nj x nj block of Lxx is accumulated in memory (KnotDiff::lqq_mem, e.g. LDS) instead of
                                    // registers: 2 nj^2 registers that the Jacobian of a 7-joint chain needs at the same time

// calc (+ calcDiff): x[4NJ], u[NU] -> xnext, cost (+ compact derivatives).
// u == nullptr selects the model's "u is None" default (terminal node).
// CH is the chain policy (Chain3D / ChainPlanar); WHAT a mask of kEval* bits.
template <int NJ, int DAM, int WHAT, class CH>
ASLR_DEV void knot_eval(const typename CH::Consts &cc, const ModelRegs<NJ, ModelDims<NJ, DAM>::nu> &mr,
                        const DevModel &dm, const double *frame_ref,
                        const double (&x)[4 * NJ], const double *u_in, double (&xnext)[4 * NJ],
                        double &cost_out, KnotDiff<NJ, ModelDims<NJ, DAM>::nu> *kd,
                        double *xout_o = nullptr, const double *pre = nullptr, double *resid = nullptr) {
  constexpr int NX = 4 * NJ, NU = ModelDims<NJ, DAM>::nu;
  constexpr bool RESID = (WHAT & kEvalResid) != 0;
  constexpr bool FASTREACH = (WHAT & kEvalFastReach) != 0 && (WHAT & kEvalDiff) == 0;
  int roff = 0; // running offset into `resid`
  constexpr bool DIFF = (WHAT & kEvalDiff) != 0;
  constexpr bool DYN = (DIFF || (WHAT & kEvalDyn) != 0) && (WHAT & kEvalSkipDyn) == 0;
  constexpr bool COST = (DIFF || (WHAT & kEvalCost) != 0) && (WHAT & kEvalSkipCost) == 0;
  constexpr bool PRE = (WHAT & kEvalPre) != 0;
  const aslr_model_t &m = dm.m;
  double u[NU];
  if (u_in) {
    ASLR_UNROLL for (int i = 0; i < NU; ++i) u[i] = u_in[i];
  } else {
    ASLR_UNROLL for (int i = 0; i < NU; ++i) u[i] = (DAM == ASLR_DAM_VSA && i >= NJ) ? 3.0 : 0.0;
  }
  double q[NJ], v[NJ], dqm[NJ];
  ASLR_UNROLL for (int i = 0; i < NJ; ++i) { q[i] = x[i]; v[i] = x[2 * NJ + i]; dqm[i] = x[i] - x[NJ + i]; }

  CH ch(cc);
  constexpr bool NEEDCH = DYN || DIFF; // a cost-only evaluation takes the lean forward kinematics below
  constexpr bool WORLDONLY = DIFF && !DYN; // cost derivatives only: joint placements, no parent-to-child transforms
  if constexpr (WORLDONLY) ch.setup_world(q); else if constexpr (NEEDCH) ch.setup(q);

  if (DYN) {
    // stiffness / coupling torque / motor torque
    double Kmat[NJ][NJ], tau_m[NJ], tau_c[NJ];
    if (DAM == ASLR_DAM_VSA) {
      ASLR_UNROLL for (int i = 0; i < NJ; ++i) {
        ASLR_UNROLL for (int j = 0; j < NJ; ++j) Kmat[i][j] = (i == j) ? u[NJ + i] : 0.0;
        tau_m[i] = u[i];
      }
    } else {
      ASLR_UNROLL for (int i = 0; i < NJ; ++i) {
        double s = 0.0;
        ASLR_UNROLL for (int j = 0; j < NJ; ++j) { Kmat[i][j] = mr.K[i][j]; s += mr.S[i][j] * u[j]; }
        tau_m[i] = s;
      }
    }
    ASLR_UNROLL for (int i = 0; i < NJ; ++i) {
      double s = 0.0;
      ASLR_UNROLL for (int j = 0; j < NJ; ++j) s += Kmat[i][j] * dqm[j];
      tau_c[i] = s;
    }
    double Minv[NJ][NJ], xout[2 * NJ];
    if constexpr (PRE) {
      ASLR_UNROLL for (int i = 0; i < 2 * NJ; ++i) xout[i] = pre[i];
      ASLR_UNROLL for (int i = 0; i < NJ; ++i)
        ASLR_UNROLL for (int j = 0; j < NJ; ++j) Minv[i][j] = pre[2 * NJ + i * NJ + j];
    } else {
      double M[NJ][NJ], nle[NJ];
      ch.nle(v, nle);
      ch.mass(M);
      spd_inverse_fast<NJ>(M, Minv);
      ASLR_UNROLL for (int i = 0; i < NJ; ++i) {
        double s = 0.0, s2 = 0.0;
        ASLR_UNROLL for (int j = 0; j < NJ; ++j) {
          s += Minv[i][j] * (-nle[j] - tau_c[j]);
          s2 += mr.Binv[i][j] * (tau_m[j] + tau_c[j]);
        }
        xout[i] = s;
        xout[NJ + i] = s2;
      }
    }
    if (xout_o) {
      ASLR_UNROLL for (int i = 0; i < 2 * NJ; ++i) xout_o[i] = xout[i];
    }
    // semi-implicit Euler (integrated_action.py:23-24)
    const double dt = mr.dt;
    ASLR_UNROLL for (int i = 0; i < 2 * NJ; ++i) {
      xnext[i] = x[i] + (x[2 * NJ + i] * dt + xout[i] * dt * dt);
      xnext[2 * NJ + i] = x[2 * NJ + i] + xout[i] * dt;
    }

    if (DIFF) {
      if constexpr (PRE) {
        // the team kernel has formed M^-1 [-dtau/dq - K | K | -dtau/dv] (same sums, same order); only the motor block is left
        kd->arows = pre + dyn_oa_c(NJ);
        ASLR_UNROLL for (int i = 0; i < NJ; ++i)
          ASLR_UNROLL for (int j = 0; j < NJ; ++j) {
            double bk = 0.0;
            ASLR_UNROLL for (int l = 0; l < NJ; ++l) bk += mr.Binv[i][l] * Kmat[l][j];
            kd->Bk[i][j] = bk;
          }
      } else {
        double ddq[NJ][NJ], ddv[NJ][NJ];
        ch.rnea_derivatives(v, xout, ddq, ddv);
        ASLR_UNROLL for (int i = 0; i < NJ; ++i)
          ASLR_UNROLL for (int j = 0; j < NJ; ++j) {
            double sq = 0.0, sk = 0.0, sv = 0.0, bk = 0.0;
            ASLR_UNROLL for (int l = 0; l < NJ; ++l) {
              sq += Minv[i][l] * (-ddq[l][j] - Kmat[l][j]);
              sk += Minv[i][l] * Kmat[l][j];
              sv += Minv[i][l] * (-ddv[l][j]);
              bk += mr.Binv[i][l] * Kmat[l][j];
            }
            kd->Aqq[i][j] = sq; kd->Aqm[i][j] = sk; kd->Aqv[i][j] = sv; kd->Bk[i][j] = bk;
          }
      }
      ASLR_UNROLL for (int i = 0; i < NJ; ++i)
        ASLR_UNROLL for (int j = 0; j < NU; ++j) { kd->Ful[i][j] = 0.0; kd->Fum[i][j] = 0.0; }
      if (DAM == ASLR_DAM_VSA) {
        ASLR_UNROLL for (int i = 0; i < NJ; ++i)
          ASLR_UNROLL for (int j = 0; j < NJ; ++j) {
            kd->Ful[i][NJ + j] = Minv[i][j] * (-x[j] + x[NJ + j]);
            kd->Fum[i][NJ + j] = mr.Binv[i][j] * (x[j] - x[NJ + j]);
            kd->Fum[i][j] = mr.Binv[i][j];
          }
      } else if (NU > 1) {
        ASLR_UNROLL for (int i = 0; i < NJ; ++i)
          ASLR_UNROLL for (int j = 0; j < NU; ++j) {
            double s = 0.0;
            ASLR_UNROLL for (int l = 0; l < NJ; ++l) s += mr.Binv[i][l] * mr.S[l][j];
            kd->Fum[i][j] = s;
          }
      }
    }
  }
  if (!COST) return;
  if (DIFF) {
    ASLR_UNROLL for (int i = 0; i < NX; ++i) { kd->Lx[i] = 0.0; kd->Lxxd[i] = 0.0; }
    ASLR_UNROLL for (int i = 0; i < NU; ++i) { kd->Lu[i] = 0.0; kd->Luud[i] = 0.0; }
    ASLR_UNROLL for (int i = 0; i < NJ; ++i)
      ASLR_UNROLL for (int j = 0; j < NJ; ++j) { if constexpr ((WHAT & kEvalLqqMem) != 0) kd->lqq_mem[i * NJ + j] = 0.0; else kd->Lqq[i][j] = 0.0; }
  }

  // ---- cost stack ----
  double cost = 0.0;
  for (int ci = 0; ci < m.ncosts; ++ci) {
    const aslr_cost_t &ct = m.costs[ci];
    const double w = ct.weight;
    if (FASTREACH && ct.type == ASLR_COST_FRAME_PLACEMENT) {
      if constexpr (FASTREACH) {
        const double *ref = frame_ref ? frame_ref : ct.ref;
        double r[6];
        CH::reach_residual(cc, q, ct.frame_joint, ct.frame_p, dm.fr_c[ci], dm.fr_s[ci], dm.fr_phi[ci], ref + 9, r);
        double a = 0.0;
        ASLR_UNROLL for (int i = 0; i < 6; ++i) a += ct.act_w[i] * r[i] * r[i];
        cost += w * 0.5 * a;
        if constexpr (RESID) { ASLR_UNROLL for (int i = 0; i < 6; ++i) resid[roff + i] = r[i]; roff += 6; }
      }
    } else if (ct.type == ASLR_COST_FRAME_PLACEMENT) {
      const int fj = ct.frame_joint;
      SE3d oMj;
      if constexpr (WORLDONLY) oMj = ch.joint_world_ready(fj);
      else if constexpr (NEEDCH) oMj = ch.joint_world(fj);
      else oMj = CH::world_of(cc, q, fj);
      const SE3d F = SE3d{m3(ct.frame_R), v3(ct.frame_p)};
      const SE3d oMf = se3_mul(oMj, F);
      const double *ref = frame_ref ? frame_ref : ct.ref;
      const M3 Rr = m3(ref);
      const V3 pr = v3(ref + 9);
      SE3d rMf; // Mref^-1 * oMf
      rMf.R = mulTN(Rr, oMf.R);
      rMf.p = mulT(Rr, oMf.p - pr);
      Log6 lg;
      log6_shared(rMf, lg);
      const double *r = lg.r;
      double a = 0.0;
      ASLR_UNROLL for (int i = 0; i < 6; ++i) a += ct.act_w[i] * r[i] * r[i];
      cost += w * 0.5 * a;
      if constexpr (RESID) { ASLR_UNROLL for (int i = 0; i < 6; ++i) resid[roff + i] = r[i]; roff += 6; }
      if (DIFF) {
        M3 A, Bm;
        jlog6_shared(rMf, lg, A, Bm);
        double Jr[6][NJ];
        ASLR_UNROLL for (int j = 0; j < NJ; ++j) {
          SV col;
          if constexpr (WORLDONLY) col = ch.jac_col_seq(j, oMf); else col = ch.jac_col(j, oMf);
          if (j > fj) col = sv_zero();
          const V3 top = mul(A, col.lin) + mul(Bm, col.ang);
          const V3 bot = mul(A, col.ang);
          Jr[0][j] = top.x; Jr[1][j] = top.y; Jr[2][j] = top.z;
          Jr[3][j] = bot.x; Jr[4][j] = bot.y; Jr[5][j] = bot.z;
        }
        ASLR_UNROLL for (int j = 0; j < NJ; ++j) {
          double s = 0.0;
          ASLR_UNROLL for (int i = 0; i < 6; ++i) s += Jr[i][j] * ct.act_w[i] * r[i];
          kd->Lx[j] += w * s;
          ASLR_UNROLL for (int l = 0; l < NJ; ++l) {
            double hh = 0.0;
            ASLR_UNROLL for (int i = 0; i < 6; ++i) hh += Jr[i][j] * ct.act_w[i] * Jr[i][l];
            if constexpr ((WHAT & kEvalLqqMem) != 0) kd->lqq_mem[j * NJ + l] += w * hh; else kd->Lqq[j][l] += w * hh;
          }
        }
      }
    } else if (ct.type == ASLR_COST_STATE) {
      double a = 0.0;
      ASLR_UNROLL for (int i = 0; i < NX; ++i) {
        const double r = x[i] - ct.ref[i];
        a += ct.act_w[i] * r * r;
        if (DIFF) { kd->Lx[i] += w * ct.act_w[i] * r; kd->Lxxd[i] += w * ct.act_w[i]; }
        if constexpr (RESID) resid[roff + i] = r;
      }
      if constexpr (RESID) roff += NX;
      cost += w * 0.5 * a;
    } else if (ct.type == ASLR_COST_CONTROL) {
      double a = 0.0;
      ASLR_UNROLL for (int i = 0; i < NU; ++i) {
        const double r = u[i] - ct.ref[i];
        a += ct.act_w[i] * r * r;
        if (DIFF) { kd->Lu[i] += w * ct.act_w[i] * r; kd->Luud[i] += w * ct.act_w[i]; }
        if constexpr (RESID) resid[roff + i] = r;
      }
      if constexpr (RESID) roff += NU;
      cost += w * 0.5 * a;
    } else if (ct.type == ASLR_COST_PENDULUM) {
      if (NJ >= 2) {
        double s1, c1, s2, c2;
        sincos(x[0], &s1, &c1);
        sincos(x[NJ >= 2 ? 1 : 0], &s2, &c2);
        const double r[6] = {s1, s2, 1.0 + c1, 1.0 + c2, x[NX > 4 ? 4 : 0], x[NX > 5 ? 5 : 0]};
        const double *aw = ct.act_w;
        double a = 0.0;
        ASLR_UNROLL for (int i = 0; i < 6; ++i) a += aw[i] * r[i] * r[i];
        cost += w * 0.5 * a;
        if constexpr (RESID) { ASLR_UNROLL for (int i = 0; i < 6; ++i) resid[roff + i] = r[i]; roff += 6; }
        if (DIFF) {
          kd->Lx[0] += w * (c1 * aw[0] * r[0] - s1 * aw[2] * r[2]);
          kd->Lx[NJ >= 2 ? 1 : 0] += w * (c2 * aw[1] * r[1] - s2 * aw[3] * r[3]);
          kd->Lx[NX > 4 ? 4 : 0] += w * aw[4] * r[4];
          kd->Lx[NX > 5 ? 5 : 0] += w * aw[5] * r[5];
          kd->Lxxd[0] += w * ((c1 * c1 - s1 * s1) * aw[0] + (s1 * s1 + (1.0 - c1) * c1) * aw[2]);
          kd->Lxxd[NJ >= 2 ? 1 : 0] += w * ((c2 * c2 - s2 * s2) * aw[1] + (s2 * s2 + (1.0 - c2) * c2) * aw[3]);
          kd->Lxxd[NX > 4 ? 4 : 0] += w * aw[4];
          kd->Lxxd[NX > 5 ? 5 : 0] += w * aw[5];
        }
      }
    } else if (ct.type == ASLR_COST_STIFFNESS) {
      constexpr int H = NU / 2;
      double a = 0.0;
      ASLR_UNROLL for (int i = 0; i < H; ++i) {
        a += ct.lambda * (u[H + i] - ct.ref[i]);
        if (DIFF) kd->Lu[H + i] += w * ct.lambda;
        if constexpr (RESID) resid[roff + i] = ct.lambda * (u[H + i] - ct.ref[i]);
      }
      if constexpr (RESID) roff += H;
      cost += w * a;
    }
  }
  cost_out = cost;
}

// ---------------------------------------------------------------------------------------------
// record element generator: element E (compile-time) of the DERIV record from the compact set.
// Arithmetic order follows integrated_action.py:31-37.
// ---------------------------------------------------------------------------------------------
template <int NJ, int NU>
struct RecLayout {
  static constexpr int NX = 4 * NJ, NV = 2 * NJ;
  static constexpr int oFx = 0, oFu = oFx + NX * NX, oLxx = oFu + NX * NU, oLxu = oLxx + NX * NX,
                       oLuu = oLxu + NX * NU, oLx = oLuu + NU * NU, oLu = oLx + NX, oEnd = oLu + NU;
  static constexpr int len = (oEnd + 15) / 16 * 16;
};

template <int NJ, int NU, int E, bool LAZY = false>
ASLR_DEV double rec_elem(const KnotDiff<NJ, NU> &k, double dt) {
  using L = RecLayout<NJ, NU>;
  constexpr int NX = L::NX, NV = L::NV;
  if constexpr (E < L::oFu) {
    constexpr int r = E / NX, cc = E % NX;
    constexpr int ri = r % NV; // row of da_dx
    double a; // da_dx[ri][cc]
    if constexpr (ri < NJ) {
      if constexpr (cc >= 3 * NJ) a = 0.0;
      else if constexpr (LAZY) a = k.arows[ri * dyn_row_c(NJ) + cc]; // (a DYN row is [Aqq | Aqm | Aqv] of that row)
      else if constexpr (cc < NJ) a = k.Aqq[ri][cc];
      else if constexpr (cc < 2 * NJ) a = k.Aqm[ri][cc - NJ];
      else a = k.Aqv[ri][cc - 2 * NJ];
    } else {
      if constexpr (cc < NJ) a = k.Bk[ri - NJ][cc];
      else if constexpr (cc < 2 * NJ) a = -k.Bk[ri - NJ][cc - NJ];
      else a = 0.0;
    }
    double val;
    if constexpr (r < NV) {
      double top = a * dt;
      if constexpr (cc == NV + r) top += 1.0;
      val = dt * top;
    } else {
      val = dt * a;
    }
    if constexpr (r == cc) val += 1.0;
    return val;
  } else if constexpr (E < L::oLxx) {
    constexpr int e = E - L::oFu, r = e / NU, cc = e % NU, ri = r % NV;
    double a;
    if constexpr (ri < NJ) a = k.Ful[ri][cc]; else a = k.Fum[ri - NJ][cc];
    if constexpr (r < NV) return dt * (a * dt);
    else return dt * a;
  } else if constexpr (E < L::oLxu) {
    constexpr int e = E - L::oLxx, r = e / NX, cc = e % NX;
    double val = 0.0;
    if constexpr (r < NJ && cc < NJ) { if constexpr (LAZY) val = k.lqq_mem[r * NJ + cc]; else val = k.Lqq[r][cc]; }
    if constexpr (r == cc) val += k.Lxxd[r];
    return val;
  } else if constexpr (E < L::oLuu) {
    return 0.0;
  } else if constexpr (E < L::oLx) {
    constexpr int e = E - L::oLuu, r = e / NU, cc = e % NU;
    if constexpr (r == cc) return k.Luud[r]; else return 0.0;
  } else if constexpr (E < L::oLu) {
    return k.Lx[E - L::oLx];
  } else if constexpr (E < L::oEnd) {
    return k.Lu[E - L::oLu];
  } else {
    return 0.0;
  }
}

// structural zero test of record element E: true when rec_elem<NJ, NU, E>() is 0.0 whatever the knot (the DERIV
// region is zero-filled at problem creation, so a 16-double chunk made of such elements never needs writing)
template <int NJ, int NU, int E>
constexpr bool rec_elem_is_zero() {
  using L = RecLayout<NJ, NU>;
  constexpr int NX = L::NX;
  if (E < L::oLxx) return false;                       // Fx / Fu: data (or 1, dt) everywhere that matters
  if (E < L::oLxu) {                                   // Lxx: the nj x nj block and the diagonal
    constexpr int e = E - L::oLxx, r = e / NX, cc = e % NX;
    return !((r < NJ && cc < NJ) || r == cc);
  }
  if (E < L::oLuu) return true;                        // Lxu
  if (E < L::oLx) {                                    // Luu: diagonal
    constexpr int e = E - L::oLuu, r = e / NU, cc = e % NU;
    return r != cc;
  }
  return E >= L::oEnd;                                 // Lx, Lu: data; padding: zero
}
// model-only test of record element E: structural zero, or a diagonal entry of Lxx outside the nj x nj block /
// of Luu, which is a sum of cost weights (state and control regularisers) and does not depend on the knot -- PROVIDED
// the cost stack has no cost with a state-dependent diagonal Hessian (CostModelDoublePendulum; the host checks)
// (SEA: Fu = dt [dt a_u; a_u] with a_u = [0; B^-1 S] -- free_fwddyn_asr.py:86-88 -- has no state in it either)
template <int NJ, int NU, int E, bool SEA = false>
constexpr bool rec_elem_is_model_only() {
  using L = RecLayout<NJ, NU>;
  constexpr int NX = L::NX;
  if (rec_elem_is_zero<NJ, NU, E>()) return true;
  if (SEA && E >= L::oFu && E < L::oLxx) return true;
  // SEA: the motor rows of da_dx are [B^-1 K | -B^-1 K | 0 | 0] (free_fwddyn_asr.py:82-85) -- constants of the model, and
  // so are the Fx rows built from them (rows nj .. 2nj-1 of each half of the Euler Jacobian, integrated_action.py:31-35)
  if (SEA && E < L::oFu) { constexpr int r = E / NX; return (r % L::NV) >= NJ; }
  if (E >= L::oLxx && E < L::oLxu) {
    constexpr int e = E - L::oLxx, r = e / NX, cc = e % NX;
    return r == cc && r >= NJ;
  }
  return E >= L::oLuu && E < L::oLx;
}
template <int NJ, int NU, int C, int CHUNK, bool SEA = false, int I = 0>
constexpr bool rec_chunk_is_model_only() {
  if constexpr (I == CHUNK) return true;
  else return rec_elem_is_model_only<NJ, NU, C * CHUNK + I, SEA>() && rec_chunk_is_model_only<NJ, NU, C, CHUNK, SEA, I + 1>();
}
template <int NJ, int NU, int C, int CHUNK, int I = 0>
constexpr bool rec_chunk_is_zero() {
  if constexpr (I == CHUNK) return true;
  else return rec_elem_is_zero<NJ, NU, C * CHUNK + I>() && rec_chunk_is_zero<NJ, NU, C, CHUNK, I + 1>();
}

// compile-time loop helper
template <int I, int N, class F>
ASLR_DEV void static_for(F &&f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

} // namespace aslr
