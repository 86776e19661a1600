/*
 * aslr_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see aslr_oracle.h for the parity status:
 * model code follows the reference's Python files; dynamics / solvers restate Pinocchio /
 * Crocoddyl's published algorithms; solver parity with Crocoddyl itself is "parity unpinned").
 *
 * Plain C, float64, no FMA contraction assumptions, straightforward loops.  Citations are
 * relative to the reference checkout (python/aslr_to/..., examples/...) or to SURVEY.md.
 */
#include "aslr_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define NJ ASLR_MAX_NJ
#define NX ASLR_MAX_NX
#define NU ASLR_MAX_NU

/* ======================================================================================= */
/* 3-D and spatial algebra (Pinocchio conventions: motion = [linear; angular],             */
/* force = [linear; angular], SE3 (R,p) maps child coordinates to parent coordinates).     */
/* ======================================================================================= */
typedef struct { double R[9]; double p[3]; } se3_t;
typedef struct { double lin[3]; double ang[3]; } sv_t; /* spatial motion or force */

static void cross3(const double *a, const double *b, double *c) {
  double c0 = a[1] * b[2] - a[2] * b[1];
  double c1 = a[2] * b[0] - a[0] * b[2];
  double c2 = a[0] * b[1] - a[1] * b[0];
  c[0] = c0; c[1] = c1; c[2] = c2;
}
static double dot3(const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static void mat3_vec(const double *R, const double *v, double *o) {
  double o0 = R[0] * v[0] + R[1] * v[1] + R[2] * v[2];
  double o1 = R[3] * v[0] + R[4] * v[1] + R[5] * v[2];
  double o2 = R[6] * v[0] + R[7] * v[1] + R[8] * v[2];
  o[0] = o0; o[1] = o1; o[2] = o2;
}
static void mat3T_vec(const double *R, const double *v, double *o) {
  double o0 = R[0] * v[0] + R[3] * v[1] + R[6] * v[2];
  double o1 = R[1] * v[0] + R[4] * v[1] + R[7] * v[2];
  double o2 = R[2] * v[0] + R[5] * v[1] + R[8] * v[2];
  o[0] = o0; o[1] = o1; o[2] = o2;
}
static void mat3_mul(const double *A, const double *B, double *C) {
  double t[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) t[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
  memcpy(C, t, sizeof t);
}
/* rotation by angle q about unit axis (Rodrigues; JointModelRevoluteUnaligned) */
static void axis_angle(const double *ax, double q, double *R) {
  double s = sin(q), c = cos(q), v = 1.0 - c;
  R[0] = ax[0] * ax[0] * v + c;         R[1] = ax[0] * ax[1] * v - ax[2] * s; R[2] = ax[0] * ax[2] * v + ax[1] * s;
  R[3] = ax[1] * ax[0] * v + ax[2] * s; R[4] = ax[1] * ax[1] * v + c;         R[5] = ax[1] * ax[2] * v - ax[0] * s;
  R[6] = ax[2] * ax[0] * v - ax[1] * s; R[7] = ax[2] * ax[1] * v + ax[0] * s; R[8] = ax[2] * ax[2] * v + c;
}
static void se3_mul(const se3_t *A, const se3_t *B, se3_t *C) {
  se3_t t;
  mat3_mul(A->R, B->R, t.R);
  mat3_vec(A->R, B->p, t.p);
  for (int i = 0; i < 3; ++i) t.p[i] += A->p[i];
  *C = t;
}
static void se3_inv(const se3_t *A, se3_t *C) {
  se3_t t;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) t.R[3 * i + j] = A->R[3 * j + i];
  mat3T_vec(A->R, A->p, t.p);
  for (int i = 0; i < 3; ++i) t.p[i] = -t.p[i];
  *C = t;
}
static void se3_identity(se3_t *A) {
  memset(A, 0, sizeof *A);
  A->R[0] = A->R[4] = A->R[8] = 1.0;
}
/* motion: child -> parent */
static void motion_act(const se3_t *M, const sv_t *m, sv_t *o) {
  sv_t t; double c[3];
  mat3_vec(M->R, m->ang, t.ang);
  mat3_vec(M->R, m->lin, t.lin);
  cross3(M->p, t.ang, c);
  for (int i = 0; i < 3; ++i) t.lin[i] += c[i];
  *o = t;
}
/* motion: parent -> child */
static void motion_actinv(const se3_t *M, const sv_t *m, sv_t *o) {
  sv_t t; double c[3], d[3];
  cross3(M->p, m->ang, c);
  for (int i = 0; i < 3; ++i) d[i] = m->lin[i] - c[i];
  mat3T_vec(M->R, d, t.lin);
  mat3T_vec(M->R, m->ang, t.ang);
  *o = t;
}
/* force: child -> parent */
static void force_act(const se3_t *M, const sv_t *f, sv_t *o) {
  sv_t t; double c[3];
  mat3_vec(M->R, f->lin, t.lin);
  mat3_vec(M->R, f->ang, t.ang);
  cross3(M->p, t.lin, c);
  for (int i = 0; i < 3; ++i) t.ang[i] += c[i];
  *o = t;
}
/* motion x motion */
static void crm(const sv_t *a, const sv_t *b, sv_t *o) {
  sv_t t; double c1[3], c2[3];
  cross3(a->ang, b->lin, c1);
  cross3(a->lin, b->ang, c2);
  for (int i = 0; i < 3; ++i) t.lin[i] = c1[i] + c2[i];
  cross3(a->ang, b->ang, t.ang);
  *o = t;
}
/* motion x* force */
static void crf(const sv_t *a, const sv_t *f, sv_t *o) {
  sv_t t; double c1[3], c2[3];
  cross3(a->ang, f->lin, t.lin);
  cross3(a->ang, f->ang, c1);
  cross3(a->lin, f->lin, c2);
  for (int i = 0; i < 3; ++i) t.ang[i] = c1[i] + c2[i];
  *o = t;
}
/* spatial inertia (mass, com lever c, rotational inertia about the COM) times motion */
static void inertia_mul(double mass, const double *c, const double *I, const sv_t *m, sv_t *o) {
  sv_t t; double cw[3], Iw[3], cf[3];
  cross3(c, m->ang, cw);
  for (int i = 0; i < 3; ++i) t.lin[i] = mass * (m->lin[i] - cw[i]);
  mat3_vec(I, m->ang, Iw);
  cross3(c, t.lin, cf);
  for (int i = 0; i < 3; ++i) t.ang[i] = Iw[i] + cf[i];
  *o = t;
}
static void sv_add(sv_t *a, const sv_t *b) {
  for (int i = 0; i < 3; ++i) { a->lin[i] += b->lin[i]; a->ang[i] += b->ang[i]; }
}
static void sv_zero(sv_t *a) { memset(a, 0, sizeof *a); }

/* liMi(q_i) = jointPlacement_i * Rot(axis_i, q_i) */
static void joint_placement(const aslr_chain_t *c, int i, double q, se3_t *liMi) {
  double Rj[9];
  axis_angle(c->axis[i], q, Rj);
  mat3_mul(c->joint_R[i], Rj, liMi->R);
  memcpy(liMi->p, c->joint_p[i], sizeof liMi->p);
}

/* ======================================================================================= */
/* RNEA, CRBA (via RNEA columns), nle, RNEA derivatives (tangent mode)                      */
/* ======================================================================================= */
typedef struct {
  se3_t liMi[NJ];
  sv_t v[NJ], a[NJ], h[NJ], f[NJ], vJ[NJ];
  sv_t Xa[NJ], Xv[NJ]; /* parent acceleration / velocity expressed in frame i */
} rnea_ws_t;

static void rnea_core(const aslr_chain_t *c, const double *q, const double *v, const double *a,
                      const double *grav, double *tau, rnea_ws_t *w) {
  const int nj = c->nj;
  sv_t vp, ap;
  sv_zero(&vp); sv_zero(&ap);
  for (int i = 0; i < 3; ++i) ap.lin[i] = -grav[i]; /* a_0 = -gravity */
  for (int i = 0; i < nj; ++i) {
    joint_placement(c, i, q[i], &w->liMi[i]);
    sv_zero(&w->vJ[i]);
    for (int k = 0; k < 3; ++k) w->vJ[i].ang[k] = c->axis[i][k] * v[i];
    motion_actinv(&w->liMi[i], &vp, &w->Xv[i]);
    w->v[i] = w->Xv[i];
    sv_add(&w->v[i], &w->vJ[i]);
    motion_actinv(&w->liMi[i], &ap, &w->Xa[i]);
    w->a[i] = w->Xa[i];
    for (int k = 0; k < 3; ++k) w->a[i].ang[k] += c->axis[i][k] * a[i];
    sv_t vxvj;
    crm(&w->v[i], &w->vJ[i], &vxvj);
    sv_add(&w->a[i], &vxvj);
    inertia_mul(c->mass[i], c->com[i], c->inertia[i], &w->v[i], &w->h[i]);
    sv_t Ia, vxh;
    inertia_mul(c->mass[i], c->com[i], c->inertia[i], &w->a[i], &Ia);
    crf(&w->v[i], &w->h[i], &vxh);
    w->f[i] = Ia;
    sv_add(&w->f[i], &vxh);
    vp = w->v[i];
    ap = w->a[i];
  }
  for (int i = nj - 1; i >= 0; --i) {
    tau[i] = dot3(c->axis[i], w->f[i].ang);
    if (i > 0) {
      sv_t fp;
      force_act(&w->liMi[i], &w->f[i], &fp);
      sv_add(&w->f[i - 1], &fp);
    }
  }
}

void aslr_cpu_rnea(const aslr_chain_t *c, const double *q, const double *v, const double *a,
                   double *tau) {
  rnea_ws_t w;
  rnea_core(c, q, v, a, c->gravity, tau, &w);
}

/* joint-space inertia matrix: column j = RNEA(q, 0, e_j) without gravity (M e_j), symmetric
 * like the Python binding's data.M after computeAllTerms. */
void aslr_cpu_crba(const aslr_chain_t *c, const double *q, double *M) {
  const int nj = c->nj;
  double zero[NJ] = {0}, e[NJ], g0[3] = {0, 0, 0}, col[NJ];
  rnea_ws_t w;
  for (int j = 0; j < nj; ++j) {
    memset(e, 0, sizeof e);
    e[j] = 1.0;
    rnea_core(c, q, zero, e, g0, col, &w);
    for (int i = 0; i < nj; ++i) M[i * nj + j] = col[i];
  }
  for (int i = 0; i < nj; ++i)
    for (int j = i + 1; j < nj; ++j) {
      double s = 0.5 * (M[i * nj + j] + M[j * nj + i]);
      M[i * nj + j] = M[j * nj + i] = s;
    }
}

/* data.nle = C(q,v) v + g(q) = RNEA(q, v, 0) */
void aslr_cpu_nle(const aslr_chain_t *c, const double *q, const double *v, double *nle) {
  double zero[NJ] = {0};
  rnea_ws_t w;
  rnea_core(c, q, v, zero, c->gravity, nle, &w);
}

/* pinocchio.computeRNEADerivatives(model, data, q, v, a) -> dtau_dq, dtau_dv (row-major nj x nj),
 * used at free_fwddyn_asr.py:75 / free_fwddyn_vsa.py:79.  Exact forward-mode (tangent)
 * differentiation of the recursion above, one direction per column. */
void aslr_cpu_rnea_derivatives(const aslr_chain_t *c, const double *q, const double *v,
                               const double *a, double *dtau_dq, double *dtau_dv) {
  const int nj = c->nj;
  rnea_ws_t w;
  double tau[NJ];
  rnea_core(c, q, v, a, c->gravity, tau, &w); /* w.f now holds the ACCUMULATED forces */
  for (int kind = 0; kind < 2; ++kind) {
    double *out = kind == 0 ? dtau_dq : dtau_dv;
    for (int j = 0; j < nj; ++j) {
      sv_t dv[NJ], da[NJ], df[NJ];
      sv_t dvp, dap;
      sv_zero(&dvp); sv_zero(&dap);
      sv_t Sj;
      sv_zero(&Sj);
      for (int k = 0; k < 3; ++k) Sj.ang[k] = c->axis[j][k];
      for (int i = 0; i < nj; ++i) {
        sv_t t;
        motion_actinv(&w.liMi[i], &dvp, &dv[i]);
        motion_actinv(&w.liMi[i], &dap, &da[i]);
        sv_t dvJ;
        sv_zero(&dvJ);
        if (i == j) {
          if (kind == 0) {
            crm(&w.Xv[i], &Sj, &t); sv_add(&dv[i], &t);
            crm(&w.Xa[i], &Sj, &t); sv_add(&da[i], &t);
          } else {
            sv_add(&dv[i], &Sj);
            dvJ = Sj;
          }
        }
        crm(&dv[i], &w.vJ[i], &t); sv_add(&da[i], &t);
        crm(&w.v[i], &dvJ, &t);    sv_add(&da[i], &t);
        sv_t Ida, Idv, t1, t2;
        inertia_mul(c->mass[i], c->com[i], c->inertia[i], &da[i], &Ida);
        inertia_mul(c->mass[i], c->com[i], c->inertia[i], &dv[i], &Idv);
        crf(&dv[i], &w.h[i], &t1);
        crf(&w.v[i], &Idv, &t2);
        df[i] = Ida; sv_add(&df[i], &t1); sv_add(&df[i], &t2);
        dvp = dv[i];
        dap = da[i];
      }
      for (int i = nj - 1; i >= 0; --i) {
        out[i * nj + j] = dot3(c->axis[i], df[i].ang);
        if (i > 0) {
          sv_t t;
          force_act(&w.liMi[i], &df[i], &t);
          sv_add(&df[i - 1], &t);
          if (kind == 0 && i == j) {
            sv_t SxF;
            crf(&Sj, &w.f[i], &SxF);
            force_act(&w.liMi[i], &SxF, &t);
            sv_add(&df[i - 1], &t);
          }
        }
      }
    }
  }
}

/* ======================================================================================= */
/* kinematics: frame placement, LOCAL frame Jacobian                                        */
/* ======================================================================================= */
static void forward_kinematics(const aslr_chain_t *c, const double *q, se3_t *oMi) {
  se3_t cur, li;
  se3_identity(&cur);
  for (int i = 0; i < c->nj; ++i) {
    joint_placement(c, i, q[i], &li);
    se3_mul(&cur, &li, &cur);
    oMi[i] = cur;
  }
}

void aslr_cpu_frame_placement(const aslr_chain_t *c, const double *q, int joint, const double *fR,
                              const double *fp, double *oR, double *op) {
  se3_t oMi[NJ], F, oMf;
  forward_kinematics(c, q, oMi);
  memcpy(F.R, fR, sizeof F.R);
  memcpy(F.p, fp, sizeof F.p);
  se3_mul(&oMi[joint], &F, &oMf);
  memcpy(oR, oMf.R, sizeof oMf.R);
  memcpy(op, oMf.p, sizeof oMf.p);
}

/* pinocchio.getFrameJacobian(..., LOCAL) (residual_frame_placement.py:20-21):
 * column j = (oMf^-1 oMj).act(S_j) for the joints supporting the frame, 0 otherwise. */
void aslr_cpu_frame_jacobian(const aslr_chain_t *c, const double *q, int joint, const double *fR,
                             const double *fp, double *J) {
  const int nj = c->nj;
  se3_t oMi[NJ], F, oMf, fMo;
  forward_kinematics(c, q, oMi);
  memcpy(F.R, fR, sizeof F.R);
  memcpy(F.p, fp, sizeof F.p);
  se3_mul(&oMi[joint], &F, &oMf);
  se3_inv(&oMf, &fMo);
  memset(J, 0, sizeof(double) * 6 * nj);
  for (int j = 0; j <= joint; ++j) {
    se3_t fMj;
    se3_mul(&fMo, &oMi[j], &fMj);
    sv_t S, col;
    sv_zero(&S);
    for (int k = 0; k < 3; ++k) S.ang[k] = c->axis[j][k];
    motion_act(&fMj, &S, &col);
    for (int k = 0; k < 3; ++k) {
      J[k * nj + j] = col.lin[k];
      J[(3 + k) * nj + j] = col.ang[k];
    }
  }
}

/* ======================================================================================= */
/* SE(3) / SO(3) log maps and their Jacobians (Pinocchio 2.6 explog.hpp, published formulas) */
/* ======================================================================================= */
#define TAYLOR_PREC 1.220703125e-04 /* TaylorSeriesExpansion<double>::precision<3>() = eps^(1/4) = 2^-13 */

static double log3(const double *R, double *w) {
  /* theta from the trace, nominal antisymmetric-part formula, dedicated branch near pi */
  const double PI = 3.14159265358979323846;
  double tr = R[0] + R[4] + R[8], theta;
  if (tr >= 3.0) { tr = 3.0; theta = 0.0; }
  else if (tr <= -1.0) { tr = -1.0; theta = PI; }
  else theta = acos((tr - 1.0) / 2.0);
  if (theta >= PI - 1e-2) {
    const double cphi = -(tr - 1.0) / 2.0;
    const double beta = theta * theta / (1.0 + cphi);
    double tmp[3] = {(R[0] + cphi) * beta, (R[4] + cphi) * beta, (R[8] + cphi) * beta};
    w[0] = (R[7] > R[5] ? 1.0 : -1.0) * (tmp[0] > 0.0 ? sqrt(tmp[0]) : 0.0);
    w[1] = (R[2] > R[6] ? 1.0 : -1.0) * (tmp[1] > 0.0 ? sqrt(tmp[1]) : 0.0);
    w[2] = (R[3] > R[1] ? 1.0 : -1.0) * (tmp[2] > 0.0 ? sqrt(tmp[2]) : 0.0);
  } else {
    const double t = ((theta > TAYLOR_PREC) ? theta / sin(theta) : 1.0) / 2.0;
    w[0] = t * (R[7] - R[5]);
    w[1] = t * (R[2] - R[6]);
    w[2] = t * (R[3] - R[1]);
  }
  return theta;
}

static void exp3(const double *w, double *R) {
  double t2 = dot3(w, w), t = sqrt(t2);
  double a, b; /* R = I + a [w]x + b [w]x^2 */
  if (t < 1e-8) { a = 1.0 - t2 / 6.0; b = 0.5 - t2 / 24.0; }
  else { a = sin(t) / t; b = (1.0 - cos(t)) / t2; }
  double W[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0}, W2[9];
  mat3_mul(W, W, W2);
  for (int i = 0; i < 9; ++i) R[i] = a * W[i] + b * W2[i];
  R[0] += 1.0; R[4] += 1.0; R[8] += 1.0;
}

/* pinocchio.log(M).vector (residual_frame_placement.py:14-15): [v; w] */
void aslr_cpu_log6(const double *R, const double *p, double *r6) {
  double w[3];
  const double t = log3(R, w);
  const double t2 = t * t;
  double alpha, beta;
  if (t < TAYLOR_PREC) {
    alpha = 1.0 - t2 / 12.0 - t2 * t2 / 720.0;
    beta = 1.0 / 12.0 + t2 / 720.0;
  } else {
    const double st = sin(t), ct = cos(t);
    alpha = t * st / (2.0 * (1.0 - ct));
    beta = 1.0 / t2 - st / (2.0 * t * (1.0 - ct));
  }
  double wxp[3];
  cross3(w, p, wxp);
  const double wp = dot3(w, p);
  for (int i = 0; i < 3; ++i) {
    r6[i] = alpha * p[i] - 0.5 * wxp[i] + beta * wp * w[i];
    r6[3 + i] = w[i];
  }
}

void aslr_cpu_exp6(const double *r6, double *R, double *p) {
  const double *v = r6, *w = r6 + 3;
  double t2 = dot3(w, w), t = sqrt(t2);
  exp3(w, R);
  /* p = V v, V = I + b [w]x + c [w]x^2 */
  double b, cc;
  if (t < 1e-8) { b = 0.5 - t2 / 24.0; cc = 1.0 / 6.0 - t2 / 120.0; }
  else { b = (1.0 - cos(t)) / t2; cc = (t - sin(t)) / (t2 * t); }
  double wxv[3], wxwxv[3];
  cross3(w, v, wxv);
  cross3(w, wxv, wxwxv);
  for (int i = 0; i < 3; ++i) p[i] = v[i] + b * wxv[i] + cc * wxwxv[i];
}

static void jlog3(double theta, const double *w, double *J) {
  const double t2 = theta * theta;
  double alpha, diag;
  if (theta < TAYLOR_PREC) {
    alpha = 1.0 / 12.0 + t2 / 720.0;
    diag = 0.5 * (2.0 - t2 / 6.0);
  } else {
    const double st = sin(theta), ct = cos(theta);
    const double st_1mct = st / (1.0 - ct);
    alpha = 1.0 / t2 - st_1mct / (2.0 * theta);
    diag = 0.5 * (theta * st_1mct);
  }
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) J[3 * i + j] = alpha * w[i] * w[j];
  J[0] += diag; J[4] += diag; J[8] += diag;
  /* addSkew(0.5 w) */
  J[1] -= 0.5 * w[2]; J[2] += 0.5 * w[1];
  J[3] += 0.5 * w[2]; J[5] -= 0.5 * w[0];
  J[6] -= 0.5 * w[1]; J[7] += 0.5 * w[0];
}

/* pinocchio.Jlog6(M) (residual_frame_placement.py:19): 6x6 row-major, blocks [[A,B],[0,A]] */
void aslr_cpu_jlog6(const double *R, const double *p, double *J) {
  double w[3];
  const double t = log3(R, w);
  const double t2 = t * t;
  double beta, beta_dot_over_theta;
  if (t < TAYLOR_PREC) {
    beta = 1.0 / 12.0 + t2 / 720.0;
    beta_dot_over_theta = 1.0 / 360.0;
  } else {
    const double tinv = 1.0 / t, t2inv = tinv * tinv;
    const double st = sin(t), ct = cos(t);
    const double inv_2_2ct = 1.0 / (2.0 * (1.0 - ct));
    beta = t2inv - st * tinv * inv_2_2ct;
    beta_dot_over_theta = -2.0 * t2inv * t2inv + (1.0 + st * tinv) * t2inv * inv_2_2ct;
  }
  double A[9], C[9], Bm[9];
  jlog3(t, w, A);
  const double wTp = dot3(w, p);
  double v3[3];
  for (int i = 0; i < 3; ++i) v3[i] = (beta_dot_over_theta * wTp) * w[i] - (t2 * beta_dot_over_theta + 2.0 * beta) * p[i];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) C[3 * i + j] = v3[i] * w[j] + beta * w[i] * p[j];
  C[0] += wTp * beta; C[4] += wTp * beta; C[8] += wTp * beta;
  /* addSkew(0.5 p) */
  C[1] -= 0.5 * p[2]; C[2] += 0.5 * p[1];
  C[3] += 0.5 * p[2]; C[5] -= 0.5 * p[0];
  C[6] -= 0.5 * p[1]; C[7] += 0.5 * p[0];
  mat3_mul(C, A, Bm);
  memset(J, 0, sizeof(double) * 36);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      J[6 * i + j] = A[3 * i + j];
      J[6 * i + 3 + j] = Bm[3 * i + j];
      J[6 * (3 + i) + 3 + j] = A[3 * i + j];
    }
}

/* ======================================================================================= */
/* small dense helpers                                                                      */
/* ======================================================================================= */
/* Cholesky LL^T of an n x n SPD matrix (row-major, in place, lower).  Returns 0 on success,
 * 1 if a pivot is not strictly positive (Eigen::LLT info() == NumericalIssue). */
static int chol(int n, double *A) {
  for (int j = 0; j < n; ++j) {
    double d = A[j * n + j];
    for (int k = 0; k < j; ++k) d -= A[j * n + k] * A[j * n + k];
    if (!(d > 0.0)) return 1;
    d = sqrt(d);
    A[j * n + j] = d;
    for (int i = j + 1; i < n; ++i) {
      double s = A[i * n + j];
      for (int k = 0; k < j; ++k) s -= A[i * n + k] * A[j * n + k];
      A[i * n + j] = s / d;
    }
  }
  return 0;
}
static void chol_solve(int n, const double *L, double *b) { /* in place, one rhs */
  for (int i = 0; i < n; ++i) {
    double s = b[i];
    for (int k = 0; k < i; ++k) s -= L[i * n + k] * b[k];
    b[i] = s / L[i * n + i];
  }
  for (int i = n - 1; i >= 0; --i) {
    double s = b[i];
    for (int k = i + 1; k < n; ++k) s -= L[k * n + i] * b[k];
    b[i] = s / L[i * n + i];
  }
}
/* inverse of an SPD matrix via Cholesky (np.linalg.inv(data.M), free_fwddyn_asr.py:40) */
static int spd_inverse(int n, const double *A, double *Ainv) {
  double L[NU * NU], e[NU];
  memcpy(L, A, sizeof(double) * n * n);
  if (chol(n, L)) return 1;
  for (int j = 0; j < n; ++j) {
    memset(e, 0, sizeof e);
    e[j] = 1.0;
    chol_solve(n, L, e);
    for (int i = 0; i < n; ++i) Ainv[i * n + j] = e[i];
  }
  return 0;
}
/* general inverse by Gauss-Jordan with partial pivoting (np.linalg.inv(self.B)) */
static int gen_inverse(int n, const double *A, double *Ainv) {
  double a[NJ * NJ * 2];
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) { a[i * 2 * n + j] = A[i * n + j]; a[i * 2 * n + n + j] = (i == j); }
  for (int col = 0; col < n; ++col) {
    int piv = col;
    for (int r = col + 1; r < n; ++r) if (fabs(a[r * 2 * n + col]) > fabs(a[piv * 2 * n + col])) piv = r;
    if (a[piv * 2 * n + col] == 0.0) return 1;
    if (piv != col) for (int j = 0; j < 2 * n; ++j) { double t = a[col * 2 * n + j]; a[col * 2 * n + j] = a[piv * 2 * n + j]; a[piv * 2 * n + j] = t; }
    double d = a[col * 2 * n + col];
    for (int j = 0; j < 2 * n; ++j) a[col * 2 * n + j] /= d;
    for (int r = 0; r < n; ++r) if (r != col) {
      double f = a[r * 2 * n + col];
      if (f != 0.0) for (int j = 0; j < 2 * n; ++j) a[r * 2 * n + j] -= f * a[col * 2 * n + j];
    }
  }
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) Ainv[i * n + j] = a[i * 2 * n + n + j];
  return 0;
}

/* ======================================================================================= */
/* Differential action model: calc + calcDiff + cost stack                                  */
/* ======================================================================================= */
static __thread double *g_resid_sink; /* set by aslr_cpu_dam_residuals around one aslr_cpu_dam call */

void aslr_cpu_dam(const aslr_chain_t *c, const aslr_model_t *m, const double *frame_ref,
                  const double *x, const double *u_in, double *xout_o, double *cost_o,
                  double *Fx_o, double *Fu_o, double *Lx_o, double *Lu_o, double *Lxx_o,
                  double *Lxu_o, double *Luu_o) {
  const int nj = c->nj, nx = 4 * nj, nu = m->nu, nv = 2 * nj;
  double u_none[NU];
  const double *u = u_in;
  if (u == NULL) { /* free_fwddyn_vsa.py:21-23 / free_fwddyn_asr.py:21-22 */
    for (int i = 0; i < nu; ++i) u_none[i] = 0.0;
    if (m->dam == ASLR_DAM_VSA) for (int i = nu / 2; i < nu; ++i) u_none[i] = 3.0;
    u = u_none;
  }
  /* slices (free_fwddyn_asr.py:27-30) */
  const double *q_l = x, *q_m = x + nj, *v_l = x + 2 * nj;
  double dq[NJ];
  for (int i = 0; i < nj; ++i) dq[i] = q_l[i] - q_m[i];

  /* stiffness, coupling torque, motor torque */
  double K[NJ * NJ], tau_m[NJ], tau_couple[NJ];
  if (m->dam == ASLR_DAM_VSA) { /* K = diag(u[nu/2:]) (free_fwddyn_vsa.py:34); tau_m = u[:nu/2] */
    memset(K, 0, sizeof K);
    for (int i = 0; i < nj; ++i) { K[i * nj + i] = u[nj + i]; tau_m[i] = u[i]; }
  } else { /* self.K ; tau[nv/2:] = S u (actuation_asr.py:10, __init__.py:268) */
    memcpy(K, m->K, sizeof(double) * nj * nj);
    for (int i = 0; i < nj; ++i) {
      double s = 0.0;
      for (int j = 0; j < nu; ++j) s += m->S[i * nu + j] * u[j];
      tau_m[i] = s;
    }
  }
  for (int i = 0; i < nj; ++i) {
    double s = 0.0;
    for (int j = 0; j < nj; ++j) s += K[i * nj + j] * dq[j];
    tau_couple[i] = s; /* free_fwddyn_asr.py:35 */
  }
  /* computeAllTerms -> M, nle ; Minv, Binv (free_fwddyn_asr.py:38-41) */
  double M[NJ * NJ], Minv[NJ * NJ], Binv[NJ * NJ], nle[NJ];
  aslr_cpu_crba(c, q_l, M);
  aslr_cpu_nle(c, q_l, v_l, nle);
  spd_inverse(nj, M, Minv);
  gen_inverse(nj, m->B, Binv);
  /* xout (free_fwddyn_asr.py:43,46; free_fwddyn_vsa.py:44,47); tau[:nv/2] = 0 for every
   * actuation model of the package */
  double xout[2 * NJ];
  for (int i = 0; i < nj; ++i) {
    double s = 0.0, s2 = 0.0;
    for (int j = 0; j < nj; ++j) {
      s += Minv[i * nj + j] * (-nle[j] - tau_couple[j]);
      s2 += Binv[i * nj + j] * (tau_m[j] + tau_couple[j]);
    }
    xout[i] = s;
    xout[nj + i] = s2;
  }
  if (xout_o) memcpy(xout_o, xout, sizeof(double) * nv);

  /* ---- calcDiff: dynamics (free_fwddyn_asr.py:75-89 / free_fwddyn_vsa.py:79-92) ---- */
  double Fx[2 * NJ * NX], Fu[2 * NJ * NU];
  memset(Fx, 0, sizeof Fx);
  memset(Fu, 0, sizeof Fu);
  double dtau_dq[NJ * NJ], dtau_dv[NJ * NJ];
  aslr_cpu_rnea_derivatives(c, q_l, v_l, xout, dtau_dq, dtau_dv);
  for (int i = 0; i < nj; ++i)
    for (int j = 0; j < nj; ++j) {
      double s_q = 0.0, s_k = 0.0, s_v = 0.0, b_k = 0.0;
      for (int k = 0; k < nj; ++k) {
        s_q += Minv[i * nj + k] * (-dtau_dq[k * nj + j] - K[k * nj + j]);
        s_k += Minv[i * nj + k] * K[k * nj + j];
        s_v += Minv[i * nj + k] * (-dtau_dv[k * nj + j]);
        b_k += Binv[i * nj + k] * K[k * nj + j];
      }
      Fx[i * nx + j] = s_q;               /* ddq_dq */
      Fx[i * nx + nj + j] = s_k;          /* Minv K */
      Fx[i * nx + 2 * nj + j] = s_v;      /* ddq_dv */
      Fx[(nj + i) * nx + j] = b_k;        /* Binv K */
      Fx[(nj + i) * nx + nj + j] = -b_k;  /* -Binv K */
    }
  if (m->dam == ASLR_DAM_VSA) {
    for (int i = 0; i < nj; ++i)
      for (int j = 0; j < nj; ++j) {
        Fu[i * nu + nj + j] = Minv[i * nj + j] * (-q_l[j] + q_m[j]);      /* vsa:89 */
        Fu[(nj + i) * nu + nj + j] = Binv[i * nj + j] * (q_l[j] - q_m[j]); /* vsa:90 */
        Fu[(nj + i) * nu + j] = Binv[i * nj + j];                          /* vsa:92 */
      }
  } else if (nu > 1) { /* asr:88-89 */
    for (int i = 0; i < nj; ++i)
      for (int j = 0; j < nu; ++j) {
        double s = 0.0;
        for (int k = 0; k < nj; ++k) s += Binv[i * nj + k] * m->S[k * nu + j];
        Fu[(nj + i) * nu + j] = s;
      }
  }
  if (Fx_o) memcpy(Fx_o, Fx, sizeof(double) * nv * nx);
  if (Fu_o) memcpy(Fu_o, Fu, sizeof(double) * nv * nu);

  /* ---- cost stack: CostModelSum of residual costs ---- */
  double *rs = g_resid_sink; /* aslr_cpu_dam_residuals: data.r, the stacked residual vectors (integrated_action.py:17-18) */
  double cost = 0.0, Lx[NX], Lu[NU], Lxx[NX * NX], Lxu[NX * NU], Luu[NU * NU];
  memset(Lx, 0, sizeof Lx); memset(Lu, 0, sizeof Lu);
  memset(Lxx, 0, sizeof Lxx); memset(Lxu, 0, sizeof Lxu); memset(Luu, 0, sizeof Luu);
  for (int ci = 0; ci < m->ncosts; ++ci) {
    const aslr_cost_t *ct = &m->costs[ci];
    const double w = ct->weight;
    switch (ct->type) {
    case ASLR_COST_FRAME_PLACEMENT: { /* residual_frame_placement.py:13-24 */
      se3_t Mref, oMf, refinv, rMf;
      const double *ref = frame_ref ? frame_ref : ct->ref;
      memcpy(Mref.R, ref, sizeof Mref.R);
      memcpy(Mref.p, ref + 9, sizeof Mref.p);
      aslr_cpu_frame_placement(c, q_l, ct->frame_joint, ct->frame_R, ct->frame_p, oMf.R, oMf.p);
      se3_inv(&Mref, &refinv);
      se3_mul(&refinv, &oMf, &rMf);
      double r[6], Jl[36], fJf[6 * NJ], Jr[6 * NJ];
      aslr_cpu_log6(rMf.R, rMf.p, r);
      aslr_cpu_jlog6(rMf.R, rMf.p, Jl);
      aslr_cpu_frame_jacobian(c, q_l, ct->frame_joint, ct->frame_R, ct->frame_p, fJf);
      for (int i = 0; i < 6; ++i)
        for (int j = 0; j < nj; ++j) {
          double s = 0.0;
          for (int k = 0; k < 6; ++k) s += Jl[6 * i + k] * fJf[k * nj + j];
          Jr[i * nj + j] = s; /* Rx[:, :nq_l] */
        }
      double a = 0.0;
      for (int i = 0; i < 6; ++i) a += ct->act_w[i] * r[i] * r[i];
      cost += w * 0.5 * a;
      if (rs) { memcpy(rs, r, sizeof r); rs += 6; }
      for (int j = 0; j < nj; ++j) {
        double s = 0.0;
        for (int i = 0; i < 6; ++i) s += Jr[i * nj + j] * ct->act_w[i] * r[i];
        Lx[j] += w * s;
        for (int k = 0; k < nj; ++k) {
          double h = 0.0;
          for (int i = 0; i < 6; ++i) h += Jr[i * nj + j] * ct->act_w[i] * Jr[i * nj + k];
          Lxx[j * nx + k] += w * h;
        }
      }
    } break;
    case ASLR_COST_STATE: { /* r = state.diff(xref, x), Rx = I */
      double a = 0.0;
      for (int i = 0; i < nx; ++i) {
        const double r = x[i] - ct->ref[i];
        a += ct->act_w[i] * r * r;
        Lx[i] += w * ct->act_w[i] * r;
        Lxx[i * nx + i] += w * ct->act_w[i];
        if (rs) rs[i] = r;
      }
      if (rs) rs += nx;
      cost += w * 0.5 * a;
    } break;
    case ASLR_COST_CONTROL: { /* r = u - uref, Ru = I */
      double a = 0.0;
      for (int i = 0; i < nu; ++i) {
        const double r = u[i] - ct->ref[i];
        a += ct->act_w[i] * r * r;
        Lu[i] += w * ct->act_w[i] * r;
        Luu[i * nu + i] += w * ct->act_w[i];
        if (rs) rs[i] = r;
      }
      if (rs) rs += nu;
      cost += w * 0.5 * a;
    } break;
    case ASLR_COST_PENDULUM: { /* __init__.py:228-249 */
      const double c1 = cos(x[0]), c2 = cos(x[1]), s1 = sin(x[0]), s2 = sin(x[1]);
      const double r[6] = {s1, s2, 1.0 + c1, 1.0 + c2, x[4], x[5]};
      const double *aw = ct->act_w;
      double a = 0.0;
      for (int i = 0; i < 6; ++i) a += aw[i] * r[i] * r[i];
      cost += w * 0.5 * a;
      if (rs) { memcpy(rs, r, sizeof r); rs += 6; }
      /* Lx = Rx^T Ar */
      Lx[0] += w * (c1 * aw[0] * r[0] - s1 * aw[2] * r[2]);
      Lx[1] += w * (c2 * aw[1] * r[1] - s2 * aw[3] * r[3]);
      Lx[4] += w * aw[4] * r[4];
      Lx[5] += w * aw[5] * r[5];
      /* Lxx = diag(Rxx^T diag(Arr)) */
      Lxx[0 * nx + 0] += w * ((c1 * c1 - s1 * s1) * aw[0] + (s1 * s1 + (1.0 - c1) * c1) * aw[2]);
      Lxx[1 * nx + 1] += w * ((c2 * c2 - s2 * s2) * aw[1] + (s2 * s2 + (1.0 - c2) * c2) * aw[3]);
      Lxx[4 * nx + 4] += w * aw[4];
      Lxx[5 * nx + 5] += w * aw[5];
    } break;
    case ASLR_COST_STIFFNESS: { /* stiffness_cost.py:13-20: cost = sum(lamda (K - Kref)) */
      const int h = nu / 2;
      double a = 0.0;
      for (int i = 0; i < h; ++i) {
        a += ct->lambda * (u[h + i] - ct->ref[i]);
        Lu[h + i] += w * ct->lambda;
        if (rs) rs[i] = ct->lambda * (u[h + i] - ct->ref[i]); /* stiffness_cost.py:15 */
      }
      if (rs) rs += h;
      cost += w * a;
    } break;
    default: break;
    }
  }
  if (cost_o) *cost_o = cost;
  if (Lx_o) memcpy(Lx_o, Lx, sizeof(double) * nx);
  if (Lu_o) memcpy(Lu_o, Lu, sizeof(double) * nu);
  if (Lxx_o) memcpy(Lxx_o, Lxx, sizeof(double) * nx * nx);
  if (Lxu_o) memcpy(Lxu_o, Lxu, sizeof(double) * nx * nu);
  if (Luu_o) memcpy(Luu_o, Luu, sizeof(double) * nu * nu);
}

void aslr_cpu_dam_residuals(const aslr_chain_t *c, const aslr_model_t *m, const double *frame_ref,
                            const double *x, const double *u, double *r) {
  g_resid_sink = r;
  aslr_cpu_dam(c, m, frame_ref, x, u, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL);
  g_resid_sink = NULL;
}

static int rec_len(int nx, int nu) {
  int n = 2 * nx * nx + 2 * nx * nu + nu * nu + nx + nu;
  return (n + 15) / 16 * 16;
}

/* IntegratedActionModelEulerASR.calc / calcDiff (integrated_action.py:13-42) */
void aslr_cpu_knot(const aslr_chain_t *c, const aslr_model_t *m, const double *frame_ref,
                   const double *x, const double *u, double *xnext, double *cost, double *rec) {
  const int nj = c->nj, nx = 4 * nj, nu = m->nu, nv = 2 * nj;
  const double dt = m->dt;
  double xout[2 * NJ], cst, da_dx[2 * NJ * NX], da_du[2 * NJ * NU];
  double Lx[NX], Lu[NU], Lxx[NX * NX], Lxu[NX * NU], Luu[NU * NU];
  aslr_cpu_dam(c, m, frame_ref, x, u, xout, &cst, da_dx, da_du, Lx, Lu, Lxx, Lxu, Luu);
  if (cost) *cost = cst; /* NOT scaled by dt (integrated_action.py:19) */
  if (xnext) {
    /* dx = [x[nq:] dt + acc dt^2 ; acc dt] ; xnext = integrate(x, dx) (integrated_action.py:23-24) */
    for (int i = 0; i < nv; ++i) {
      const double dxq = x[nv + i] * dt + xout[i] * dt * dt;
      const double dxv = xout[i] * dt;
      xnext[i] = x[i] + dxq;
      xnext[nv + i] = x[nv + i] + dxv;
    }
  }
  if (!rec) return;
  double *Fx = rec, *Fu = Fx + nx * nx, *rLxx = Fu + nx * nu, *rLxu = rLxx + nx * nx,
         *rLuu = rLxu + nx * nu, *rLx = rLuu + nu * nu, *rLu = rLx + nx;
  memset(rec, 0, sizeof(double) * rec_len(nx, nu));
  /* ddx_dx = vstack[da_dx dt, da_dx]; ddx_dx[i, nv+i] += 1; Fx = I + dt ddx_dx (integrated_action.py:31-35) */
  for (int i = 0; i < nv; ++i)
    for (int j = 0; j < nx; ++j) {
      double top = da_dx[i * nx + j] * dt;
      if (j == nv + i) top += 1.0;
      Fx[i * nx + j] = dt * top;
      Fx[(nv + i) * nx + j] = dt * da_dx[i * nx + j];
    }
  for (int i = 0; i < nx; ++i) Fx[i * nx + i] += 1.0;
  /* Fu = dt vstack[da_du dt, da_du] (integrated_action.py:36-37) */
  for (int i = 0; i < nv; ++i)
    for (int j = 0; j < nu; ++j) {
      Fu[i * nu + j] = dt * (da_du[i * nu + j] * dt);
      Fu[(nv + i) * nu + j] = dt * da_du[i * nu + j];
    }
  memcpy(rLxx, Lxx, sizeof(double) * nx * nx);
  memcpy(rLxu, Lxu, sizeof(double) * nx * nu);
  memcpy(rLuu, Luu, sizeof(double) * nu * nu);
  memcpy(rLx, Lx, sizeof(double) * nx);
  memcpy(rLu, Lu, sizeof(double) * nu);
}

/* ======================================================================================= */
/* batched calc / calcDiff                                                                  */
/* ======================================================================================= */
static int desc_check(const aslr_problem_desc_t *d) {
  if (!d || d->B <= 0 || d->T <= 0 || d->nmodels <= 0 || d->nmodels > ASLR_MAX_MODELS) return ASLR_E_INVALID;
  if (d->chain.nj <= 0 || d->chain.nj > NJ) return ASLR_E_INVALID;
  return ASLR_OK;
}

int aslr_cpu_calc_diff(const aslr_problem_desc_t *d, const double *xs, const double *us,
                       double *xnext, double *cost, double *deriv) {
  if (desc_check(d)) return ASLR_E_INVALID;
  const int B = d->B, T = d->T, nx = 4 * d->chain.nj;
  const int nu = d->models[d->node_model[0]].nu, rl = rec_len(nx, nu);
  for (int t = 0; t <= T; ++t)
    for (int b = 0; b < B; ++b) {
      const aslr_model_t *m = &d->models[d->node_model[t]];
      const size_t tb = (size_t)t * B + b;
      const double *fr = d->frame_ref ? d->frame_ref + 12 * (size_t)b : NULL;
      aslr_cpu_knot(&d->chain, m, fr, xs + tb * nx, t < T ? us + tb * nu : NULL,
                    xnext ? xnext + tb * nx : NULL, cost ? cost + tb : NULL,
                    deriv ? deriv + tb * rl : NULL);
    }
  return ASLR_OK;
}

int aslr_cpu_calc(const aslr_problem_desc_t *d, const double *xs, const double *us, double *xnext,
                  double *cost) {
  return aslr_cpu_calc_diff(d, xs, us, xnext, cost, NULL);
}

/* ======================================================================================= */
/* BoxQP (Tassa et al. 2014 projected Newton)                                */
/* ======================================================================================= */
/* diagnostics (serial runs only): number of calls, projected-Newton iterations, line-search trials */
/* thread-local: OpenMP workers keep their own (the getters report the calling thread, i.e. serial runs) */
static __thread long long g_qp_calls, g_qp_iters, g_qp_trials, g_qp_maxed;
static __thread long long g_qp_hist_it[16], g_qp_hist_tr[32], g_qp_hist_c0[2][16];
void aslr_cpu_boxqp_hist_c0(long long *o32, int reset) {
  for (int i = 0; i < 32; ++i) o32[i] = g_qp_hist_c0[i / 16][i % 16];
  if (reset) memset(g_qp_hist_c0, 0, sizeof g_qp_hist_c0);
}
/* optional per-call log (serial runs): -2 = a trajectory's solve starts, -1 = a backward sweep starts, else the
 * number of projected-Newton iterations of one BoxQP call (calls arrive knot T-1 ... 0) */
static __thread int *g_qp_log; static __thread long g_qp_log_n, g_qp_log_cap;
void aslr_cpu_boxqp_log(int *buf, long cap) { g_qp_log = buf; g_qp_log_cap = cap; g_qp_log_n = 0; }
long aslr_cpu_boxqp_log_len(void) { return g_qp_log_n; }
/* optional dump of whole QP instances (diagnostics): per call [H (n^2) | q | lb | ub | x0 | x* | iterations] */
static __thread double *g_qp_dump; static __thread long g_qp_dump_n, g_qp_dump_cap;
void aslr_cpu_boxqp_dump(double *buf, long cap) { g_qp_dump = buf; g_qp_dump_cap = cap; g_qp_dump_n = 0; }
long aslr_cpu_boxqp_dump_len(void) { return g_qp_dump_n; }
static void qp_log(int v) { if (g_qp_log && g_qp_log_n < g_qp_log_cap) g_qp_log[g_qp_log_n++] = v; }
void aslr_cpu_boxqp_hist(long long *it16, long long *tr32, int reset) {
  for (int i = 0; i < 16; ++i) it16[i] = g_qp_hist_it[i];
  for (int i = 0; i < 32; ++i) tr32[i] = g_qp_hist_tr[i];
  if (reset) { memset(g_qp_hist_it, 0, sizeof g_qp_hist_it); memset(g_qp_hist_tr, 0, sizeof g_qp_hist_tr); }
}
void aslr_cpu_boxqp_stats(long long *out4, int reset) {
  out4[0] = g_qp_calls; out4[1] = g_qp_iters; out4[2] = g_qp_trials; out4[3] = g_qp_maxed;
  if (reset) g_qp_calls = g_qp_iters = g_qp_trials = g_qp_maxed = 0;
}

int aslr_cpu_boxqp(int n, const double *H, const double *q, const double *lb, const double *ub,
                   double *x, int maxiter, double th_acceptstep, double th_grad, double reg,
                   double *Hff_inv, int32_t *free_idx, int32_t *nf_o, int32_t *clamped_idx,
                   int32_t *nc_o) {
  double g[NU], dx[NU], xnew[NU], Hff[NU * NU], L[NU * NU];
  int nf = 0, nc = 0, k;
  for (int i = 0; i < n; ++i) x[i] = fmax(fmin(x[i], ub[i]), lb[i]);
  g_qp_calls++;
  long long tr0 = g_qp_trials;
  int nc0 = 0;
  for (k = 0; k < maxiter; ++k) {
    g_qp_iters++;
    nf = nc = 0;
    for (int i = 0; i < n; ++i) {
      double s = q[i];
      for (int j = 0; j < n; ++j) s += H[i * n + j] * x[j];
      g[i] = s;
    }
    for (int j = 0; j < n; ++j) {
      if ((x[j] == lb[j] && g[j] > 0.0) || (x[j] == ub[j] && g[j] < 0.0)) clamped_idx[nc++] = j;
      else free_idx[nf++] = j;
    }
    if (k == 0) nc0 = nc > 0;
    double gnorm = 0.0;
    for (int i = 0; i < nf; ++i) gnorm = fmax(gnorm, fabs(g[free_idx[i]]));
    for (int i = 0; i < nf; ++i)
      for (int j = 0; j < nf; ++j) Hff[i * nf + j] = H[free_idx[i] * n + free_idx[j]];
    if (gnorm <= th_grad || nf == 0) {
      if (nf > 0) { /* the caller needs Hff^-1 of the FINAL free set.  Crocoddyl recomputes it only
                     * when k == 0 and otherwise reuses the last iteration's, which is the same
                     * matrix whenever the free set did not change in the last step; recomputing
                     * always is identical in that case and well defined otherwise. */
        memcpy(L, Hff, sizeof(double) * nf * nf);
        for (int i = 0; i < nf; ++i) L[i * nf + i] += reg;
        if (spd_inverse(nf, L, Hff_inv)) { *nf_o = nf; *nc_o = nc; return -1; }
      }
      *nf_o = nf; *nc_o = nc;
      g_qp_hist_it[k < 15 ? k : 15]++;
      g_qp_hist_c0[nc0][k < 15 ? k : 15]++;
      { long long tr = g_qp_trials - tr0; g_qp_hist_tr[tr < 0 ? 0 : (tr < 31 ? tr : 31)]++; }
      return k;
    }
    for (int i = 0; i < nf; ++i) Hff[i * nf + i] += reg;
    memcpy(L, Hff, sizeof(double) * nf * nf);
    if (chol(nf, L)) { *nf_o = nf; *nc_o = nc; return -1; }
    spd_inverse(nf, Hff, Hff_inv);
    /* dx_f = -Hff^-1 (q_f + H_fc x_c) - x_f */
    double rhs[NU];
    for (int i = 0; i < nf; ++i) {
      double s = -q[free_idx[i]];
      for (int j = 0; j < nc; ++j) s -= H[free_idx[i] * n + clamped_idx[j]] * x[clamped_idx[j]];
      rhs[i] = s;
    }
    chol_solve(nf, L, rhs);
    memset(dx, 0, sizeof dx);
    for (int i = 0; i < nf; ++i) dx[free_idx[i]] = rhs[i] - x[free_idx[i]];
    /* line search on the clamped step */
    double fold = 0.0;
    for (int i = 0; i < n; ++i) {
      double s = 0.0;
      for (int j = 0; j < n; ++j) s += H[i * n + j] * x[j];
      fold += 0.5 * x[i] * s + q[i] * x[i];
    }
    double alpha = 1.0;
    int accepted = 0;
    for (int a = 0; a < ASLR_NALPHA; ++a, alpha *= 0.5) {
      g_qp_trials++;
      for (int i = 0; i < n; ++i) xnew[i] = fmax(fmin(x[i] + alpha * dx[i], ub[i]), lb[i]);
      double fnew = 0.0, gd = 0.0;
      for (int i = 0; i < n; ++i) {
        double s = 0.0;
        for (int j = 0; j < n; ++j) s += H[i * n + j] * xnew[j];
        fnew += 0.5 * xnew[i] * s + q[i] * xnew[i];
        gd += g[i] * (x[i] - xnew[i]);
      }
      if (fold - fnew > th_acceptstep * gd) {
        memcpy(x, xnew, sizeof(double) * n);
        accepted = 1;
        break;
      }
    }
    if (!accepted) {
      /* x is unchanged: every remaining iteration of Crocoddyl's loop would recompute the same gradient,
       * active set and rejected steps and finally return this x with this Hff^-1; return it now. */
      g_qp_maxed++;
      *nf_o = nf; *nc_o = nc;
      return k + 1;
    }
  }
  *nf_o = nf; *nc_o = nc;
  g_qp_maxed++;
  return k;
}

/* ======================================================================================= */
/* single-trajectory solver (SolverDDP / SolverFDDP / SolverBoxDDP)   */
/* ======================================================================================= */
typedef struct {
  int T, nx, nu, rl;
  const aslr_problem_desc_t *d;
  const aslr_solver_params_t *sp;
  const double *frame_ref; /* this trajectory's, or NULL */
  const double *x0;
  double *xs, *us, *xs_try, *us_try, *xnext, *cost_node, *rec, *fs;
  double *K, *k, *Qu, *Quuk, *Vx, *Vxx;
  double cost, cost_try, xreg, ureg, d1, d2, dg, dq, dv, dV, dVexp, stop, steplength;
  int is_feasible, was_feasible, iter, status;
  double *log; /* this trajectory's column of the per-iteration log (layout of aslr_set_iteration_log), or NULL */
  int log_cap, log_B, accepted;
} traj_t;

/* what crocoddyl's callbacks read at the end of an iteration (ASLR_LOG_* of include/aslr_to_amd.h) */
static void traj_log(const traj_t *s) {
  if (!s->log || s->iter >= s->log_cap) return;
  double *lg = s->log + (size_t)s->iter * ASLR_LOG_COUNT * s->log_B;
  const size_t B = s->log_B;
  lg[ASLR_LOG_COST * B] = s->cost; lg[ASLR_LOG_STOP * B] = s->stop; lg[ASLR_LOG_XREG * B] = s->xreg;
  lg[ASLR_LOG_STEP * B] = s->steplength; lg[ASLR_LOG_D1 * B] = s->d1; lg[ASLR_LOG_D2 * B] = s->d2;
  lg[ASLR_LOG_DV * B] = s->dV; lg[ASLR_LOG_DVEXP * B] = s->dVexp; lg[ASLR_LOG_ACCEPTED * B] = s->accepted;
  lg[ASLR_LOG_STATUS * B] = s->status; lg[ASLR_LOG_FEASIBLE * B] = s->is_feasible;
}

static int is_bad(double v) { return isnan(v) || isinf(v) || v >= 1e30; } /* crocoddyl raiseIfNaN */

static double *dalloc(size_t n) { return (double *)calloc(n ? n : 1, sizeof(double)); }

static void traj_alloc(traj_t *s, const aslr_problem_desc_t *d, const aslr_solver_params_t *sp) {
  memset(s, 0, sizeof *s);
  s->d = d; s->sp = sp;
  s->T = d->T; s->nx = 4 * d->chain.nj; s->nu = d->models[d->node_model[0]].nu;
  s->rl = rec_len(s->nx, s->nu);
  const size_t T1 = s->T + 1, nx = s->nx, nu = s->nu;
  s->xs = dalloc(T1 * nx); s->us = dalloc(T1 * nu);
  s->xs_try = dalloc(T1 * nx); s->us_try = dalloc(T1 * nu);
  s->xnext = dalloc(T1 * nx); s->cost_node = dalloc(T1);
  s->rec = dalloc(T1 * s->rl); s->fs = dalloc(T1 * nx);
  s->K = dalloc(T1 * nu * nx); s->k = dalloc(T1 * nu); s->Qu = dalloc(T1 * nu); s->Quuk = dalloc(T1 * nu);
  s->Vx = dalloc(T1 * nx); s->Vxx = dalloc(T1 * nx * nx);
}
static void traj_free(traj_t *s) {
  free(s->xs); free(s->us); free(s->xs_try); free(s->us_try); free(s->xnext); free(s->cost_node);
  free(s->rec); free(s->fs); free(s->K); free(s->k); free(s->Qu); free(s->Quuk); free(s->Vx); free(s->Vxx);
}
static const aslr_model_t *node_model(const traj_t *s, int t) { return &s->d->models[s->d->node_model[t]]; }

/* problem.calc + problem.calcDiff on (xs, us): cost_ = sum of node costs; gaps (B.2) */
static void traj_calc_diff(traj_t *s) {
  const int T = s->T, nx = s->nx, nu = s->nu;
  double cost = 0.0;
  for (int t = 0; t <= T; ++t) {
    aslr_cpu_knot(&s->d->chain, node_model(s, t), s->frame_ref, s->xs + t * nx,
                  t < T ? s->us + t * nu : NULL, s->xnext + t * nx, s->cost_node + t, s->rec + (size_t)t * s->rl);
    cost += s->cost_node[t];
  }
  s->cost = cost;
  if (!s->is_feasible) {
    int could = 1;
    double m0 = 0.0;
    for (int i = 0; i < nx; ++i) { s->fs[i] = s->x0[i] - s->xs[i]; m0 = fmax(m0, fabs(s->fs[i])); }
    if (m0 >= s->sp->th_gaptol) could = 0;
    for (int t = 0; t < T; ++t) {
      double mt = 0.0;
      for (int i = 0; i < nx; ++i) {
        s->fs[(t + 1) * nx + i] = s->xnext[t * nx + i] - s->xs[(t + 1) * nx + i];
        mt = fmax(mt, fabs(s->fs[(t + 1) * nx + i]));
      }
      if (mt >= s->sp->th_gaptol) could = 0;
    }
    s->is_feasible = could;
  } else if (!s->was_feasible) {
    memset(s->fs, 0, sizeof(double) * (T + 1) * nx);
  }
}

/* SolverDDP::backwardPass + computeGains (B.1), SolverBoxDDP::computeGains (B.5).
 * Returns 1 on "backward_error". */
static int traj_backward(traj_t *s) {
  qp_log(-1);
  const int T = s->T, nx = s->nx, nu = s->nu;
  const aslr_solver_params_t *sp = s->sp;
  double *VxxT = s->Vxx + (size_t)T * nx * nx, *VxT = s->Vx + T * nx;
  {
    const double *rec = s->rec + (size_t)T * s->rl;
    const double *Lxx = rec + nx * nx + nx * nu, *Lx = Lxx + nx * nx + nx * nu + nu * nu;
    memcpy(VxxT, Lxx, sizeof(double) * nx * nx);
    memcpy(VxT, Lx, sizeof(double) * nx);
    if (!isnan(s->xreg)) for (int i = 0; i < nx; ++i) VxxT[i * nx + i] += s->xreg;
    if (!s->is_feasible)
      for (int i = 0; i < nx; ++i) {
        double a = 0.0;
        for (int j = 0; j < nx; ++j) a += VxxT[i * nx + j] * s->fs[T * nx + j];
        VxT[i] += a;
      }
  }
  for (int t = T - 1; t >= 0; --t) {
    const double *rec = s->rec + (size_t)t * s->rl;
    const double *Fx = rec, *Fu = Fx + nx * nx, *Lxx = Fu + nx * nu, *Lxu = Lxx + nx * nx,
                 *Luu = Lxu + nx * nu, *Lx = Luu + nu * nu, *Lu = Lx + nx;
    const double *Vxx_p = s->Vxx + (size_t)(t + 1) * nx * nx, *Vx_p = s->Vx + (t + 1) * nx;
    double Qxx[NX * NX], Qxu[NX * NU], Quu[NU * NU], Qx[NX], FxTVxx[NX * NX], FuTVxx[NU * NX];
    double *Qu = s->Qu + t * nu, *K = s->K + (size_t)t * nu * nx, *kk = s->k + t * nu, *Quuk = s->Quuk + t * nu;
    for (int i = 0; i < nx; ++i)
      for (int j = 0; j < nx; ++j) {
        double a = 0.0;
        for (int l = 0; l < nx; ++l) a += Fx[l * nx + i] * Vxx_p[l * nx + j];
        FxTVxx[i * nx + j] = a;
      }
    for (int i = 0; i < nx; ++i) {
      for (int j = 0; j < nx; ++j) {
        double a = 0.0;
        for (int l = 0; l < nx; ++l) a += FxTVxx[i * nx + l] * Fx[l * nx + j];
        Qxx[i * nx + j] = Lxx[i * nx + j] + a;
      }
      double a = 0.0;
      for (int l = 0; l < nx; ++l) a += Fx[l * nx + i] * Vx_p[l];
      Qx[i] = Lx[i] + a;
    }
    for (int i = 0; i < nu; ++i)
      for (int j = 0; j < nx; ++j) {
        double a = 0.0;
        for (int l = 0; l < nx; ++l) a += Fu[l * nu + i] * Vxx_p[l * nx + j];
        FuTVxx[i * nx + j] = a;
      }
    for (int i = 0; i < nx; ++i)
      for (int j = 0; j < nu; ++j) {
        double a = 0.0;
        for (int l = 0; l < nx; ++l) a += FxTVxx[i * nx + l] * Fu[l * nu + j];
        Qxu[i * nu + j] = Lxu[i * nu + j] + a;
      }
    for (int i = 0; i < nu; ++i) {
      for (int j = 0; j < nu; ++j) {
        double a = 0.0;
        for (int l = 0; l < nx; ++l) a += FuTVxx[i * nx + l] * Fu[l * nu + j];
        Quu[i * nu + j] = Luu[i * nu + j] + a;
      }
      double a = 0.0;
      for (int l = 0; l < nx; ++l) a += Fu[l * nu + i] * Vx_p[l];
      Qu[i] = Lu[i] + a;
      if (!isnan(s->ureg)) Quu[i * nu + i] += s->ureg;
    }
    /* computeGains */
    const aslr_model_t *m = node_model(s, t);
    if (sp->solver == ASLR_SOLVER_BOXDDP && m->has_u_limits && s->is_feasible) {
      double lb[NU], ub[NU], xq[NU], Hff_inv[NU * NU], Quu_inv[NU * NU];
      int32_t fidx[NU], cidx[NU], nf, nc;
      for (int i = 0; i < nu; ++i) {
        lb[i] = m->u_lb[i] - s->us[t * nu + i];
        ub[i] = m->u_ub[i] - s->us[t * nu + i];
        xq[i] = kk[i]; /* warm start at the stored k (sign as stored) */
      }
      double x0_dump[NU];
      memcpy(x0_dump, xq, sizeof(double) * nu);
      int r = aslr_cpu_boxqp(nu, Quu, Qu, lb, ub, xq, sp->boxqp_maxiter, sp->boxqp_th_acceptstep,
                             sp->boxqp_th_grad, sp->boxqp_reg, Hff_inv, fidx, &nf, cidx, &nc);
      if (r < 0) return 1;
      if (g_qp_dump && g_qp_dump_n + nu * nu + 5 * nu + 1 <= g_qp_dump_cap) {
        double *o = g_qp_dump + g_qp_dump_n;
        memcpy(o, Quu, sizeof(double) * nu * nu); o += nu * nu;
        memcpy(o, Qu, sizeof(double) * nu); o += nu;
        memcpy(o, lb, sizeof(double) * nu); o += nu;
        memcpy(o, ub, sizeof(double) * nu); o += nu;
        memcpy(o, x0_dump, sizeof(double) * nu); o += nu;
        memcpy(o, xq, sizeof(double) * nu); o += nu;
        *o = (double)r;
        g_qp_dump_n += nu * nu + 5 * nu + 1;
      }
      qp_log(r + 10 * (nc > 0));
      memset(Quu_inv, 0, sizeof Quu_inv);
      for (int i = 0; i < nf; ++i)
        for (int j = 0; j < nf; ++j) Quu_inv[fidx[i] * nu + fidx[j]] = Hff_inv[i * nf + j];
      for (int i = 0; i < nu; ++i)
        for (int j = 0; j < nx; ++j) {
          double a = 0.0;
          for (int l = 0; l < nu; ++l) a += Quu_inv[i * nu + l] * Qxu[j * nu + l];
          K[i * nx + j] = a;
        }
      for (int i = 0; i < nu; ++i) kk[i] = -xq[i];
      for (int i = 0; i < nc; ++i) Qu[cidx[i]] = 0.0;
    } else {
      double L[NU * NU], col[NU];
      memcpy(L, Quu, sizeof(double) * nu * nu);
      if (chol(nu, L)) return 1;
      for (int j = 0; j < nx; ++j) {
        for (int i = 0; i < nu; ++i) col[i] = Qxu[j * nu + i];
        chol_solve(nu, L, col);
        for (int i = 0; i < nu; ++i) K[i * nx + j] = col[i];
      }
      for (int i = 0; i < nu; ++i) kk[i] = Qu[i];
      chol_solve(nu, L, kk);
    }
    /* value function */
    double *Vx = s->Vx + t * nx, *Vxx = s->Vxx + (size_t)t * nx * nx;
    for (int i = 0; i < nu; ++i) {
      double a = 0.0;
      for (int j = 0; j < nu; ++j) a += Quu[i * nu + j] * kk[j];
      Quuk[i] = a;
    }
    for (int i = 0; i < nx; ++i) {
      double a = 0.0, b2 = 0.0;
      for (int l = 0; l < nu; ++l) { a += K[l * nx + i] * Quuk[l]; b2 += K[l * nx + i] * Qu[l]; }
      Vx[i] = Qx[i] + a - 2.0 * b2;
      for (int j = 0; j < nx; ++j) {
        double c2 = 0.0;
        for (int l = 0; l < nu; ++l) c2 += Qxu[i * nu + l] * K[l * nx + j];
        Vxx[i * nx + j] = Qxx[i * nx + j] - c2;
      }
    }
    for (int i = 0; i < nx; ++i)
      for (int j = i + 1; j < nx; ++j) {
        const double a = 0.5 * (Vxx[i * nx + j] + Vxx[j * nx + i]);
        Vxx[i * nx + j] = Vxx[j * nx + i] = a;
      }
    if (!isnan(s->xreg)) for (int i = 0; i < nx; ++i) Vxx[i * nx + i] += s->xreg;
    if (!s->is_feasible)
      for (int i = 0; i < nx; ++i) {
        double a = 0.0;
        for (int j = 0; j < nx; ++j) a += Vxx[i * nx + j] * s->fs[t * nx + j];
        Vx[i] += a;
      }
    double mv = 0.0, mvv = 0.0;
    int nan = 0;
    for (int i = 0; i < nx; ++i) { if (isnan(Vx[i])) nan = 1; mv = fmax(mv, fabs(Vx[i])); }
    for (int i = 0; i < nx * nx; ++i) { if (isnan(Vxx[i])) nan = 1; mvv = fmax(mvv, fabs(Vxx[i])); }
    if (nan || is_bad(mv) || is_bad(mvv)) return 1;
  }
  return 0;
}

/* expectedImprovement (DDP/BoxDDP) and updateExpectedImprovement (FDDP; B.4) + stop */
static void traj_expected_improvement(traj_t *s) {
  const int T = s->T, nx = s->nx, nu = s->nu;
  double d1 = 0.0, d2 = 0.0, stop = 0.0;
  for (int t = 0; t < T; ++t)
    for (int i = 0; i < nu; ++i) {
      d1 += s->Qu[t * nu + i] * s->k[t * nu + i];
      d2 -= s->k[t * nu + i] * s->Quuk[t * nu + i];
      stop += s->Qu[t * nu + i] * s->Qu[t * nu + i];
    }
  s->stop = stop;
  if (s->sp->solver == ASLR_SOLVER_FDDP) {
    double dg = d1, dq = d2;
    if (!s->is_feasible)
      for (int t = 0; t <= T; ++t) {
        const double *Vxx = s->Vxx + (size_t)t * nx * nx, *f = s->fs + t * nx, *Vx = s->Vx + t * nx;
        for (int i = 0; i < nx; ++i) {
          double a = 0.0;
          for (int j = 0; j < nx; ++j) a += Vxx[i * nx + j] * f[j];
          dg -= Vx[i] * f[i];
          dq += f[i] * a;
        }
      }
    s->dg = dg; s->dq = dq;
  } else {
    s->d1 = d1; s->d2 = d2;
  }
}

/* forwardPass(alpha): DDP / BoxDDP (B.2, B.5) and FDDP (B.4).  Returns 1 on "forward_error". */
static int traj_forward(traj_t *s, double alpha) {
  const int T = s->T, nx = s->nx, nu = s->nu;
  const aslr_solver_params_t *sp = s->sp;
  const int fddp_gaps = sp->solver == ASLR_SOLVER_FDDP && !(s->is_feasible || alpha == 1.0);
  double xnext[NX], cost_try = 0.0;
  memcpy(xnext, s->x0, sizeof(double) * nx);
  for (int t = 0; t <= T; ++t) {
    double *xt = s->xs_try + t * nx;
    if (fddp_gaps) for (int i = 0; i < nx; ++i) xt[i] = xnext[i] + s->fs[t * nx + i] * (alpha - 1.0);
    else memcpy(xt, xnext, sizeof(double) * nx);
    if (t == T) break;
    const aslr_model_t *m = node_model(s, t);
    double *ut = s->us_try + t * nu;
    const double *K = s->K + (size_t)t * nu * nx;
    for (int i = 0; i < nu; ++i) {
      double a = s->us[t * nu + i] - s->k[t * nu + i] * alpha;
      for (int j = 0; j < nx; ++j) a -= K[i * nx + j] * (xt[j] - s->xs[t * nx + j]);
      ut[i] = a;
    }
    if (sp->solver == ASLR_SOLVER_BOXDDP && m->has_u_limits)
      for (int i = 0; i < nu; ++i) ut[i] = fmin(fmax(ut[i], m->u_lb[i]), m->u_ub[i]);
    double c;
    aslr_cpu_knot(&s->d->chain, m, s->frame_ref, xt, ut, xnext, &c, NULL);
    cost_try += c;
    double mx = 0.0;
    int nan = 0;
    for (int i = 0; i < nx; ++i) { if (isnan(xnext[i])) nan = 1; mx = fmax(mx, fabs(xnext[i])); }
    if (is_bad(cost_try) || nan || is_bad(mx)) return 1;
  }
  double c;
  aslr_cpu_knot(&s->d->chain, node_model(s, T), s->frame_ref, s->xs_try + T * nx, NULL, NULL, &c, NULL);
  cost_try += c;
  if (is_bad(cost_try)) return 1;
  s->cost_try = cost_try;
  return 0;
}

/* FDDP expectedImprovement after a trial (B.4) */
static void traj_fddp_dv(traj_t *s) {
  const int T = s->T, nx = s->nx;
  double dv = 0.0;
  if (!s->is_feasible)
    for (int t = 0; t <= T; ++t) {
      const double *Vxx = s->Vxx + (size_t)t * nx * nx, *f = s->fs + t * nx;
      for (int i = 0; i < nx; ++i) {
        double a = 0.0;
        for (int j = 0; j < nx; ++j) a += Vxx[i * nx + j] * (s->xs[t * nx + j] - s->xs_try[t * nx + j]);
        dv -= f[i] * a;
      }
    }
  s->dv = dv;
  s->d1 = s->dg + dv;
  s->d2 = s->dq - 2.0 * dv;
}

static void reg_increase(traj_t *s) {
  s->xreg *= s->sp->reg_incfactor;
  if (s->xreg > s->sp->reg_max) s->xreg = s->sp->reg_max;
  s->ureg = s->xreg;
}
static void reg_decrease(traj_t *s) {
  s->xreg /= s->sp->reg_decfactor;
  if (s->xreg < s->sp->reg_min) s->xreg = s->sp->reg_min;
  s->ureg = s->xreg;
}

/* SolverDDP::solve / SolverFDDP::solve (B.2, B.4).  Returns 1 when converged. */
static int traj_solve(traj_t *s) {
  qp_log(-2);
  const aslr_solver_params_t *sp = s->sp;
  const int T = s->T, nx = s->nx, nu = s->nu;
  s->xreg = s->ureg = isnan(sp->reg_init) ? sp->reg_min : sp->reg_init;
  s->is_feasible = sp->is_feasible;
  s->was_feasible = 0;
  s->status = 0;
  int recalc = 1;
  for (s->iter = 0; s->iter < sp->maxiter; ++s->iter) {
    for (;;) {
      if (recalc) traj_calc_diff(s);
      if (traj_backward(s)) {
        s->status |= ASLR_ST_BACKWARD_ERR;
        recalc = 0;
        { /* a regularisation that cannot grow (zero, NaN, factor <= 1) would retry for ever, in Crocoddyl too; it counts
           * as the ceiling here and in the kernels (aslr_backward.inc.hpp), so that both always terminate */
          const double before = s->xreg;
          reg_increase(s);
          if (!(s->xreg > before)) s->xreg = s->ureg = sp->reg_max;
        }
        if (s->xreg == sp->reg_max) { s->status |= ASLR_ST_REG_MAX; return 0; }
        continue;
      }
      break;
    }
    traj_expected_improvement(s);
    recalc = 0;
    double alpha = 1.0;
    s->accepted = -1;
    for (int a = 0; a < ASLR_NALPHA; ++a, alpha *= 0.5) {
      s->steplength = alpha;
      if (traj_forward(s, alpha)) { s->status |= ASLR_ST_FORWARD_ERR; continue; }
      s->dV = s->cost - s->cost_try;
      int accept = 0;
      if (sp->solver == ASLR_SOLVER_FDDP) {
        traj_fddp_dv(s);
        s->dVexp = alpha * (s->d1 + 0.5 * alpha * s->d2);
        if (s->dVexp >= 0.0) {
          if (s->d1 < sp->th_grad || s->dV > sp->th_acceptstep * s->dVexp) accept = 1;
        } else if (!s->is_feasible && s->dV > sp->th_acceptnegstep * s->dVexp) accept = 1;
      } else {
        s->dVexp = alpha * (s->d1 + 0.5 * alpha * s->d2);
        if (s->dVexp >= 0.0)
          if (s->d1 < sp->th_grad || !s->is_feasible || s->dV > sp->th_acceptstep * s->dVexp) accept = 1;
      }
      if (accept) {
        s->was_feasible = s->is_feasible;
        memcpy(s->xs, s->xs_try, sizeof(double) * (T + 1) * nx);
        memcpy(s->us, s->us_try, sizeof(double) * T * nu);
        s->is_feasible = sp->solver == ASLR_SOLVER_FDDP ? (s->was_feasible || alpha == 1.0) : 1;
        s->cost = s->cost_try;
        s->accepted = a;
        recalc = 1;
        break;
      }
    }
    if (s->steplength > sp->th_stepdec) reg_decrease(s);
    if (s->steplength <= sp->th_stepinc) {
      reg_increase(s);
      if (s->xreg == sp->reg_max) { s->status |= ASLR_ST_REG_MAX; traj_log(s); s->iter++; return 0; }
    }
    traj_log(s); /* Crocoddyl calls the callbacks here, before the convergence test */
    if (!sp->fixed_iterations && s->was_feasible && s->stop < sp->th_stop) {
      s->status |= ASLR_ST_CONVERGED;
      s->iter++;
      return 1;
    }
  }
  return 0;
}

/* gather / scatter one trajectory between the time-major batch buffers and the solver */
static void traj_load(traj_t *s, const double *xs, const double *us, int B, int b) {
  const int T = s->T, nx = s->nx, nu = s->nu;
  for (int t = 0; t <= T; ++t) memcpy(s->xs + t * nx, xs + ((size_t)t * B + b) * nx, sizeof(double) * nx);
  for (int t = 0; t < T; ++t) memcpy(s->us + t * nu, us + ((size_t)t * B + b) * nu, sizeof(double) * nu);
  s->x0 = s->d->x0 + (size_t)b * nx;
  s->frame_ref = s->d->frame_ref ? s->d->frame_ref + 12 * (size_t)b : NULL;
}
static void traj_store(const traj_t *s, double *xs, double *us, int B, int b) {
  const int T = s->T, nx = s->nx, nu = s->nu;
  for (int t = 0; t <= T; ++t) memcpy(xs + ((size_t)t * B + b) * nx, s->xs + t * nx, sizeof(double) * nx);
  for (int t = 0; t < T; ++t) memcpy(us + ((size_t)t * B + b) * nu, s->us + t * nu, sizeof(double) * nu);
}

int aslr_cpu_solve(const aslr_problem_desc_t *d, const aslr_solver_params_t *sp, double *xs,
                   double *us, double *traj_f, int32_t *traj_i, int32_t nthreads) {
  return aslr_cpu_solve_log(d, sp, xs, us, traj_f, traj_i, nthreads, NULL, 0);
}

int aslr_cpu_solve_log(const aslr_problem_desc_t *d, const aslr_solver_params_t *sp, double *xs,
                       double *us, double *traj_f, int32_t *traj_i, int32_t nthreads, double *log, int32_t log_cap) {
  if (desc_check(d)) return ASLR_E_INVALID;
  const int B = d->B;
#ifdef _OPENMP
#pragma omp parallel num_threads(nthreads > 1 ? nthreads : 1)
#endif
  {
    traj_t s;
    traj_alloc(&s, d, sp);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
    for (int b = 0; b < B; ++b) {
      traj_load(&s, xs, us, B, b);
      s.log = log ? log + b : NULL; s.log_cap = log_cap; s.log_B = B;
      traj_solve(&s);
      traj_store(&s, xs, us, B, b);
      if (traj_f) {
        traj_f[ASLR_TF_COST * B + b] = s.cost;
        traj_f[ASLR_TF_STOP * B + b] = s.stop;
        traj_f[ASLR_TF_XREG * B + b] = s.xreg;
        traj_f[ASLR_TF_D1 * B + b] = s.d1;
        traj_f[ASLR_TF_D2 * B + b] = s.d2;
        traj_f[ASLR_TF_STEP * B + b] = s.steplength;
        traj_f[ASLR_TF_DV * B + b] = s.dV;
        traj_f[ASLR_TF_DVEXP * B + b] = s.dVexp;
      }
      if (traj_i) {
        traj_i[ASLR_TI_ITER * B + b] = s.iter;
        traj_i[ASLR_TI_STATUS * B + b] = s.status;
        traj_i[ASLR_TI_FEASIBLE * B + b] = s.is_feasible;
        traj_i[ASLR_TI_WAS_FEASIBLE * B + b] = s.was_feasible;
      }
    }
    traj_free(&s);
  }
  return ASLR_OK;
}

int aslr_cpu_backward_pass(const aslr_problem_desc_t *d, const aslr_solver_params_t *sp,
                           const double *deriv, const double *gaps, const double *us,
                           const double *xreg, const int32_t *feasible, double *kgain, double *kff,
                           double *qu, double *vx, double *vxx, double *d1, double *d2,
                           double *stop, int32_t *fail) {
  if (desc_check(d)) return ASLR_E_INVALID;
  const int B = d->B;
  traj_t s;
  traj_alloc(&s, d, sp);
  const int T = s.T, nx = s.nx, nu = s.nu, rl = s.rl;
  for (int b = 0; b < B; ++b) {
    for (int t = 0; t <= T; ++t) {
      const size_t tb = (size_t)t * B + b;
      memcpy(s.rec + (size_t)t * rl, deriv + tb * rl, sizeof(double) * rl);
      memcpy(s.fs + t * nx, gaps + tb * nx, sizeof(double) * nx);
      if (t < T) {
        memcpy(s.us + t * nu, us + tb * nu, sizeof(double) * nu);
        memcpy(s.k + t * nu, kff + tb * nu, sizeof(double) * nu);
      }
    }
    s.xreg = s.ureg = xreg[b];
    s.is_feasible = feasible[b];
    fail[b] = traj_backward(&s);
    traj_expected_improvement(&s);
    d1[b] = sp->solver == ASLR_SOLVER_FDDP ? s.dg : s.d1;
    d2[b] = sp->solver == ASLR_SOLVER_FDDP ? s.dq : s.d2;
    stop[b] = s.stop;
    for (int t = 0; t <= T; ++t) {
      const size_t tb = (size_t)t * B + b;
      memcpy(vx + tb * nx, s.Vx + t * nx, sizeof(double) * nx);
      memcpy(vxx + tb * nx * nx, s.Vxx + (size_t)t * nx * nx, sizeof(double) * nx * nx);
      if (t < T) {
        memcpy(kgain + tb * nu * nx, s.K + (size_t)t * nu * nx, sizeof(double) * nu * nx);
        memcpy(kff + tb * nu, s.k + t * nu, sizeof(double) * nu);
        memcpy(qu + tb * nu, s.Qu + t * nu, sizeof(double) * nu);
      }
    }
  }
  traj_free(&s);
  return ASLR_OK;
}

int aslr_cpu_forward_pass(const aslr_problem_desc_t *d, const aslr_solver_params_t *sp,
                          double alpha, const double *xs, const double *us, const double *kgain,
                          const double *kff, const double *gaps, const int32_t *feasible,
                          double *xs_try, double *us_try, double *cost_try, int32_t *fail) {
  if (desc_check(d)) return ASLR_E_INVALID;
  const int B = d->B;
  traj_t s;
  traj_alloc(&s, d, sp);
  const int T = s.T, nx = s.nx, nu = s.nu;
  for (int b = 0; b < B; ++b) {
    traj_load(&s, xs, us, B, b);
    for (int t = 0; t <= T; ++t) {
      const size_t tb = (size_t)t * B + b;
      if (gaps) memcpy(s.fs + t * nx, gaps + tb * nx, sizeof(double) * nx);
      if (t < T) {
        memcpy(s.K + (size_t)t * nu * nx, kgain + tb * nu * nx, sizeof(double) * nu * nx);
        memcpy(s.k + t * nu, kff + tb * nu, sizeof(double) * nu);
      }
    }
    s.is_feasible = feasible ? feasible[b] : 1;
    fail[b] = traj_forward(&s, alpha);
    cost_try[b] = s.cost_try;
    for (int t = 0; t <= T; ++t) memcpy(xs_try + ((size_t)t * B + b) * nx, s.xs_try + t * nx, sizeof(double) * nx);
    for (int t = 0; t < T; ++t) memcpy(us_try + ((size_t)t * B + b) * nu, s.us_try + t * nu, sizeof(double) * nu);
  }
  traj_free(&s);
  return ASLR_OK;
}

/* ActionModelAbstract::quasiStatic: Gauss-Newton on u with pinv(Fu) */
/* x = pinv(F) b for the SYMMETRIC positive semi-definite normal matrix A = F^T F (n <= NU) and g = F^T b:
 * cyclic Jacobi eigen-decomposition A = V diag(lam) V^T (8 sweeps), x = V diag(lam_i > thr ? 1 / lam_i : 0) V^T g
 * with thr = eps * max(rows, n) * lam_max.  This is Crocoddyl's pseudoInverse (JacobiSVD of F with its singular
 * values cut at eps * max(rows, n) * sigma_max) restated on the normal equations: exact zero columns of F (the
 * VSA stiffness columns at q_l = q_m) give exact zero eigenvalues and are dropped, like there. */
static void pinv_normal_solve(int n, int rows, const double *A_in, const double *g, double *x) {
  double A[NU * NU], V[NU * NU];
  memcpy(A, A_in, sizeof(double) * n * n);
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) V[i * n + j] = i == j ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 8; ++sweep)
    for (int p = 0; p < n - 1; ++p)
      for (int q = p + 1; q < n; ++q) {
        const double apq = A[p * n + q];
        if (apq == 0.0) continue;
        const double theta = (A[q * n + q] - A[p * n + p]) / (2.0 * apq);
        const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
        for (int k = 0; k < n; ++k) { /* columns p, q */
          const double akp = A[k * n + p], akq = A[k * n + q];
          A[k * n + p] = c * akp - sn * akq;
          A[k * n + q] = sn * akp + c * akq;
        }
        for (int k = 0; k < n; ++k) { /* rows p, q */
          const double apk = A[p * n + k], aqk = A[q * n + k];
          A[p * n + k] = c * apk - sn * aqk;
          A[q * n + k] = sn * apk + c * aqk;
        }
        for (int k = 0; k < n; ++k) {
          const double vkp = V[k * n + p], vkq = V[k * n + q];
          V[k * n + p] = c * vkp - sn * vkq;
          V[k * n + q] = sn * vkp + c * vkq;
        }
      }
  double lmax = 0.0;
  for (int i = 0; i < n; ++i) lmax = fmax(lmax, fabs(A[i * n + i]));
  const double thr = 2.220446049250313e-16 * (double)(rows > n ? rows : n) * lmax;
  double y[NU];
  for (int i = 0; i < n; ++i) {
    double a = 0.0;
    for (int k = 0; k < n; ++k) a += V[k * n + i] * g[k];
    y[i] = A[i * n + i] > thr ? a / A[i * n + i] : 0.0;
  }
  for (int k = 0; k < n; ++k) {
    double a = 0.0;
    for (int i = 0; i < n; ++i) a += V[k * n + i] * y[i];
    x[k] = a;
  }
}

int aslr_cpu_quasi_static(const aslr_problem_desc_t *d, int mi, const double *frame_ref,
                          const double *x, double *u, int maxiter, double tol) {
  const aslr_model_t *m = &d->models[mi];
  const int nx = 4 * d->chain.nj, nu = m->nu, rl = rec_len(nx, nu);
  double *rec = dalloc(rl), xnext[NX], cost;
  memset(u, 0, sizeof(double) * nu);
  int it;
  for (it = 0; it < maxiter; ++it) {
    aslr_cpu_knot(&d->chain, m, frame_ref, x, u, xnext, &cost, rec);
    const double *Fu = rec + nx * nx;
    /* du = -pinv(Fu) dx through the normal equations and a thresholded eigen-decomposition (rank-deficient Fu too) */
    double A[NU * NU], rhs[NU];
    for (int i = 0; i < nu; ++i) {
      for (int j = 0; j < nu; ++j) {
        double a = 0.0;
        for (int l = 0; l < nx; ++l) a += Fu[l * nu + i] * Fu[l * nu + j];
        A[i * nu + j] = a;
      }
      double a = 0.0;
      for (int l = 0; l < nx; ++l) a += Fu[l * nu + i] * (xnext[l] - x[l]);
      rhs[i] = -a;
    }
    { double g[NU]; memcpy(g, rhs, sizeof(double) * nu); pinv_normal_solve(nu, nx, A, g, rhs); }
    double nrm = 0.0;
    for (int i = 0; i < nu; ++i) { u[i] += rhs[i]; nrm += rhs[i] * rhs[i]; }
    if (sqrt(nrm) <= tol) break;
  }
  free(rec);
  return it;
}
