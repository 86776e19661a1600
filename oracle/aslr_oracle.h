/*
 * aslr_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, float64) of the aslr_to soft-actuator DDP hot path.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the product
 * package aslr_to_amd never imports, links or calls anything under oracle/.
 *
 * PARITY STATUS: the reference's arithmetic lives in Crocoddyl / Pinocchio / example-robot-data,
 * none of which is vendored in the reference, pinned by it, or installable here (no network), and
 * the reference's tests hold no golden vectors.  So:
 *   - the L2 model code (calc / calcDiff / Euler / state / actuation / frame residual) follows the
 *     reference's own Python files line by line (cited per function);
 *   - rigid-body dynamics, SE(3) log maps, the cost stack and the DDP / FDDP / BoxDDP / BoxQP
 *     solvers restate the published algorithms of Pinocchio 2.6.x and Crocoddyl 1.9-2.0 (the
 *     generation fingerprinted in SURVEY.md 8(c)); they are pinned by the substitute oracles of
 *     tests/test_oracle_*.py (sympy Lagrangian dynamics, finite differences -- the reference's own
 *     test technique, unittest/test_vsa_freefwddyn.py:23-39 --, time-varying LQR, brute-force QP);
 *   - SOLVER PARITY WITH CROCODDYL ITSELF IS "parity unpinned"; model constants of
 *     asr_twodof / talos_arm / double_pendulum are synthetic (robots.py) and likewise unpinned.
 *
 * All buffers use the layouts of include/aslr_to_amd.h (time-major [T+1][B][...]).
 */
#ifndef ASLR_ORACLE_H
#define ASLR_ORACLE_H

#include "../include/aslr_to_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- rigid-body pieces (stand-ins for Pinocchio) ------------------------ */
void aslr_cpu_rnea(const aslr_chain_t *c, const double *q, const double *v, const double *a,
                   double *tau);
void aslr_cpu_crba(const aslr_chain_t *c, const double *q, double *M /* nj*nj */);
void aslr_cpu_nle(const aslr_chain_t *c, const double *q, const double *v, double *nle);
void aslr_cpu_rnea_derivatives(const aslr_chain_t *c, const double *q, const double *v,
                               const double *a, double *dtau_dq, double *dtau_dv);
/* oMf of a frame attached to `joint` with local placement (fR, fp); out = R[9], p[3] */
void aslr_cpu_frame_placement(const aslr_chain_t *c, const double *q, int joint, const double *fR,
                              const double *fp, double *oR, double *op);
/* LOCAL frame Jacobian, 6 x nj row-major, rows = [linear; angular] */
void aslr_cpu_frame_jacobian(const aslr_chain_t *c, const double *q, int joint, const double *fR,
                             const double *fp, double *J);
void aslr_cpu_log6(const double *R, const double *p, double *r6);
void aslr_cpu_exp6(const double *r6, double *R, double *p);
void aslr_cpu_jlog6(const double *R, const double *p, double *J /* 6x6 */);

/* ---- differential + integrated action model (one knot) --------------------------------- */
/* DifferentialFree{ASR,VSA}FwdDynamicsModel.calc + calcDiff on one (x,u); frame_ref may be NULL.
 * Outputs (any may be NULL): xout[2nj], cost, Fx[2nj*nx], Fu[2nj*nu], Lx, Lu, Lxx, Lxu, Luu. */
void aslr_cpu_dam(const aslr_chain_t *c, const aslr_model_t *m, const double *frame_ref,
                  const double *x, const double *u, double *xout, double *cost, double *Fx,
                  double *Fu, double *Lx, double *Lu, double *Lxx, double *Lxu, double *Luu);
/* IntegratedActionModelEulerASR.calc + calcDiff on one (x,u): xnext, cost and (if rec != NULL)
 * the derivative record in DERIV layout. */
void aslr_cpu_knot(const aslr_chain_t *c, const aslr_model_t *m, const double *frame_ref,
                   const double *x, const double *u, double *xnext, double *cost, double *rec);

/* ---- batched problem-level entry points (layouts of aslr_to_amd.h) --------------------- */
int aslr_cpu_calc(const aslr_problem_desc_t *d, const double *xs, const double *us, double *xnext,
                  double *cost);
int aslr_cpu_calc_diff(const aslr_problem_desc_t *d, const double *xs, const double *us,
                       double *xnext, double *cost, double *deriv);
/* backward pass for every trajectory at regularisation xreg[b]; feasible[b] selects the gap
 * terms.  Returns per-trajectory failure in fail[b] (1 = Cholesky failure / NaN). */
int aslr_cpu_backward_pass(const aslr_problem_desc_t *d, const aslr_solver_params_t *sp,
                           const double *deriv, const double *gaps, const double *us,
                           const double *xreg, const int32_t *feasible, double *kgain,
                           double *kff /* in: warm start, out */, double *qu, double *vx,
                           double *vxx, double *d1, double *d2, double *stop, int32_t *fail);
/* forward pass for ONE step length alpha (DDP / BoxDDP rule; FDDP when sp->solver says so). */
int aslr_cpu_forward_pass(const aslr_problem_desc_t *d, const aslr_solver_params_t *sp,
                          double alpha, const double *xs, const double *us, const double *kgain,
                          const double *kff, const double *gaps, const int32_t *feasible,
                          double *xs_try, double *us_try, double *cost_try, int32_t *fail);
/* full solve; xs/us in-out; traj_f [ASLR_TF_COUNT][B], traj_i [ASLR_TI_COUNT][B].
 * nthreads <= 1: serial; > 1: OpenMP over trajectories. */
int aslr_cpu_solve(const aslr_problem_desc_t *d, const aslr_solver_params_t *sp, double *xs,
                   double *us, double *traj_f, int32_t *traj_i, int32_t nthreads);
/* the same solve, also recording the per-iteration solver state in the layout of aslr_set_iteration_log
 * (log [log_cap][ASLR_LOG_COUNT][B]; NULL: no log) */
int aslr_cpu_solve_log(const aslr_problem_desc_t *d, const aslr_solver_params_t *sp, double *xs,
                       double *us, double *traj_f, int32_t *traj_i, int32_t nthreads, double *log, int32_t log_cap);
/* data.r: the stacked cost residuals of one (x, u) in the order of m->costs (integrated_action.py:17-18) */
void aslr_cpu_dam_residuals(const aslr_chain_t *c, const aslr_model_t *m, const double *frame_ref,
                            const double *x, const double *u, double *r);
/* BoxQP: returns iterations used; x in (warm start) / out; Hff_inv is n*n with
 * only the leading nf*nf block meaningful; free_idx/clamped_idx sized n. */
int aslr_cpu_boxqp(int n, const double *H, const double *q, const double *lb, const double *ub,
                   double *x, int maxiter, double th_acceptstep, double th_grad, double reg,
                   double *Hff_inv, int32_t *free_idx, int32_t *nf, int32_t *clamped_idx,
                   int32_t *nc);
/* ShootingProblem.quasiStatic for node model `mi` at state x */
int aslr_cpu_quasi_static(const aslr_problem_desc_t *d, int mi, const double *frame_ref,
                          const double *x, double *u, int maxiter, double tol);

#ifdef __cplusplus
}
#endif
#endif
