/*
 * aslr_to_amd.h -- C ABI of the MI355X-native batched DDP / FDDP / BoxDDP hot path
 *                  for aslr_to's soft-actuator (SEA / VSA) free-forward-dynamics models.
 *
 * The reference (spykspeigel/aslr_to) has no FFI of its own: its hot path sits behind
 * Crocoddyl's Python class protocol (Boost.Python bindings of C++ abstract classes).
 * This header is the C ABI that the Python layer `aslr_to_amd` binds with ctypes, and that a
 * reference maintainer would bind in place of the per-knot Python callbacks
 * (see INTEGRATION.md).  Every entry point cites the reference interface it replaces
 * (paths relative to the reference checkout).
 *
 * Conventions
 *   - plain C types only: int32_t, double, raw pointers; no torch / C++ types;
 *   - all arithmetic is IEEE float64 (dtype "f64");
 *   - every function returns int: 0 = ok, <0 = error (see ASLR_E_*); no exceptions cross
 *     the ABI.  Numerical events (Cholesky failure, NaN, reg at max, not converged) are
 *     NOT errors: they are per-trajectory status bits (ASLR_ST_*), mirroring Crocoddyl's
 *     catch-and-regularise behaviour and its `solve -> bool`;
 *   - the caller owns every buffer.  The library owns only the opaque handle; the device
 *     workspace is ONE caller-allocated buffer (a torch tensor) that the library carves into
 *     named regions (aslr_region);
 *   - kernels are enqueued on the caller's hipStream_t (passed as void*); nothing in the
 *     data path calls hipDeviceSynchronize or allocates.
 *
 * State layout (python/aslr_to/statemultibody_aslr.py:7-11): x = [q_l, q_m, v_l, v_m],
 * nx = ndx = 4*nj for the revolute chains of every BASELINE config; nu = nj (SEA) or 2*nj (VSA).
 */
#ifndef ASLR_TO_AMD_H
#define ASLR_TO_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ASLR_ABI_VERSION 2

#define ASLR_MAX_NJ     7   /* link-side DoF (2-DoF arm, 7-DoF arm)          */
#define ASLR_MAX_NX     28  /* 4 * ASLR_MAX_NJ                                */
#define ASLR_MAX_NU     14  /* 2 * ASLR_MAX_NJ (VSA)                          */
#define ASLR_MAX_COSTS  6   /* cost terms per CostModelSum                    */
#define ASLR_MAX_MODELS 4   /* distinct action models per shooting problem    */
#define ASLR_NALPHA     10  /* Crocoddyl line-search step lengths 2^-j, j<10  */

/* error codes */
#define ASLR_OK            0
#define ASLR_E_INVALID    -1  /* bad argument / unsupported size combination */
#define ASLR_E_HIP        -2  /* a HIP runtime call failed                   */
#define ASLR_E_NODEVICE   -3  /* no GPU visible                              */
#define ASLR_E_WORKSPACE  -4  /* workspace too small / misaligned            */

/* differential action model kind */
#define ASLR_DAM_SEA 0 /* DifferentialFreeASRFwdDynamicsModel (python/aslr_to/free_fwddyn_asr.py:6)  */
#define ASLR_DAM_VSA 1 /* DifferentialFreeFwdDynamicsModelVSA (python/aslr_to/free_fwddyn_vsa.py:6) */

/* cost term kind */
#define ASLR_COST_FRAME_PLACEMENT 0 /* CostModelResidual(ResidualModelFramePlacementASR) (python/aslr_to/residual_frame_placement.py:7) */
#define ASLR_COST_STATE           1 /* CostModelResidual(ResidualModelState)   (examples/two_dof_vsa_boxddp.py:30-31) */
#define ASLR_COST_CONTROL         2 /* CostModelResidual(ResidualModelControl) (examples/two_dof_vsa_boxddp.py:33-34) */
#define ASLR_COST_PENDULUM        3 /* CostModelDoublePendulum (python/aslr_to/__init__.py:223-259) */
#define ASLR_COST_STIFFNESS       4 /* CostModelStiffness (python/aslr_to/stiffness_cost.py:6-22)   */

/* solver kind */
#define ASLR_SOLVER_DDP    0 /* crocoddyl.SolverDDP    (north_star)                        */
#define ASLR_SOLVER_FDDP   1 /* crocoddyl.SolverFDDP   (examples/two_dof_sea.py:69)        */
#define ASLR_SOLVER_BOXDDP 2 /* crocoddyl.SolverBoxDDP (examples/two_dof_vsa_boxddp.py:69) */

/* per-trajectory status bits */
#define ASLR_ST_CONVERGED   1  /* was_feasible && stop < th_stop  -> solve() returned true  */
#define ASLR_ST_REG_MAX     2  /* regularisation hit reg_max      -> solve() returned false */
#define ASLR_ST_BACKWARD_ERR 4 /* at least one Cholesky failure / NaN in a backward pass (recovered by regularisation) */
#define ASLR_ST_FORWARD_ERR 8  /* at least one line-search trial produced NaN/Inf (that alpha was skipped)             */
/* CONVERGED and REG_MAX are OUTCOMES (what solve() returns, where it stopped) and BACKWARD_ERR records a decision the
 * solver took (regularise and redo); these are reproducible and comparable between implementations.  FORWARD_ERR is an
 * ADVISORY NOTE: it says that some rejected trial rollout overflowed.  A rollout that is rejected anyway can be
 * unstable (|x| doubling per knot); whether it crosses 1e30 inside the horizon or stays just below depends on the last
 * bits of the gains, so two correct implementations (or two builds of this one) may disagree on this bit for the same
 * problem while agreeing on every iterate.  Do not branch on it; the parity tests mask it (tests/_parity.py). */
#define ASLR_ST_ADVISORY_MASK ASLR_ST_FORWARD_ERR

/* Fixed-base serial chain of revolute joints (stands in for pinocchio.Model; the URDFs of
 * example_robot_data are unobtainable offline, see robots.py).  Joint j's parent is joint j-1
 * (the base for j = 0).  Pinocchio conventions: joint_R/joint_p = placement of the joint frame
 * in its parent joint frame at q = 0; inertia = rotational inertia about the COM, expressed in
 * the joint frame; com = lever of the COM in the joint frame. */
typedef struct aslr_chain {
  int32_t nj;
  int32_t _pad0;
  double gravity[3];                 /* model.gravity.linear (examples/two_dof_sea.py:20) */
  double joint_R[ASLR_MAX_NJ][9];    /* row-major 3x3 */
  double joint_p[ASLR_MAX_NJ][3];
  double axis[ASLR_MAX_NJ][3];       /* unit rotation axis in the joint frame */
  double mass[ASLR_MAX_NJ];
  double com[ASLR_MAX_NJ][3];
  double inertia[ASLR_MAX_NJ][9];    /* row-major symmetric 3x3 */
} aslr_chain_t;

/* One entry of a CostModelSum (examples/two_dof_vsa_boxddp.py:40-48). */
typedef struct aslr_cost {
  int32_t type;                      /* ASLR_COST_*                                          */
  int32_t frame_joint;               /* FRAME_PLACEMENT: joint the frame is attached to      */
  double weight;                     /* addCost(name, cost, weight)                          */
  double act_w[ASLR_MAX_NX];         /* ActivationModelWeightedQuad weights (ones for Quad)  */
  double ref[ASLR_MAX_NX];           /* STATE: xref; CONTROL: uref; STIFFNESS: Kref;
                                        FRAME_PLACEMENT: reference placement R[9] (row-major), p[3] */
  double frame_R[9];                 /* FRAME_PLACEMENT: frame placement on frame_joint      */
  double frame_p[3];
  double lambda;                     /* STIFFNESS: lamda (python/aslr_to/stiffness_cost.py:11) */
} aslr_cost_t;

/* One IntegratedActionModelEulerASR(differential, dt) (python/aslr_to/integrated_action.py:6). */
typedef struct aslr_model {
  int32_t dam;                       /* ASLR_DAM_*                                           */
  int32_t nu;                        /* SEA: nj ; VSA: 2*nj                                   */
  int32_t ncosts;
  int32_t has_u_limits;              /* u_lb/u_ub assigned (examples/two_dof_vsa_boxddp.py:59-60) */
  double dt;                         /* 0 for the terminal model (examples/two_dof_sea.py:57-58) */
  double K[ASLR_MAX_NJ * ASLR_MAX_NJ];   /* SEA spring stiffness, row-major nj x nj            */
  double B[ASLR_MAX_NJ * ASLR_MAX_NJ];   /* motor inertia, row-major nj x nj                    */
  double S[ASLR_MAX_NJ * ASLR_MAX_NJ];   /* SEA: motor rows of dtau_du (nj x nu): tau_m = S u.
                                            Identity for ASRActuation; selector for
                                            ActuationModelDoublePendulum (python/aslr_to/__init__.py:262-290) */
  double u_lb[ASLR_MAX_NU];
  double u_ub[ASLR_MAX_NU];
  aslr_cost_t costs[ASLR_MAX_COSTS];
} aslr_model_t;

/* crocoddyl.ShootingProblem(x0, runningModels, terminalModel), batched over B trajectories that
 * share structure (examples/two_dof_vsa_boxddp.py:66).  Pointers are HOST pointers, read during
 * create only. */
typedef struct aslr_problem_desc {
  int32_t B;                         /* trajectories in this shard                           */
  int32_t T;                         /* running knots                                        */
  int32_t nmodels;
  int32_t _pad0;
  aslr_chain_t chain;
  aslr_model_t models[ASLR_MAX_MODELS];
  const int32_t *node_model;         /* [T+1] model index per node; node T is the terminal   */
  const double *x0;                  /* [B][nx]                                              */
  const double *frame_ref;           /* optional [B][12] per-trajectory reference placement
                                        (R row-major, p) overriding every FRAME_PLACEMENT ref */
} aslr_problem_desc_t;

/* Solver parameters: crocoddyl.SolverDDP/FDDP/BoxDDP members. */
typedef struct aslr_solver_params {
  int32_t solver;                    /* ASLR_SOLVER_*                                        */
  int32_t maxiter;
  int32_t is_feasible;               /* solve(..., isFeasible)                               */
  int32_t fixed_iterations;          /* 1: never stop on convergence (throughput benchmarking) */
  double reg_init;                   /* NaN -> reg_min                                       */
  double th_stop;                    /* 1e-9 (examples set 1e-7)                             */
  double th_grad;                    /* 1e-12 */
  double th_gaptol;                  /* 1e-16 */
  double th_stepdec;                 /* 0.5   */
  double th_stepinc;                 /* 0.01  */
  double th_acceptstep;              /* 0.1   */
  double th_acceptnegstep;           /* 2 (FDDP) */
  double reg_min;                    /* 1e-9  */
  double reg_max;                    /* 1e9   */
  double reg_incfactor;              /* 10    */
  double reg_decfactor;              /* 10    */
  int32_t boxqp_maxiter;             /* 100   */
  int32_t _pad0;
  double boxqp_th_acceptstep;        /* 0.1   */
  double boxqp_th_grad;              /* 1e-5: SolverBoxDDP builds its BoxQP as (nu, 100, 0.1, 1e-5, 0.) */
  double boxqp_reg;                  /* 0     */
} aslr_solver_params_t;

/* Named regions of the device workspace.  Layouts (doubles unless noted), time-major so that a
 * wavefront's accesses to consecutive trajectories coalesce:
 *   XS      [T+1][B][nx]        candidate states      (solver.xs)
 *   US      [T][B][nu]          candidate controls    (solver.us)
 *   XNEXT   [T+1][B][nx]        data.xnext per node
 *   COST    [T+1][B]            data.cost per node
 *   DERIV   [T+1][B][rec]       per-knot record  Fx(nx*nx) Fu(nx*nu) Lxx(nx*nx) Lxu(nx*nu)
 *                               Luu(nu*nu) Lx(nx) Lu(nu), row-major blocks, rec padded to 16 doubles
 *   GAPS    [T+1][B][nx]        fs
 *   KGAIN   [T][B][nu*nx]       solver.K (row-major nu x nx)
 *   KFF     [T][B][nu]          solver.k
 *   QU      [T][B][nu]          solver.Qu
 *   VX      [T+1][B][nx]        solver.Vx
 *   VXX     [T+1][B][nx*nx]     solver.Vxx
 *   XS_TRY  [NALPHA][T+1][slab]   line-search candidates; slab = the B candidates of one (alpha, knot):
 *   US_TRY  [NALPHA][T][slab]       entries of width w = nx / nu doubles.  For even w <= 8 the slab is
 *                                   PIECE-INTERLEAVED in groups of 4 trajectories: [ceil(B / 4)][w / 2][4][2], i.e. entry e of
 *                                   trajectory b at ((b / 4) (w / 2) + e / 2) 8 + (b % 4) 2 + e % 2 (the four trajectories a wave
 *                                   of the rollout handles store neighbouring 16-byte pieces); otherwise plain [B][w].
 *                                   ASLR_CAND_SLAB / ASLR_CAND_OFFSET below.
 *   VXXF    [T+1][B][nx]        Vxx[t] fs[t] (FDDP expected improvement)
 *   COST_TRY [NALPHA][T+1][B]   node costs of every line-search candidate
 *   DYN     [T+1][B][2nj+3nj^2] rigid-body intermediates of calcDiff (xout, M^-1, dtau/dq, dtau/dv) handed from
 *                               the team kernel to the record assembly; only for chains with nj > 2 (else empty)
 *   TRAJ_F  [ASLR_TF_COUNT][B]  per-trajectory doubles (ASLR_TF_*)
 *   TRAJ_I  [ASLR_TI_COUNT][B]  per-trajectory int32   (ASLR_TI_*)
 *   POOL_SAVE [B][nx + 12]      scratch of aslr_solve_pool (the handle's x0 / frame_ref while pool problems occupy the slots)
 */
enum aslr_region_id {
  ASLR_R_XS = 0, ASLR_R_US, ASLR_R_XNEXT, ASLR_R_COST, ASLR_R_DERIV, ASLR_R_GAPS,
  ASLR_R_KGAIN, ASLR_R_KFF, ASLR_R_QU, ASLR_R_VX, ASLR_R_VXX, ASLR_R_XS_TRY, ASLR_R_US_TRY,
  ASLR_R_TRAJ_F, ASLR_R_TRAJ_I, ASLR_R_X0, ASLR_R_FRAME_REF, ASLR_R_VXXF, ASLR_R_DESC,
  ASLR_R_NODE_MODEL, ASLR_R_COST_TRY, ASLR_R_DYN,
  ASLR_R_POOL_SAVE /* [B][nx + 12]: the handle's own x0 / frame_ref columns while aslr_solve_pool streams problems through the slots */,
  ASLR_R_COUNT
};

/* rows of TRAJ_F */
/* candidate slabs (XS_TRY / US_TRY): doubles per (alpha, knot) and offset of entry e of trajectory b inside one */
#define ASLR_CAND_INTERLEAVED(w) ((w) % 2 == 0 && (w) <= 8)
#define ASLR_CAND_SLAB(B, w) (ASLR_CAND_INTERLEAVED(w) ? (((int64_t)(B) + 3) / 4) * 4 * (w) : (int64_t)(B) * (w))
#define ASLR_CAND_OFFSET(b, e, w) \
  (ASLR_CAND_INTERLEAVED(w) ? (((int64_t)(b) / 4) * ((w) / 2) + (e) / 2) * 8 + ((b) % 4) * 2 + (e) % 2 : (int64_t)(b) * (w) + (e))

enum {
  ASLR_TF_COST = 0, ASLR_TF_STOP, ASLR_TF_XREG, ASLR_TF_D1, ASLR_TF_D2, ASLR_TF_STEP,
  ASLR_TF_DV, ASLR_TF_DVEXP, ASLR_TF_DG, ASLR_TF_DQ, ASLR_TF_COST_TRY0 /* ..+NALPHA */,
  ASLR_TF_DVTRY0 = ASLR_TF_COST_TRY0 + ASLR_NALPHA /* FDDP dv per alpha, ..+NALPHA */,
  ASLR_TF_COUNT = ASLR_TF_DVTRY0 + ASLR_NALPHA
};
/* rows of TRAJ_I */
enum {
  ASLR_TI_ITER = 0, ASLR_TI_STATUS, ASLR_TI_FEASIBLE, ASLR_TI_WAS_FEASIBLE, ASLR_TI_RECALC,
  ASLR_TI_ACCEPTED /* index of accepted alpha, -1 none */, ASLR_TI_DONE, ASLR_TI_NTRIALS,
  ASLR_TI_GAPFLAG /* some |gap| >= th_gaptol seen by the last calcDiff sweep */,
  ASLR_TI_TRYFAIL0 /* ..+NALPHA: the rollout of that step length produced NaN / Inf ("forward_error") */,
  ASLR_TI_COUNT = ASLR_TI_TRYFAIL0 + ASLR_NALPHA
};

/* Per-iteration log (aslr_set_iteration_log): what crocoddyl.CallbackLogger / CallbackVerbose read from the solver at
 * the end of an iteration (examples/double_pendulum.py:77-79, examples/two_dof_sea.py:75), per trajectory.
 * Layout [capacity][ASLR_LOG_COUNT][B] doubles; row i of trajectory b is written by that trajectory's i-th
 * iteration (the integer fields as doubles).  Iterations past `capacity` are not recorded. */
enum {
  ASLR_LOG_COST = 0,  /* solver.cost after the iteration                                        */
  ASLR_LOG_STOP,      /* solver.stoppingCriteria()                                              */
  ASLR_LOG_XREG,      /* solver.x_reg (= u_reg) after the regularisation update                 */
  ASLR_LOG_STEP,      /* solver.stepLength: the accepted step length, or the last one tried     */
  ASLR_LOG_D1, ASLR_LOG_D2, /* solver.expectedImprovement() (CallbackLogger.grads = -d2)        */
  ASLR_LOG_DV, ASLR_LOG_DVEXP,
  ASLR_LOG_ACCEPTED,  /* index of the accepted step length, -1: every trial rejected            */
  ASLR_LOG_STATUS,    /* ASLR_ST_* bits as a callback of the iteration sees them (before the convergence test) */
  ASLR_LOG_FEASIBLE,  /* solver.isFeasible after the iteration                                  */
  ASLR_LOG_COUNT
};

typedef struct aslr_region {
  int64_t offset;                    /* in bytes from the workspace base                     */
  int64_t bytes;
} aslr_region_t;

typedef struct aslr_problem aslr_problem_t; /* opaque */

/* ---- ABI self-description (callable without a GPU) ------------------------------------ */
int aslr_abi_version(void);
/* sizeof() of the POD structs above as compiled, so a binding can check its mirror:
 * which = 0 chain, 1 cost, 2 model, 3 problem_desc, 4 solver_params, 5 region, 6 pool */
int64_t aslr_sizeof(int which);
/* record length in doubles (padded) for given nx, nu */
int32_t aslr_record_len(int32_t nx, int32_t nu);
/* fill defaults of crocoddyl.SolverDDP */
void aslr_solver_params_default(aslr_solver_params_t *p, int32_t solver);
/* workspace bytes needed by a problem of this description (no GPU needed) */
int64_t aslr_workspace_bytes(const aslr_problem_desc_t *desc);

/* ---- problem handle -------------------------------------------------------------------- */
/* Replaces crocoddyl.ShootingProblem(x0, [runningModel]*T, terminalModel)
 * (examples/two_dof_vsa_boxddp.py:66) for B trajectories.  `workspace` is a device buffer of at
 * least aslr_workspace_bytes(desc) bytes, 256-byte aligned, owned by the caller and kept alive
 * until destroy.  Uploads the description, x0 and frame_ref on `stream`. */
int aslr_problem_create(const aslr_problem_desc_t *desc, void *workspace, int64_t workspace_bytes,
                        void *stream, aslr_problem_t **out);
int aslr_problem_destroy(aslr_problem_t *p);
int aslr_problem_region(const aslr_problem_t *p, int32_t region_id, aslr_region_t *out);

/* ---- the hot path ---------------------------------------------------------------------- */
/* ShootingProblem.calc(xs, us): IntegratedActionModelEulerASR.calc on every node
 * (python/aslr_to/integrated_action.py:13-26 -> free_fwddyn_asr.py:20-56 / free_fwddyn_vsa.py:20-57).
 * Reads XS, US; writes XNEXT, COST. */
int aslr_calc(aslr_problem_t *p, void *stream);
/* ShootingProblem.calcDiff(xs, us) (+ the calc it relies on): integrated_action.py:28-42 ->
 * free_fwddyn_asr.py:58-92 / free_fwddyn_vsa.py:59-94, residual_frame_placement.py:17-24.
 * Reads XS, US; writes XNEXT, COST, DERIV. */
int aslr_calc_diff(aslr_problem_t *p, void *stream);
/* SolverDDP.backwardPass / SolverBoxDDP.computeGains at regularisation
 * TRAJ_F[XREG] with gaps GAPS (used when TRAJ_I[FEASIBLE] == 0).  Reads DERIV, GAPS, US, KFF;
 * writes KGAIN, KFF, QU, VX, VXX, TRAJ_F[D1,D2,STOP,DG,DQ], TRAJ_I[STATUS]. */
int aslr_backward_pass(aslr_problem_t *p, const aslr_solver_params_t *sp, void *stream);
/* SolverDDP/FDDP/BoxDDP.forwardPass for all ASLR_NALPHA step lengths.
 * Reads XS, US, KGAIN, KFF, GAPS; writes XS_TRY, US_TRY, COST_TRY, TRAJ_F[COST_TRY0..] (NaN for a
 * step length whose rollout failed). */
int aslr_forward_pass(aslr_problem_t *p, const aslr_solver_params_t *sp, void *stream);
/* solver.solve(init_xs, init_us, maxiter, isFeasible, regInit) (examples/two_dof_vsa_boxddp.py:81):
 * XS/US hold the warm start on entry and the solution on exit; TRAJ_F/TRAJ_I the per-trajectory
 * cost / stop / iter / status.  Enqueues everything on `stream`; `iters_done` (host, optional)
 * receives the number of lock-step batch iterations launched.  If sp->fixed_iterations == 0 the
 * host polls the active-trajectory count every `poll_every` iterations (one 4-byte D2H). */
int aslr_solve(aslr_problem_t *p, const aslr_solver_params_t *sp, int32_t poll_every,
               void *stream, int32_t *iters_done);
/* A POOL of P shooting problems of the same structure (same chain, models, T) solved through the B trajectory slots of
 * the handle, each to its own convergence: the reference solves its problems one after the other
 * (`solver.solve(...)` per problem, examples/two_dof_vsa_boxddp.py:81); a lock-step batch of B waits for its slowest
 * member (BoxDDP at C3: 400 lock-step iterations where a trajectory needs 107 on average).  Here a slot whose problem has
 * stopped (converged, regularisation at its maximum, or sp->maxiter iterations) is flushed to the outputs and refilled
 * with the next problem of the pool every `refill_every` iterations, on the device, so the slots stay busy.
 * Every problem starts from its own initial guess -- xs_init / us_init, e.g. `[x0] * (T + 1)` and the quasi-static controls
 * of examples/two_dof_sea.py:77-81, or a cold start (xs = 0, us = 0: `solve([], [], maxiter)`) when they are NULL -- and
 * goes through exactly the iterations aslr_solve would give it: results do not depend on B, on the slot or on
 * refill_every, bit for bit.
 * All pointers are DEVICE pointers owned by the caller.
 * Side effects on the handle: while the pool runs, the slots' X0 / FRAME_REF columns hold the pool's problems; the
 * handle's own are kept in POOL_SAVE and put back before the call returns, so a later aslr_solve / rollout sees the
 * problems the handle was created with.  XS / US / the per-trajectory state are left with the last slot contents
 * (every solver entry point re-initialises them from its own arguments).  Pool targets always take the general SE(3)
 * log map (the opt-in closed-form reach residual is validated for the create-time references only).  The call fails
 * (ASLR_E_INVALID, aslr_last_error) if fewer than P problems were flushed within its iteration bound. */
typedef struct aslr_pool {
  int32_t P;                  /* problems in the pool                                                          */
  int32_t _pad0;
  const double *x0;           /* [P][nx] initial states                                                        */
  const double *frame_ref;    /* [P][12] reference placements (R row-major, p), or NULL; needs a problem created
                                 with a frame_ref table                                                       */
  double *xs_out;             /* [P][T+1][nx] solutions (batch-major)                                          */
  double *us_out;             /* [P][T][nu]                                                                    */
  double *stat_f;             /* [P][4]: cost, stop, x_reg, last step length                                   */
  int32_t *stat_i;            /* [P][2]: iterations, ASLR_ST_* status word                                     */
  int32_t *slot_problem;      /* [B] scratch: the problem in each slot (-1: idle)                              */
  int32_t *counters;          /* [2] scratch: next problem to hand out, problems finished                      */
  const double *xs_init;      /* [P][T+1][nx] initial guesses (batch-major), or NULL: zeros                     */
  const double *us_init;      /* [P][T][nu], or NULL: zeros                                                    */
} aslr_pool_t;
/* `iters_done` (host, optional) receives the lock-step iterations launched.  The host polls the finished-problem
 * counter every `poll_every` iterations (one 4-byte D2H). */
int aslr_solve_pool(aslr_problem_t *p, const aslr_solver_params_t *sp, const aslr_pool_t *pool, int32_t refill_every,
                    int32_t poll_every, void *stream, int32_t *iters_done);

/* one lock-step DDP iteration (calcDiff sweep + backward pass + line search), the unit the
 * benchmark's "step" times.  `first` != 0 re-initialises the per-trajectory solver state. */
int aslr_iterate(aslr_problem_t *p, const aslr_solver_params_t *sp, int32_t first, void *stream);
/* `n` lock-step iterations (aslr_iterate n times; `first` applies to the first of them).  With sub-shards
 * (aslr_set_subshards) each sub-shard runs its n iterations on its own stream, free of the others; the caller's stream
 * is forked from on entry and joined on exit, so for the caller the call is "n iterations enqueued on my stream". */
int aslr_iterate_n(aslr_problem_t *p, const aslr_solver_params_t *sp, int32_t first, int32_t n, void *stream);
/* Iterate the shard as `n` (1..8) contiguous sub-shards of trajectories: the first on the caller's stream, the others on
 * internal HIP streams created here (boundaries on multiples of 64 trajectories).  More streams in use than the HIP
 * runtime has hardware queues (GPU_MAX_HW_QUEUES, default 4) serialise: n <= 4 unless that variable is raised.  Trajectories are independent (one ShootingProblem per solve in the
 * reference, examples/two_dof_vsa_boxddp.py:66), so results do not depend on n, bit for bit; what changes is the
 * schedule: the latency-bound sweeps (backward pass, rollout: one wave per SIMD) of one sub-shard overlap the
 * streaming kernels (calc / calcDiff, trial costs) of the others.  Applies to aslr_iterate_n and aslr_solve;
 * aslr_iterate_timed and the stand-alone passes always cover the whole shard on the caller's stream.  n = 1: off. */
int aslr_set_subshards(aslr_problem_t *p, int32_t n);
/* aslr_iterate with HIP events recorded on `stream` around each of its three kernels
 * (calc/calcDiff sweep, backward pass, forward pass + line search); waits for the last event and
 * returns the three durations in milliseconds.  Measurement aid for bench.py's roofline leg. */
int aslr_iterate_timed(aslr_problem_t *p, const aslr_solver_params_t *sp, int32_t first, void *stream,
                       float *ms3);
/* commit the last accepted line-search candidate into XS/US (called by aslr_solve on exit). */
int aslr_finalize(aslr_problem_t *p, void *stream);
/* number of trajectories neither converged nor failed; synchronises `stream`. */
int aslr_count_active(aslr_problem_t *p, void *stream, int32_t *active);

/* DifferentialFree{ASR,VSA}FwdDynamicsModel.calc + calcDiff on `n` arbitrary points with model
 * `model_index` of the problem (the path the reference's unit tests exercise:
 * unittest/test_vsa_freefwddyn.py:26-38).  All pointers are DEVICE pointers; x [n][nx], u [n][nu];
 * outputs (row-major, any may be NULL): xout [n][2nj], cost [n], Fx [n][2nj*nx], Fu [n][2nj*nu],
 * Lx [n][nx], Lu [n][nu], Lxx [n][nx*nx], Lxu [n][nx*nu], Luu [n][nu*nu].  Trajectory 0's
 * frame_ref override (if any) applies. */
int aslr_dam_eval(aslr_problem_t *p, int32_t model_index, int32_t n, const double *x,
                  const double *u, double *xout, double *cost, double *Fx, double *Fu, double *Lx,
                  double *Lu, double *Lxx, double *Lxu, double *Luu, void *stream);

/* The cost residuals of DifferentialFree{ASR,VSA}FwdDynamicsModel.calc on `n` points: data.r, which
 * IntegratedActionModelEulerASR.calc copies (python/aslr_to/integrated_action.py:17-18) -- the residual vectors of
 * the model's cost terms stacked in the order of aslr_model_t.costs: FRAME_PLACEMENT 6 (log6 of Mref^-1 oMf,
 * python/aslr_to/residual_frame_placement.py:13-15), STATE nx, CONTROL nu, PENDULUM 6
 * (python/aslr_to/__init__.py:231), STIFFNESS nu/2 (python/aslr_to/stiffness_cost.py:15).  DEVICE pointers;
 * r [n][aslr_residual_len(model, nj)]. */
int aslr_dam_residuals(aslr_problem_t *p, int32_t model_index, int32_t n, const double *x, const double *u,
                       double *r, void *stream);
/* length of that stacked residual vector (host only, no GPU needed) */
int32_t aslr_residual_len(const aslr_model_t *m, int32_t nj);

/* data.differential.multibody.pinocchio.oMf[frame] (examples/two_dof_sea.py:82-86, examples/two_dof_vsa_boxddp.py:83-84):
 * world placement of a frame attached to joint `frame_joint` with local placement (frame_R row-major 3x3, frame_p;
 * HOST pointers) at `n` configurations.  x: DEVICE pointer, point i at x + i * x_stride doubles, its first nj entries
 * are the link positions q_l (so XS itself can be passed with x_stride = nx).  oMf: DEVICE [n][12] = R row-major, p. */
int aslr_frame_placement(aslr_problem_t *p, int32_t frame_joint, const double *frame_R, const double *frame_p,
                         int32_t n, const double *x, int64_t x_stride, double *oMf, void *stream);

/* Per-iteration log of the solver state (layout and fields: ASLR_LOG_* above).  `log` is a DEVICE buffer of
 * capacity * ASLR_LOG_COUNT * B doubles owned by the caller and kept alive while set; NULL switches logging off.
 * Written by the line-search kernel of aslr_iterate / aslr_solve: no host round trip per iteration. */
int aslr_set_iteration_log(aslr_problem_t *p, double *log, int32_t capacity);

/* ShootingProblem.quasiStatic(xs) (examples/two_dof_sea.py:78): for every running node, the control that
 * holds XS[t] still under the node's model, by Crocoddyl's base-class Gauss-Newton
 * (u = 0; u -= pinv(Fu) (xnext - x), at most `maxiter` times, until |du| <= tol; Crocoddyl uses 100, 1e-9).
 * pinv is the thresholded pseudo-inverse (singular directions of Fu, e.g. the VSA stiffness columns at q_l = q_m,
 * get no update).  Reads XS; writes US.  `iters_dev` (optional DEVICE pointer, [T][B] int32) receives the
 * iterations used. */
int aslr_quasi_static(aslr_problem_t *p, int32_t maxiter, double tol, int32_t *iters_dev, void *stream);

/* last HIP error string of this thread (static storage) */
const char *aslr_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* ASLR_TO_AMD_H */
