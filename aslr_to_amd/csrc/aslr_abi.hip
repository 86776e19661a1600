// aslr_abi.hip -- the C ABI of include/aslr_to_amd.h: handle, workspace carving, solver driver.
// The kernels live in aslr_calc_*.hip / aslr_backward_*.hip / aslr_forward_*.hip.
#include <new>
#include <vector>

#include "aslr_common.hpp"

using namespace aslr;

namespace {
thread_local char g_err_storage[kErrLen] = {0};
}
namespace aslr {
char *err_buf() { return g_err_storage; }
} // namespace aslr
#define g_err g_err_storage

namespace {

// per-trajectory solver state at solve() entry
__global__ void init_state_kernel(KArgs a, double reg0, int is_feasible) {
  const int b = a.b0 + blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= a.b1) return;
  const int B = a.B;
  for (int r = 0; r < ASLR_TF_COUNT; ++r) a.traj_f[r * B + b] = 0.0;
  for (int r = 0; r < ASLR_TI_COUNT; ++r) a.traj_i[r * B + b] = 0;
  a.traj_f[ASLR_TF_XREG * B + b] = reg0;
  a.traj_i[ASLR_TI_FEASIBLE * B + b] = is_feasible;
  a.traj_i[ASLR_TI_RECALC * B + b] = 1;
  a.traj_i[ASLR_TI_ACCEPTED * B + b] = -1;
}

__global__ void reset_accepted_kernel(KArgs a) {
  const int b = a.b0 + blockIdx.x * blockDim.x + threadIdx.x;
  if (b < a.b1) a.traj_i[ASLR_TI_ACCEPTED * a.B + b] = -1;
}

// aslr_solve_pool: flush the slots whose problem has stopped and hand them the next problem of the pool.
// One 64-thread block per slot; nothing to do for a slot that is still iterating.
struct PoolDev {
  int32_t P, nx, nu;
  const double *x0, *frame_ref, *xs_init, *us_init;
  double *xs_out, *us_out, *stat_f;
  int32_t *stat_i, *slot_problem, *counters;
};
__global__ void __launch_bounds__(64) pool_refill_kernel(KArgs a, PoolDev pl, double reg0, int is_feasible) {
  const int b = blockIdx.x, tid = threadIdx.x, B = a.B, T = a.T, nx = pl.nx, nu = pl.nu;
  int32_t *TI = a.traj_i;
  double *TF = a.traj_f;
  __shared__ int next_j, skip;
  const int j = pl.slot_problem[b];
  if (tid == 0) { // one thread decides for the block (other blocks bump the counter during this launch)
    const int done = TI[ASLR_TI_DONE * B + b];
    skip = j >= 0 && !done; // still iterating
    if (j < 0 && __atomic_load_n(&pl.counters[0], __ATOMIC_RELAXED) >= pl.P) {
      // idle and nothing left to hand out (the counter only grows).  The slot must still be MARKED idle: in the first
      // launch of a pool smaller than the slot count a late block sees the counter past P before it has ever run, and a
      // slot left with DONE = 0 would be iterated with whatever state the handle held (on a fresh handle x_reg = 0: a
      // backward sweep that fails and cannot raise its regularisation)
      TI[ASLR_TI_DONE * B + b] = 1;
      TI[ASLR_TI_ACCEPTED * B + b] = -1;
      skip = 1;
    }
  }
  __syncthreads();
  if (skip) return;
  if (j >= 0) { // ---- flush: the last accepted candidate is the solution (solver.xs / solver.us) ----
    const int acc = TI[ASLR_TI_ACCEPTED * B + b];
    double *xo = pl.xs_out + (size_t)j * (T + 1) * nx, *uo = pl.us_out + (size_t)j * T * nu;
    for (int t = tid; t <= T; t += 64) {
      for (int i = 0; i < nx; ++i) {
        const size_t src = acc >= 0 ? ((size_t)acc * (T + 1) + t) * ASLR_CAND_SLAB(B, nx) + ASLR_CAND_OFFSET(b, i, nx) : 0;
        xo[(size_t)t * nx + i] = acc >= 0 ? a.xs_try[src] : a.xs[((size_t)t * B + b) * nx + i];
      }
      if (t < T)
        for (int i = 0; i < nu; ++i) {
          const size_t src = acc >= 0 ? ((size_t)acc * T + t) * ASLR_CAND_SLAB(B, nu) + ASLR_CAND_OFFSET(b, i, nu) : 0;
          uo[(size_t)t * nu + i] = acc >= 0 ? a.us_try[src] : a.us[((size_t)t * B + b) * nu + i];
        }
    }
    if (tid == 0) {
      pl.stat_f[4 * (size_t)j + 0] = TF[ASLR_TF_COST * B + b];
      pl.stat_f[4 * (size_t)j + 1] = TF[ASLR_TF_STOP * B + b];
      pl.stat_f[4 * (size_t)j + 2] = TF[ASLR_TF_XREG * B + b];
      pl.stat_f[4 * (size_t)j + 3] = TF[ASLR_TF_STEP * B + b];
      pl.stat_i[2 * (size_t)j + 0] = TI[ASLR_TI_ITER * B + b];
      pl.stat_i[2 * (size_t)j + 1] = TI[ASLR_TI_STATUS * B + b];
      atomicAdd(&pl.counters[1], 1);
    }
  }
  __syncthreads(); // (the flush reads this slot's columns; the refill below overwrites them)
  if (tid == 0) {
    const int n = atomicAdd(&pl.counters[0], 1);
    next_j = n < pl.P ? n : -1;
  }
  __syncthreads();
  const int jn = next_j;
  if (jn < 0) { // pool exhausted: the slot idles (DONE stays set)
    if (tid == 0) { pl.slot_problem[b] = -1; TI[ASLR_TI_DONE * B + b] = 1; TI[ASLR_TI_ACCEPTED * B + b] = -1; }
    return;
  }
  // ---- refill: problem jn starts in slot b from its initial guess (zeros: solve([], [], maxiter)) ----
  for (int t = tid; t <= T; t += 64) {
    const size_t tb = (size_t)t * B + b;
    for (int i = 0; i < nx; ++i) {
      a.xs[tb * nx + i] = pl.xs_init ? pl.xs_init[((size_t)jn * (T + 1) + t) * nx + i] : 0.0;
      a.gaps[tb * nx + i] = 0.0; a.vxxf[tb * nx + i] = 0.0;
    }
    if (t < T)
      for (int i = 0; i < nu; ++i) {
        a.us[tb * nu + i] = pl.us_init ? pl.us_init[((size_t)jn * T + t) * nu + i] : 0.0;
        a.kff[tb * nu + i] = 0.0;
      }
  }
  if (tid < nx) const_cast<double *>(a.x0)[(size_t)b * nx + tid] = pl.x0[(size_t)jn * nx + tid];
  if (tid < 12 && pl.frame_ref && a.frame_ref) const_cast<double *>(a.frame_ref)[12 * (size_t)b + tid] = pl.frame_ref[12 * (size_t)jn + tid];
  if (tid == 0) { // per-trajectory solver state, as init_state_kernel sets it
    for (int r = 0; r < ASLR_TF_COUNT; ++r) TF[r * B + b] = 0.0;
    for (int r = 0; r < ASLR_TI_COUNT; ++r) TI[r * B + b] = 0;
    TF[ASLR_TF_XREG * B + b] = reg0;
    TI[ASLR_TI_FEASIBLE * B + b] = is_feasible;
    TI[ASLR_TI_RECALC * B + b] = 1;
    TI[ASLR_TI_ACCEPTED * B + b] = -1;
    pl.slot_problem[b] = jn;
  }
}

} // namespace

// =================================================================================================
// host side: handle, workspace carving, dispatch
// =================================================================================================
// Sub-shard 0 runs on the caller's stream, the others on internal ones.  HIP maps streams onto GPU_MAX_HW_QUEUES hardware
// queues (4 unless that environment variable says otherwise): more streams in use than queues serialise them (measured:
// 5 streams on 4 queues run 1.6x SLOWER than one), so 4 sub-shards is the default ceiling and more need the variable.
constexpr int kMaxSub = 8;

struct aslr_problem {
  aslr_problem_desc_t desc; // host copy (pointers nulled)
  int nj, nx, nu, dam, rec;
  char *ws;
  int64_t ws_bytes;
  aslr_region_t regions[ASLR_R_COUNT];
  KArgs k;
  int32_t *h_done; // pinned staging for count_active
  int bwd_hs, blk_mfma; // launch-path switches (ASLR_BWD_HS, ASLR_BLK_MFMA), read from the environment at create time
  hipEvent_t ev[4];
  bool have_ev;
  // sub-shards (aslr_set_subshards): contiguous trajectory ranges [sub_b[s], sub_b[s + 1]) iterated on their own
  // streams, so that the latency-bound sweeps of one overlap the throughput-bound kernels of the others
  int nsub;
  int32_t sub_b[kMaxSub + 1];
  hipStream_t sub_stream[kMaxSub];
  hipEvent_t sub_fork, sub_join[kMaxSub];
  bool have_sub;
  unsigned sub_flags;
  // model-only chunks of the DERIV records (cost-weight diagonals): const_ok = the cost stacks allow skipping them,
  // const_written = a sweep that evaluated every knot of every trajectory has put them in place
  bool const_ok, const_written;
  bool const_pending_full; // aslr_iterate_n: the first sweep of this call is the one that puts them in place
  int pool_maxiter;        // > 0 inside aslr_solve_pool: trajectories stop by themselves after this many iterations
};

namespace {

int64_t align_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }

int desc_dims(const aslr_problem_desc_t *d, int *nj, int *nx, int *nu, int *dam) {
  if (!d || d->B <= 0 || d->T <= 0 || d->nmodels <= 0 || d->nmodels > ASLR_MAX_MODELS) return ASLR_E_INVALID;
  if (d->chain.nj <= 0 || d->chain.nj > ASLR_MAX_NJ) return ASLR_E_INVALID;
  *nj = d->chain.nj;
  *nx = 4 * d->chain.nj;
  *nu = d->models[0].nu;
  *dam = d->models[0].dam;
  for (int i = 0; i < d->nmodels; ++i) {
    const aslr_model_t &m = d->models[i];
    if (m.nu != *nu || m.dam != *dam) return ASLR_E_INVALID; // one (nu, dam) per problem
    if (m.dam == ASLR_DAM_VSA ? m.nu != 2 * *nj : m.nu != *nj) return ASLR_E_INVALID;
    if (m.ncosts < 0 || m.ncosts > ASLR_MAX_COSTS) return ASLR_E_INVALID;
    for (int c = 0; c < m.ncosts; ++c) {
      const aslr_cost_t &ct = m.costs[c];
      if (ct.type < 0 || ct.type > ASLR_COST_STIFFNESS) return ASLR_E_INVALID;
      if (ct.type == ASLR_COST_FRAME_PLACEMENT && (ct.frame_joint < 0 || ct.frame_joint >= *nj)) return ASLR_E_INVALID;
      if (ct.type == ASLR_COST_STIFFNESS && m.dam != ASLR_DAM_VSA) return ASLR_E_INVALID;
      if (ct.type == ASLR_COST_PENDULUM && *nj < 2) return ASLR_E_INVALID;
    }
  }
  return ASLR_OK;
}

void carve(const aslr_problem_desc_t *d, int nx, int nu, aslr_region_t *r, int64_t *total) {
  const int64_t B = d->B, T = d->T, T1 = T + 1, rec = rec_len_c(nx, nu), D = sizeof(double);
  int64_t sizes[ASLR_R_COUNT];
  sizes[ASLR_R_XS] = T1 * B * nx * D;
  sizes[ASLR_R_US] = T * B * nu * D;
  sizes[ASLR_R_XNEXT] = T1 * B * nx * D;
  sizes[ASLR_R_COST] = T1 * B * D;
  sizes[ASLR_R_DERIV] = T1 * B * rec * D;
  sizes[ASLR_R_GAPS] = T1 * B * nx * D;
  sizes[ASLR_R_KGAIN] = T * B * nu * nx * D;
  sizes[ASLR_R_KFF] = T * B * nu * D;
  sizes[ASLR_R_QU] = T * B * nu * D;
  sizes[ASLR_R_VX] = T1 * B * nx * D;
  sizes[ASLR_R_VXX] = T1 * B * nx * nx * D;
  sizes[ASLR_R_XS_TRY] = (int64_t)ASLR_NALPHA * T1 * ASLR_CAND_SLAB(B, nx) * D;
  sizes[ASLR_R_US_TRY] = (int64_t)ASLR_NALPHA * T * ASLR_CAND_SLAB(B, nu) * D;
  sizes[ASLR_R_TRAJ_F] = (int64_t)ASLR_TF_COUNT * B * D;
  sizes[ASLR_R_TRAJ_I] = (int64_t)ASLR_TI_COUNT * B * sizeof(int32_t);
  sizes[ASLR_R_X0] = B * nx * D;
  sizes[ASLR_R_FRAME_REF] = B * 12 * D;
  sizes[ASLR_R_VXXF] = T1 * B * nx * D;
  sizes[ASLR_R_DESC] = sizeof(DevDesc);
  sizes[ASLR_R_NODE_MODEL] = T1 * sizeof(int32_t);
  sizes[ASLR_R_COST_TRY] = (int64_t)ASLR_NALPHA * T1 * B * D;
  sizes[ASLR_R_DYN] = T1 * B * dyn_len_c(nx / 4) * D;
  sizes[ASLR_R_POOL_SAVE] = B * (nx + 12) * D;
  int64_t off = 0;
  for (int i = 0; i < ASLR_R_COUNT; ++i) {
    r[i].offset = off;
    r[i].bytes = sizes[i];
    off += align_up(sizes[i], 256);
  }
  *total = off;
}

// Gauss-Jordan inverse of the motor inertia B (np.linalg.inv(self.B), free_fwddyn_asr.py:41)
bool invert(int n, const double *A, double *Ainv) {
  std::vector<double> a(n * 2 * n);
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) { a[i * 2 * n + j] = A[i * n + j]; a[i * 2 * n + n + j] = i == j ? 1.0 : 0.0; }
  for (int col = 0; col < n; ++col) {
    int piv = col;
    for (int r = col + 1; r < n; ++r) if (std::fabs(a[r * 2 * n + col]) > std::fabs(a[piv * 2 * n + col])) piv = r;
    if (a[piv * 2 * n + col] == 0.0) return false;
    if (piv != col) for (int j = 0; j < 2 * n; ++j) std::swap(a[col * 2 * n + j], a[piv * 2 * n + j]);
    const double d = a[col * 2 * n + col];
    for (int j = 0; j < 2 * n; ++j) a[col * 2 * n + j] /= d;
    for (int r = 0; r < n; ++r) if (r != col) {
      const double f = a[r * 2 * n + col];
      if (f != 0.0) for (int j = 0; j < 2 * n; ++j) a[r * 2 * n + j] -= f * a[col * 2 * n + j];
    }
  }
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) Ainv[i * n + j] = a[i * 2 * n + n + j];
  return true;
}

// Planar restatement of the chain when every joint axis is +z and every joint placement is a rotation
// about z (exact tests: the tables are built from exact 0 / 1 entries).  ASLR_NO_PLANAR=1 disables it.
void fill_planar(const aslr_chain_t &c, PlanarChain *pl) {
  memset(pl, 0, sizeof *pl);
  const char *e = getenv("ASLR_NO_PLANAR");
  if (e && atoi(e)) return;
  double z = 0.0;
  for (int i = 0; i < c.nj; ++i) {
    const double *R = c.joint_R[i], *ax = c.axis[i];
    if (!(ax[0] == 0.0 && ax[1] == 0.0 && ax[2] == 1.0)) return;
    if (!(R[2] == 0.0 && R[5] == 0.0 && R[6] == 0.0 && R[7] == 0.0 && R[8] == 1.0)) return;
    if (!(R[0] == R[4] && R[1] == -R[3])) return;
    pl->cphi[i] = R[0];
    pl->sphi[i] = R[3];
    pl->phi[i] = std::atan2(R[3], R[0]);
    pl->px[i] = c.joint_p[i][0];
    pl->py[i] = c.joint_p[i][1];
    z += c.joint_p[i][2];
    pl->pz[i] = z; // world height of joint frame i
    pl->m[i] = c.mass[i];
    pl->cx[i] = c.com[i][0];
    pl->cy[i] = c.com[i][1];
    pl->izz[i] = c.inertia[i][8];
  }
  pl->gx = c.gravity[0];
  pl->gy = c.gravity[1];
  if (c.nj == 2) { // closed-form constants of the two-link chain (ChainPlanar<2>::mass2 / nle2)
    const double J1 = pl->izz[0] + pl->m[0] * (pl->cx[0] * pl->cx[0] + pl->cy[0] * pl->cy[0]);
    const double J2 = pl->izz[1] + pl->m[1] * (pl->cx[1] * pl->cx[1] + pl->cy[1] * pl->cy[1]);
    const double p2x = pl->px[1], p2y = pl->py[1];
    pl->two[0] = J1 + J2 + pl->m[1] * (p2x * p2x + p2y * p2y);
    pl->two[1] = J2;
    pl->two[2] = pl->m[1] * (p2x * pl->cx[1] + p2y * pl->cy[1]);
    pl->two[3] = pl->m[1] * (p2y * pl->cx[1] - p2x * pl->cy[1]);
    pl->two[4] = -(pl->m[0] * pl->cy[0] + pl->m[1] * p2y);
    pl->two[5] = pl->m[0] * pl->cx[0] + pl->m[1] * p2x;
    pl->two[6] = -pl->m[1] * pl->cy[1];
    pl->two[7] = pl->m[1] * pl->cx[1];
  }
  pl->ok = 1;
}

// rotation about z (exact pattern test, as for the joint placements); *c, *s its cos / sin
bool z_rotation(const double *R, double *c, double *s) {
  if (!(R[2] == 0.0 && R[5] == 0.0 && R[6] == 0.0 && R[7] == 0.0 && R[8] == 1.0)) return false;
  if (!(R[0] == R[4] && R[1] == -R[3])) return false;
  *c = R[0]; *s = R[3];
  return true;
}
bool is_identity3(const double *R) {
  for (int i = 0; i < 9; ++i) if (R[i] != (i % 4 == 0 ? 1.0 : 0.0)) return false;
  return true;
}

// Closed-form frame-placement residuals (ChainPlanar::reach_residual) need: a planar chain, every cost frame turned
// about z on its joint, every reference rotation -- cost defaults and per-trajectory overrides -- exactly the identity.
// OFF unless ASLR_PLANAR_REACH=1: the closed form agrees with the general SE(3) log to 1e-12 relative, but the last
// iterations of a converging solve compare trial costs that differ by 1e-12 relative (dV ~ 1e-10 on a cost of ~300),
// so a differently ROUNDED cost flips one of those line-search decisions in 12 % of the trajectories of the headline
// batch (512 of 4096 against 6 with the general path, profiles/r02/parity_headline_4096_planar_reach.txt), for 8 us
// per iteration: the trial-cost kernel is bound by reading the 396 MB of candidates, not by the log map.
void fill_planar_reach(const aslr_problem_desc_t *d, DevDesc *hd) {
  hd->planar.reach_ok = 0;
  for (int i = 0; i < d->nmodels; ++i)
    for (int c = 0; c < ASLR_MAX_COSTS; ++c) { hd->models[i].fr_c[c] = 1.0; hd->models[i].fr_s[c] = 0.0; hd->models[i].fr_phi[c] = 0.0; }
  const char *e = getenv("ASLR_PLANAR_REACH");
  if (!hd->planar.ok || !(e && atoi(e))) return;
  for (int i = 0; i < d->nmodels; ++i)
    for (int c = 0; c < d->models[i].ncosts; ++c) {
      const aslr_cost_t &ct = d->models[i].costs[c];
      if (ct.type != ASLR_COST_FRAME_PLACEMENT) continue;
      double cc, ss;
      if (!z_rotation(ct.frame_R, &cc, &ss) || !is_identity3(ct.ref)) return;
      hd->models[i].fr_c[c] = cc; hd->models[i].fr_s[c] = ss; hd->models[i].fr_phi[c] = std::atan2(ss, cc);
    }
  if (d->frame_ref)
    for (int b = 0; b < d->B; ++b) if (!is_identity3(d->frame_ref + 12 * (size_t)b)) return;
  hd->planar.reach_ok = 1;
}

SolverDev to_dev(const aslr_solver_params_t *sp, int standalone, int store_v) {
  SolverDev s;
  s.solver = sp->solver; s.fixed_iterations = sp->fixed_iterations;
  s.th_stop = sp->th_stop; s.th_grad = sp->th_grad; s.th_gaptol = sp->th_gaptol;
  s.th_stepdec = sp->th_stepdec; s.th_stepinc = sp->th_stepinc; s.th_acceptstep = sp->th_acceptstep;
  s.th_acceptnegstep = sp->th_acceptnegstep;
  s.reg_min = sp->reg_min; s.reg_max = sp->reg_max; s.reg_incfactor = sp->reg_incfactor;
  s.reg_decfactor = sp->reg_decfactor;
  s.boxqp_maxiter = sp->boxqp_maxiter; s.boxqp_th_acceptstep = sp->boxqp_th_acceptstep;
  s.boxqp_th_grad = sp->boxqp_th_grad; s.boxqp_reg = sp->boxqp_reg;
  s.standalone = standalone; s.store_v = store_v; s.maxiter_traj = 0;
  return s;
}

// ---- launch helpers: dispatch to the per-size translation units ----
int launch_calc(aslr_problem *p, bool diff, int mode, double th_gaptol, hipStream_t st, bool all_computed = false) {
  if (diff && p->const_ok && !(mode & kModeNoCompute)) {
    if (p->const_written) mode |= kModeSkipConst;
    else if (all_computed) p->const_written = true; // this launch writes them
  }
  if (p->nj == 2) return launch_calc_nj2(p->k, p->dam, diff, mode, th_gaptol, st);
  if (p->nj == 7) return launch_calc_nj7(p->k, p->dam, diff, mode, th_gaptol, st);
  snprintf(g_err, sizeof g_err, "unsupported nj=%d", p->nj);
  return ASLR_E_INVALID;
}

ModelLimits make_limits(const aslr_problem *p) {
  ModelLimits lim;
  memset(&lim, 0, sizeof lim);
  for (int i = 0; i < p->desc.nmodels; ++i) {
    lim.has[i] = p->desc.models[i].has_u_limits;
    for (int c = 0; c < ASLR_MAX_NU; ++c) { lim.lb[i][c] = p->desc.models[i].u_lb[c]; lim.ub[i][c] = p->desc.models[i].u_ub[c]; }
  }
  return lim;
}

int backward_hs(const aslr_problem *p) {
  // 0: each size picks its default decomposition; ASLR_BWD_HS forces the register-column kernel with that
  // many lanes per column (tests and comparisons).  Read once, when the handle is created.
  return p->bwd_hs;
}

// nj = 7 with VSA actuation (nx = 28, nu = 14) is built at the MODEL level only -- calc / calcDiff sweeps, dam_eval,
// dam_residuals, frame placements: what the reference exercises for that combination
// (unittest/test_free_placementcost_free_fwddyn.py:12-46) -- not in the solver kernels
int solver_unsupported(const aslr_problem *p) {
  if (p->nj == 7 && p->dam == ASLR_DAM_VSA) {
    snprintf(g_err, sizeof g_err, "the solver kernels are not built for (nj=7, VSA): model-level evaluation only");
    return ASLR_E_INVALID;
  }
  return ASLR_OK;
}

int launch_backward(aslr_problem *p, const SolverDev &sd, hipStream_t st, bool all_feasible = false) {
  if (int rc = solver_unsupported(p)) return rc;
  const int hs = backward_hs(p);
  const ModelLimits lim = make_limits(p);
  if (p->nx == 8) return launch_backward_nx8(p->k, p->nu, hs, sd, lim, all_feasible, st);
  if (p->nx == 28) return launch_backward_nx28(p->k, p->nu, (hs == 0 && !p->blk_mfma) ? -1 : hs, sd, lim, all_feasible, st);
  snprintf(g_err, sizeof g_err, "unsupported (nx=%d, nu=%d)", p->nx, p->nu);
  return ASLR_E_INVALID;
}

int launch_forward(aslr_problem *p, const SolverDev &sd, hipStream_t st) {
  if (int rc = solver_unsupported(p)) return rc;
  const ModelLimits lim = make_limits(p);
  if (p->nj == 2) return launch_forward_nj2(p->k, p->dam, sd, lim, st);
  if (p->nj == 7) return launch_forward_nj7(p->k, p->dam, sd, lim, st);
  snprintf(g_err, sizeof g_err, "unsupported nj=%d", p->nj);
  return ASLR_E_INVALID;
}

} // namespace

extern "C" {

int aslr_abi_version(void) { return ASLR_ABI_VERSION; }

int64_t aslr_sizeof(int which) {
  switch (which) {
  case 0: return sizeof(aslr_chain_t);
  case 1: return sizeof(aslr_cost_t);
  case 2: return sizeof(aslr_model_t);
  case 3: return sizeof(aslr_problem_desc_t);
  case 4: return sizeof(aslr_solver_params_t);
  case 5: return sizeof(aslr_region_t);
  case 6: return sizeof(aslr_pool_t);
  default: return -1;
  }
}

int32_t aslr_record_len(int32_t nx, int32_t nu) { return rec_len_c(nx, nu); }

void aslr_solver_params_default(aslr_solver_params_t *p, int32_t solver) {
  memset(p, 0, sizeof *p);
  p->solver = solver;
  p->maxiter = 100;
  p->reg_init = NAN;
  p->th_stop = 1e-9;
  p->th_grad = 1e-12;
  p->th_gaptol = 1e-16;
  p->th_stepdec = 0.5;
  p->th_stepinc = 0.01;
  p->th_acceptstep = 0.1;
  p->th_acceptnegstep = 2.0;
  p->reg_min = 1e-9;
  p->reg_max = 1e9;
  p->reg_incfactor = 10.0;
  p->reg_decfactor = 10.0;
  p->boxqp_maxiter = 100;
  p->boxqp_th_acceptstep = 0.1;
  p->boxqp_th_grad = 1e-5;
  p->boxqp_reg = 0.0;
}

int64_t aslr_workspace_bytes(const aslr_problem_desc_t *desc) {
  int nj, nx, nu, dam;
  if (desc_dims(desc, &nj, &nx, &nu, &dam)) return ASLR_E_INVALID;
  aslr_region_t r[ASLR_R_COUNT];
  int64_t total;
  carve(desc, nx, nu, r, &total);
  return total;
}

int aslr_problem_create(const aslr_problem_desc_t *desc, void *workspace, int64_t workspace_bytes, void *stream,
                        aslr_problem_t **out) {
  if (!out) return ASLR_E_INVALID;
  *out = nullptr;
  int nj, nx, nu, dam;
  if (desc_dims(desc, &nj, &nx, &nu, &dam)) { snprintf(g_err, sizeof g_err, "invalid problem description"); return ASLR_E_INVALID; }
  if (!desc->node_model || !desc->x0) return ASLR_E_INVALID;
  for (int t = 0; t <= desc->T; ++t)
    if (desc->node_model[t] < 0 || desc->node_model[t] >= desc->nmodels) return ASLR_E_INVALID;
  if (!(nj == 2 || nj == 7)) {
    snprintf(g_err, sizeof g_err, "unsupported nj=%d: built for nj=2 and nj=7 (SEA / VSA)", nj);
    return ASLR_E_INVALID;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { snprintf(g_err, sizeof g_err, "no HIP device"); return ASLR_E_NODEVICE; }
  aslr_problem *p = new (std::nothrow) aslr_problem();
  if (p) {
    const char *eh = getenv("ASLR_BWD_HS"), *em = getenv("ASLR_BLK_MFMA");
    p->bwd_hs = eh ? atoi(eh) : 0;
    p->blk_mfma = em ? (atoi(em) != 0) : 1;
  }
  if (!p) return ASLR_E_INVALID;
  p->desc = *desc;
  p->desc.node_model = nullptr; p->desc.x0 = nullptr; p->desc.frame_ref = nullptr;
  p->nj = nj; p->nx = nx; p->nu = nu; p->dam = dam; p->rec = rec_len_c(nx, nu);
  p->const_written = false;
  p->const_ok = true; // every cost type but the pendulum cost has a knot-independent diagonal Hessian outside Lqq
  for (int i = 0; i < desc->nmodels; ++i)
    for (int c = 0; c < desc->models[i].ncosts; ++c)
      if (desc->models[i].costs[c].type == ASLR_COST_PENDULUM) p->const_ok = false;
  int64_t total;
  carve(desc, nx, nu, p->regions, &total);
  if (!workspace || workspace_bytes < total || (reinterpret_cast<uintptr_t>(workspace) & 255u)) {
    snprintf(g_err, sizeof g_err, "workspace needs %lld bytes, 256-B aligned (got %lld)", (long long)total, (long long)workspace_bytes);
    delete p;
    return ASLR_E_WORKSPACE;
  }
  p->ws = static_cast<char *>(workspace);
  p->ws_bytes = workspace_bytes;
  hipStream_t st = static_cast<hipStream_t>(stream);
  auto reg = [&](int id) { return p->ws + p->regions[id].offset; };
  // device description (with Binv)
  DevDesc *hd = new DevDesc();
  memset(hd, 0, sizeof *hd);
  hd->chain = desc->chain;
  fill_planar(desc->chain, &hd->planar);
  const int planar_ok = hd->planar.ok;
  fill_planar_reach(desc, hd);
  const int planar_reach = hd->planar.reach_ok;
  for (int i = 0; i < desc->nmodels; ++i) {
    hd->models[i].m = desc->models[i];
    if (!invert(nj, desc->models[i].B, hd->models[i].Binv)) {
      snprintf(g_err, sizeof g_err, "motor inertia B of model %d is singular", i);
      delete hd; delete p;
      return ASLR_E_INVALID;
    }
  }
  hipError_t e = hipMemcpyAsync(reg(ASLR_R_DESC), hd, sizeof(DevDesc), hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = hipMemcpyAsync(reg(ASLR_R_NODE_MODEL), desc->node_model, sizeof(int32_t) * (desc->T + 1), hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = hipMemcpyAsync(reg(ASLR_R_X0), desc->x0, sizeof(double) * desc->B * nx, hipMemcpyHostToDevice, st);
  if (e == hipSuccess && desc->frame_ref) e = hipMemcpyAsync(reg(ASLR_R_FRAME_REF), desc->frame_ref, sizeof(double) * desc->B * 12, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = hipMemsetAsync(reg(ASLR_R_TRAJ_I), 0, p->regions[ASLR_R_TRAJ_I].bytes, st);
  if (e == hipSuccess) e = hipMemsetAsync(reg(ASLR_R_TRAJ_F), 0, p->regions[ASLR_R_TRAJ_F].bytes, st);
  if (e == hipSuccess) e = hipMemsetAsync(reg(ASLR_R_GAPS), 0, p->regions[ASLR_R_GAPS].bytes, st);
  // calc_kernel never rewrites the record chunks that are structurally zero (Lxu, the empty parts of Lxx / Luu)
  if (e == hipSuccess) e = hipMemsetAsync(reg(ASLR_R_DERIV), 0, p->regions[ASLR_R_DERIV].bytes, st);
  if (e == hipSuccess) e = hipMemsetAsync(reg(ASLR_R_KFF), 0, p->regions[ASLR_R_KFF].bytes, st);
  if (e == hipSuccess) e = hipMemsetAsync(reg(ASLR_R_VXXF), 0, p->regions[ASLR_R_VXXF].bytes, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st); // the host staging buffers die below
  delete hd;
  if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void **>(&p->h_done), sizeof(int32_t) * desc->B, 0);
  if (e != hipSuccess) {
    snprintf(g_err, sizeof g_err, "upload failed: %s", hipGetErrorString(e));
    delete p;
    return ASLR_E_HIP;
  }
  KArgs &k = p->k;
  k.desc = reinterpret_cast<const DevDesc *>(reg(ASLR_R_DESC));
  k.node_model = reinterpret_cast<const int32_t *>(reg(ASLR_R_NODE_MODEL));
  k.x0 = reinterpret_cast<const double *>(reg(ASLR_R_X0));
  k.frame_ref = desc->frame_ref ? reinterpret_cast<const double *>(reg(ASLR_R_FRAME_REF)) : nullptr;
  k.xs = (double *)reg(ASLR_R_XS); k.us = (double *)reg(ASLR_R_US); k.xnext = (double *)reg(ASLR_R_XNEXT);
  k.cost = (double *)reg(ASLR_R_COST); k.deriv = (double *)reg(ASLR_R_DERIV); k.gaps = (double *)reg(ASLR_R_GAPS);
  k.kgain = (double *)reg(ASLR_R_KGAIN); k.kff = (double *)reg(ASLR_R_KFF); k.qu = (double *)reg(ASLR_R_QU);
  k.vx = (double *)reg(ASLR_R_VX); k.vxx = (double *)reg(ASLR_R_VXX); k.xs_try = (double *)reg(ASLR_R_XS_TRY);
  k.us_try = (double *)reg(ASLR_R_US_TRY); k.vxxf = (double *)reg(ASLR_R_VXXF);
  k.cost_try = (double *)reg(ASLR_R_COST_TRY);
  k.dyn = (double *)reg(ASLR_R_DYN);
  k.traj_f = (double *)reg(ASLR_R_TRAJ_F); k.traj_i = (int32_t *)reg(ASLR_R_TRAJ_I);
  k.B = desc->B; k.T = desc->T;
  k.b0 = 0; k.b1 = desc->B;
  p->nsub = 1; p->have_sub = false; p->pool_maxiter = 0;
  p->sub_b[0] = 0; p->sub_b[1] = desc->B;
  k.planar = planar_ok;
  k.planar_reach = planar_reach;
  k.iter_log = nullptr; k.log_cap = 0;
  k.seg_t0 = 0; k.seg_t1 = desc->T;
  k.pipeline = getenv("ASLR_PIPELINE") ? atoi(getenv("ASLR_PIPELINE")) : 1; // (read once, when the handle is created)
  *out = p;
  return ASLR_OK;
}

int aslr_problem_destroy(aslr_problem_t *p) {
  if (!p) return ASLR_OK;
  if (p->h_done) (void)hipHostFree(p->h_done);
  if (p->have_ev) for (int i = 0; i < 4; ++i) (void)hipEventDestroy(p->ev[i]);
  if (p->have_sub) {
    for (int i = 0; i < kMaxSub; ++i)
      if (p->sub_stream[i]) { (void)hipStreamDestroy(p->sub_stream[i]); (void)hipEventDestroy(p->sub_join[i]); }
    (void)hipEventDestroy(p->sub_fork);
  }
  delete p;
  return ASLR_OK;
}

int aslr_problem_region(const aslr_problem_t *p, int32_t region_id, aslr_region_t *out) {
  if (!p || !out || region_id < 0 || region_id >= ASLR_R_COUNT) return ASLR_E_INVALID;
  *out = p->regions[region_id];
  return ASLR_OK;
}

int aslr_calc(aslr_problem_t *p, void *stream) {
  if (!p) return ASLR_E_INVALID;
  return launch_calc(p, false, 0, -1.0, static_cast<hipStream_t>(stream));
}

int aslr_calc_diff(aslr_problem_t *p, void *stream) {
  if (!p) return ASLR_E_INVALID;
  return launch_calc(p, true, 0, -1.0, static_cast<hipStream_t>(stream), true);
}

int aslr_backward_pass(aslr_problem_t *p, const aslr_solver_params_t *sp, void *stream) {
  if (!p || !sp) return ASLR_E_INVALID;
  return launch_backward(p, to_dev(sp, 1, 1), static_cast<hipStream_t>(stream));
}

int aslr_forward_pass(aslr_problem_t *p, const aslr_solver_params_t *sp, void *stream) {
  if (!p || !sp) return ASLR_E_INVALID;
  return launch_forward(p, to_dev(sp, 1, 0), static_cast<hipStream_t>(stream));
}

namespace {
// one lock-step iteration of the trajectories [b0, b1) on stream st
int iterate_range(aslr_problem *p, const aslr_solver_params_t *sp, int first, int b0, int b1, hipStream_t st) {
  const KArgs full = p->k;
  p->k.b0 = b0; p->k.b1 = b1; // (the launch helpers read p->k; restored below)
  int rc = ASLR_OK;
  if (first) {
    const double reg0 = std::isnan(sp->reg_init) ? sp->reg_min : sp->reg_init;
    hipLaunchKernelGGL(init_state_kernel, dim3((b1 - b0 + 255) / 256), dim3(256), 0, st, p->k, reg0, sp->is_feasible);
    if (hipGetLastError() != hipSuccess) rc = ASLR_E_HIP;
  }
  SolverDev sd = to_dev(sp, 0, 0);
  sd.maxiter_traj = p->pool_maxiter;
  // (the model-only record chunks are marked written by a sweep that covers the WHOLE shard: the last sub-shard's)
  if (!rc) rc = launch_calc(p, true, kModeCommit | kModeSolver, sp->th_gaptol, st, (first != 0 || p->pool_maxiter > 0) && b1 == p->desc.B && p->const_pending_full);
  if (!rc) rc = launch_backward(p, sd, st);
  if (!rc) rc = launch_forward(p, sd, st);
  p->k = full;
  return rc;
}

// fork: the sub-shard streams wait for what the caller's stream has enqueued so far
int sub_fork(aslr_problem *p, hipStream_t st) {
  HIP_TRY(hipEventRecord(p->sub_fork, st));
  for (int s = 1; s < p->nsub; ++s) HIP_TRY(hipStreamWaitEvent(p->sub_stream[s], p->sub_fork, 0));
  return ASLR_OK;
}
// join: the caller's stream waits for every sub-shard stream
int sub_join(aslr_problem *p, hipStream_t st) {
  for (int s = 1; s < p->nsub; ++s) {
    HIP_TRY(hipEventRecord(p->sub_join[s], p->sub_stream[s]));
    HIP_TRY(hipStreamWaitEvent(st, p->sub_join[s], 0));
  }
  return ASLR_OK;
}
} // namespace

int aslr_set_subshards(aslr_problem_t *p, int32_t n) {
  if (!p || n < 1 || n > kMaxSub) return ASLR_E_INVALID;
  const int B = p->desc.B;
  // boundaries on multiples of 64 trajectories (a calc block; also whole groups of the interleaved candidate slabs)
  const int blocks = (B + 63) / 64;
  if (n > blocks) n = blocks;
  if (n > 1 && !p->have_sub) {
    const char *eb = getenv("ASLR_SUB_BLOCKING");
    const unsigned flags = (eb && atoi(eb)) ? hipStreamDefault : hipStreamNonBlocking;
    for (int i = 0; i < kMaxSub; ++i) { p->sub_stream[i] = nullptr; p->sub_join[i] = nullptr; }
    HIP_TRY(hipEventCreateWithFlags(&p->sub_fork, hipEventDisableTiming));
    p->have_sub = true;
    p->sub_flags = flags;
  }
  for (int i = 1; i < n; ++i) { // (streams are created as they are first needed: unused ones would still take queue slots)
    if (p->sub_stream[i]) continue;
    HIP_TRY(hipStreamCreateWithFlags(&p->sub_stream[i], p->sub_flags));
    HIP_TRY(hipEventCreateWithFlags(&p->sub_join[i], hipEventDisableTiming));
  }
  p->nsub = n;
  for (int s = 0; s <= n; ++s) {
    const long long cut = (long long)blocks * s / n * 64;
    p->sub_b[s] = (int32_t)(cut > B ? B : cut);
  }
  p->sub_b[n] = B;
  return ASLR_OK;
}

int aslr_iterate_n(aslr_problem_t *p, const aslr_solver_params_t *sp, int32_t first, int32_t n, void *stream) {
  if (!p || !sp || n < 0) return ASLR_E_INVALID;
  if (int rc = solver_unsupported(p)) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int B = p->desc.B;
  if (p->pool_maxiter == 0) p->const_pending_full = !p->const_written; // (aslr_solve_pool sets it itself)
  if (p->nsub <= 1) {
    for (int it = 0; it < n; ++it)
      if (int rc = iterate_range(p, sp, first && it == 0, 0, B, st)) return rc;
    return ASLR_OK;
  }
  // Each sub-shard runs its n iterations on its own stream with no synchronisation between sub-shards (trajectories
  // are independent): their kernels interleave freely, the serial sweeps of one under the streaming kernels of the
  // others.  The caller's stream is forked from before and joined after, so for the caller the call is still "n
  // iterations enqueued on my stream".
  if (int rc = sub_fork(p, st)) return rc;
  for (int it = 0; it < n; ++it)
    for (int s = 0; s < p->nsub; ++s)
      if (int rc = iterate_range(p, sp, first && it == 0, p->sub_b[s], p->sub_b[s + 1], s == 0 ? st : p->sub_stream[s])) return rc;
  return sub_join(p, st);
}

int aslr_iterate(aslr_problem_t *p, const aslr_solver_params_t *sp, int32_t first, void *stream) {
  return aslr_iterate_n(p, sp, first, 1, stream);
}

int aslr_iterate_timed(aslr_problem_t *p, const aslr_solver_params_t *sp, int32_t first, void *stream, float *ms3) {
  if (!p || !sp || !ms3) return ASLR_E_INVALID;
  if (int rc = solver_unsupported(p)) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (!p->have_ev) {
    for (int i = 0; i < 4; ++i) HIP_TRY(hipEventCreate(&p->ev[i]));
    p->have_ev = true;
  }
  if (first) {
    const double reg0 = std::isnan(sp->reg_init) ? sp->reg_min : sp->reg_init;
    hipLaunchKernelGGL(init_state_kernel, dim3((p->desc.B + 255) / 256), dim3(256), 0, st, p->k, reg0, sp->is_feasible);
    HIP_TRY(hipGetLastError());
  }
  const SolverDev sd = to_dev(sp, 0, 0); // (whole shard on the caller's stream, whatever aslr_set_subshards says)
  HIP_TRY(hipEventRecord(p->ev[0], st));
  int rc = launch_calc(p, true, kModeCommit | kModeSolver, sp->th_gaptol, st, first != 0); // the first sweep evaluates all
  if (rc) return rc;
  HIP_TRY(hipEventRecord(p->ev[1], st));
  rc = launch_backward(p, sd, st);
  if (rc) return rc;
  HIP_TRY(hipEventRecord(p->ev[2], st));
  rc = launch_forward(p, sd, st);
  if (rc) return rc;
  HIP_TRY(hipEventRecord(p->ev[3], st));
  HIP_TRY(hipEventSynchronize(p->ev[3]));
  for (int i = 0; i < 3; ++i) HIP_TRY(hipEventElapsedTime(&ms3[i], p->ev[i], p->ev[i + 1]));
  return ASLR_OK;
}

int aslr_finalize(aslr_problem_t *p, void *stream) {
  if (!p) return ASLR_E_INVALID;
  hipStream_t st = static_cast<hipStream_t>(stream);
  int rc = launch_calc(p, false, kModeCommit | kModeNoCompute, -1.0, st);
  if (rc) return rc;
  hipLaunchKernelGGL(reset_accepted_kernel, dim3((p->desc.B + 255) / 256), dim3(256), 0, st, p->k);
  HIP_TRY(hipGetLastError());
  return ASLR_OK;
}

int aslr_count_active(aslr_problem_t *p, void *stream, int32_t *active) {
  if (!p || !active) return ASLR_E_INVALID;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int B = p->desc.B;
  HIP_TRY(hipMemcpyAsync(p->h_done, p->k.traj_i + (size_t)ASLR_TI_DONE * B, sizeof(int32_t) * B, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  int n = 0;
  for (int b = 0; b < B; ++b) n += p->h_done[b] ? 0 : 1;
  *active = n;
  return ASLR_OK;
}

int aslr_solve(aslr_problem_t *p, const aslr_solver_params_t *sp, int32_t poll_every, void *stream, int32_t *iters_done) {
  if (!p || !sp) return ASLR_E_INVALID;
  if (sp->maxiter <= 0) { // nothing to iterate: the candidate stays as set (no accepted trial to commit)
    if (iters_done) *iters_done = 0;
    return ASLR_OK;
  }
  // iterations are enqueued in chunks of poll_every (all of them when the host never polls): within a chunk the
  // sub-shard streams run free of each other
  int it = 0;
  const bool polling = !sp->fixed_iterations && poll_every > 0;
  while (it < sp->maxiter) {
    const int chunk = polling ? (poll_every < sp->maxiter - it ? poll_every : sp->maxiter - it) : sp->maxiter - it;
    int rc = aslr_iterate_n(p, sp, it == 0, chunk, stream);
    if (rc) return rc;
    it += chunk;
    if (polling && it < sp->maxiter) {
      int32_t active = 0;
      rc = aslr_count_active(p, stream, &active);
      if (rc) return rc;
      if (active == 0) break;
    }
  }
  if (iters_done) *iters_done = it;
  return aslr_finalize(p, stream);
}

int aslr_solve_pool(aslr_problem_t *p, const aslr_solver_params_t *sp, const aslr_pool_t *pool, int32_t refill_every,
                    int32_t poll_every, void *stream, int32_t *iters_done) {
  if (!p || !sp || !pool || pool->P <= 0 || !pool->x0 || !pool->xs_out || !pool->us_out || !pool->stat_f || !pool->stat_i ||
      !pool->slot_problem || !pool->counters || sp->maxiter <= 0 || sp->fixed_iterations)
    return ASLR_E_INVALID;
  if (pool->frame_ref && !p->k.frame_ref) {
    snprintf(g_err, sizeof g_err, "aslr_solve_pool: per-problem frame references need a problem created with a frame_ref table");
    return ASLR_E_INVALID;
  }
  if (int rc = solver_unsupported(p)) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int B = p->desc.B;
  if (refill_every <= 0) refill_every = 4;
  if (poll_every <= 0) poll_every = 4 * refill_every;
  if (poll_every > sp->maxiter + refill_every) poll_every = sp->maxiter + refill_every; // (a short maxiter must still be polled)
  PoolDev pl;
  pl.P = pool->P; pl.nx = p->nx; pl.nu = p->nu;
  pl.x0 = pool->x0; pl.frame_ref = pool->frame_ref; pl.xs_out = pool->xs_out; pl.us_out = pool->us_out;
  pl.xs_init = pool->xs_init; pl.us_init = pool->us_init;
  pl.stat_f = pool->stat_f; pl.stat_i = pool->stat_i; pl.slot_problem = pool->slot_problem; pl.counters = pool->counters;
  const double reg0 = std::isnan(sp->reg_init) ? sp->reg_min : sp->reg_init;
  // every slot starts idle; the first refill hands out the first B problems
  HIP_TRY(hipMemsetAsync(pool->slot_problem, 0xFF, sizeof(int32_t) * B, st));
  HIP_TRY(hipMemsetAsync(pool->counters, 0, sizeof(int32_t) * 2, st));
  HIP_TRY(hipMemsetAsync(p->k.traj_i + (size_t)ASLR_TI_DONE * B, 0, sizeof(int32_t) * B, st));
  // the refill overwrites the slots' x0 / frame_ref columns with the pool's: keep the handle's own and put them back
  // on exit, so that a later solve / rollout on this handle sees the problems it was created with
  double *save = reinterpret_cast<double *>(p->ws + p->regions[ASLR_R_POOL_SAVE].offset);
  HIP_TRY(hipMemcpyAsync(save, p->k.x0, sizeof(double) * B * p->nx, hipMemcpyDeviceToDevice, st));
  if (p->k.frame_ref) HIP_TRY(hipMemcpyAsync(save + (size_t)B * p->nx, p->k.frame_ref, sizeof(double) * B * 12, hipMemcpyDeviceToDevice, st));
  aslr_solver_params_t spi = *sp;
  const KArgs saved = p->k;
  p->k.iter_log = nullptr; p->k.log_cap = 0; // (a slot's iteration index restarts with every problem)
  // the closed-form reach residual was validated for the references given at create time only (fill_planar_reach):
  // pool targets take the general log map
  if (pool->frame_ref) p->k.planar_reach = 0;
  p->pool_maxiter = sp->maxiter;
  int it = 0, rc = ASLR_OK;
  // the slowest possible schedule: every problem takes maxiter iterations, one wave of B problems after the other
  const long long cap = ((long long)(pool->P + B - 1) / B + 1) * (long long)(sp->maxiter + refill_every);
  bool first = true;
  while (it < cap) {
    hipLaunchKernelGGL(pool_refill_kernel, dim3(B), dim3(64), 0, st, p->k, pl, reg0, sp->is_feasible);
    if (hipGetLastError() != hipSuccess) { rc = ASLR_E_HIP; break; }
    // (the first sweep covers every slot when the pool fills them all: it puts the model-only record chunks in place)
    p->const_pending_full = first && pool->P >= B && !p->const_written;
    rc = aslr_iterate_n(p, &spi, 0, refill_every, stream);
    if (rc) break;
    first = false;
    it += refill_every;
    if (it % poll_every < refill_every) {
      int32_t fin = 0;
      // (flush before counting: the kernel above ran before these iterations)
      hipLaunchKernelGGL(pool_refill_kernel, dim3(B), dim3(64), 0, st, p->k, pl, reg0, sp->is_feasible);
      if (hipGetLastError() != hipSuccess ||
          hipMemcpyAsync(p->h_done, pool->counters + 1, sizeof(int32_t), hipMemcpyDeviceToHost, st) != hipSuccess ||
          hipStreamSynchronize(st) != hipSuccess) { rc = ASLR_E_HIP; break; }
      fin = p->h_done[0];
      if (fin >= pool->P) break;
    }
  }
  int32_t fin_total = -1;
  if (!rc) { // one more flush, then every problem must have been written out
    hipLaunchKernelGGL(pool_refill_kernel, dim3(B), dim3(64), 0, st, p->k, pl, reg0, sp->is_feasible);
    if (hipGetLastError() != hipSuccess ||
        hipMemcpyAsync(p->h_done, pool->counters + 1, sizeof(int32_t), hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess) rc = ASLR_E_HIP;
    else fin_total = p->h_done[0];
  }
  p->pool_maxiter = 0;
  p->k = saved;
  {
    hipError_t e = hipMemcpyAsync(const_cast<double *>(p->k.x0), save, sizeof(double) * B * p->nx, hipMemcpyDeviceToDevice, st);
    if (e == hipSuccess && p->k.frame_ref) e = hipMemcpyAsync(const_cast<double *>(p->k.frame_ref), save + (size_t)B * p->nx, sizeof(double) * B * 12, hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess && !rc) rc = ASLR_E_HIP;
  }
  if (iters_done) *iters_done = it;
  if (rc) return rc;
  if (fin_total < pool->P) {
    snprintf(g_err, sizeof g_err, "aslr_solve_pool: %d of %d problems finished within the iteration bound", fin_total, pool->P);
    return ASLR_E_INVALID;
  }
  hipLaunchKernelGGL(reset_accepted_kernel, dim3((B + 255) / 256), dim3(256), 0, st, p->k);
  HIP_TRY(hipGetLastError());
  return ASLR_OK;
}

int aslr_dam_eval(aslr_problem_t *p, int32_t model_index, int32_t n, const double *x, const double *u, double *xout,
                  double *cost, double *Fx, double *Fu, double *Lx, double *Lu, double *Lxx, double *Lxu, double *Luu,
                  void *stream) {
  if (!p || n <= 0 || model_index < 0 || model_index >= p->desc.nmodels || !x || !u) return ASLR_E_INVALID;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (p->nj == 2) return launch_dam_eval_nj2(p->k, p->dam, model_index, n, x, u, xout, cost, Fx, Fu, Lx, Lu, Lxx, Lxu, Luu, st);
  if (p->nj == 7) return launch_dam_eval_nj7(p->k, p->dam, model_index, n, x, u, xout, cost, Fx, Fu, Lx, Lu, Lxx, Lxu, Luu, st);
  snprintf(g_err, sizeof g_err, "unsupported nj=%d", p->nj);
  return ASLR_E_INVALID;
}

int32_t aslr_residual_len(const aslr_model_t *m, int32_t nj) {
  if (!m || nj <= 0 || nj > ASLR_MAX_NJ || m->ncosts < 0 || m->ncosts > ASLR_MAX_COSTS) return ASLR_E_INVALID;
  int nr = 0;
  for (int c = 0; c < m->ncosts; ++c) {
    switch (m->costs[c].type) {
    case ASLR_COST_FRAME_PLACEMENT: nr += 6; break;
    case ASLR_COST_STATE: nr += 4 * nj; break;
    case ASLR_COST_CONTROL: nr += m->nu; break;
    case ASLR_COST_PENDULUM: nr += 6; break;
    case ASLR_COST_STIFFNESS: nr += m->nu / 2; break;
    default: return ASLR_E_INVALID;
    }
  }
  return nr;
}

int aslr_dam_residuals(aslr_problem_t *p, int32_t model_index, int32_t n, const double *x, const double *u, double *r,
                       void *stream) {
  if (!p || n <= 0 || model_index < 0 || model_index >= p->desc.nmodels || !x || !u || !r) return ASLR_E_INVALID;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int nr = aslr_residual_len(&p->desc.models[model_index], p->nj);
  if (nr <= 0) return nr < 0 ? nr : ASLR_OK;
  if (p->nj == 2) return launch_dam_residuals_nj2(p->k, p->dam, model_index, n, x, u, r, nr, st);
  if (p->nj == 7) return launch_dam_residuals_nj7(p->k, p->dam, model_index, n, x, u, r, nr, st);
  snprintf(g_err, sizeof g_err, "unsupported nj=%d", p->nj);
  return ASLR_E_INVALID;
}

int aslr_frame_placement(aslr_problem_t *p, int32_t frame_joint, const double *frame_R, const double *frame_p, int32_t n,
                         const double *x, int64_t x_stride, double *oMf, void *stream) {
  if (!p || n <= 0 || frame_joint < 0 || frame_joint >= p->nj || !frame_R || !frame_p || !x || !oMf || x_stride < p->nj)
    return ASLR_E_INVALID;
  hipStream_t st = static_cast<hipStream_t>(stream);
  FrameArg F;
  for (int i = 0; i < 9; ++i) F.R[i] = frame_R[i];
  for (int i = 0; i < 3; ++i) F.p[i] = frame_p[i];
  if (p->nj == 2) return launch_frame_placement_nj2(p->k, frame_joint, F, n, x, x_stride, oMf, st);
  if (p->nj == 7) return launch_frame_placement_nj7(p->k, frame_joint, F, n, x, x_stride, oMf, st);
  snprintf(g_err, sizeof g_err, "unsupported nj=%d", p->nj);
  return ASLR_E_INVALID;
}

int aslr_set_iteration_log(aslr_problem_t *p, double *log, int32_t capacity) {
  if (!p || (log && capacity <= 0)) return ASLR_E_INVALID;
  p->k.iter_log = log;
  p->k.log_cap = log ? capacity : 0;
  return ASLR_OK;
}

int aslr_quasi_static(aslr_problem_t *p, int32_t maxiter, double tol, int32_t *iters_dev, void *stream) {
  if (!p || maxiter <= 0) return ASLR_E_INVALID;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (p->nj == 2) return launch_quasi_static_nj2(p->k, p->dam, maxiter, tol, iters_dev, st);
  if (int rc = solver_unsupported(p)) return rc;
  if (p->nj == 7) return launch_quasi_static_nj7(p->k, p->dam, maxiter, tol, iters_dev, st);
  snprintf(g_err, sizeof g_err, "unsupported nj=%d", p->nj);
  return ASLR_E_INVALID;
}

const char *aslr_last_error(void) { return g_err; }

} // extern "C"
