"""The slice of Crocoddyl's Python API that aslr_to's scripts touch, re-hosted on
the MI355X engine: ShootingProblem, SolverDDP / SolverFDDP / SolverBoxDDP, the cost / activation /
residual descriptors and the two callbacks.  Use as

    from aslr_to_amd import crocoddyl

A problem is a *batch* of B shooting problems sharing structure (same robot, same cost types) with
per-trajectory x0 and, optionally, per-trajectory frame-placement targets; `ShootingProblem(x0,
runningModels, terminalModel)` is the B = 1 case and behaves like the single-problem API
(lists of numpy arrays in `solver.xs`, `solver.us`).
"""
import math
import sys

import numpy as np

from . import _abi
from .lowering import lower_problem, shard_rows
from .models import (ActivationModelQuad, ActivationModelWeightedQuad, CostModelResidual,  # noqa: F401
                     CostModelSum, Jcomponent, ResidualModelControl, ResidualModelState)


class _NodePinocchio(object):
    """data.differential.multibody.pinocchio of a node: oMf[frame_id] from the GPU (aslr_frame_placement at the
    node's link positions XS[t]), what the scripts print after a solve (examples/two_dof_sea.py:82-86,
    examples/two_dof_vsa_boxddp.py:83-84).  B = 1: an object with .rotation / .translation; a batch: the same with a
    leading batch dimension (device tensors)."""

    def __init__(self, node):
        self._node = node

    @property
    def oMf(self):
        return _NodeFrames(self._node)


class _NodeFrames(object):
    def __init__(self, node):
        self._node = node

    def __getitem__(self, fid):
        from .engine import _SE3View
        p, t = self._node._p, self._node._t
        e = p.engine
        model = p.runningModels[0].state.pinocchio if p.runningModels else p.terminalModel.state.pinocchio
        fr = model.frames[fid]
        if fr.parent < 0:
            return _SE3View(fr.placement.rotation.copy(), fr.placement.translation.copy())
        x = e.region(_abi.R_XS)[t]  # [B, nx], contiguous
        R, pos = e.frame_placement(fr.parent, fr.placement.rotation, fr.placement.translation, x)
        if p.batch == 1:
            return _SE3View(R[0].cpu().numpy(), pos[0].cpu().numpy())
        return _SE3View(R, pos)

    def __len__(self):
        p = self._node._p
        model = p.runningModels[0].state.pinocchio if p.runningModels else p.terminalModel.state.pinocchio
        return len(model.frames)


class _NodeMultibody(object):
    def __init__(self, node):
        self.pinocchio = _NodePinocchio(node)


class _NodeData(object):
    """runningDatas[t] / terminalData: views of the engine's per-node results."""

    def __init__(self, problem, t):
        self._p, self._t = problem, t
        self.differential = self
        self.multibody = _NodeMultibody(self)
        self.pinocchio = self.multibody.pinocchio

    def _blk(self, name):
        e = self._p.engine
        blk = e.deriv_block(name)[self._t, 0].cpu().numpy()
        if name in ("Fu", "Lu", "Lxu", "Luu"):
            blk = e.cut_u(blk)
        return e.cut_u(blk, axis=-2) if name == "Luu" else blk

    @property
    def r(self):
        """data.r: the stacked cost residuals of this node (integrated_action.py:17-18), trajectory 0."""
        p, t = self._p, self._t
        e = p.engine
        model = p.runningModels[t] if t < p.T else p.terminalModel
        x = e.region(_abi.R_XS)[t, 0].cpu().numpy()
        u = e.region(_abi.R_US)[t, 0].cpu().numpy() if t < p.T else model.differential._default_u()
        mi = int(p.lowered.node_model[t])
        dam = model.differential
        return dam.costs.order_residuals(e.dam_residuals(mi, x, u)[0], dam.state.ndx, dam.nu, dam.nu_dev)

    xnext = property(lambda s: s._p.engine.region(_abi.R_XNEXT)[s._t, 0].cpu().numpy())
    cost = property(lambda s: float(s._p.engine.region(_abi.R_COST)[s._t, 0].item()))
    Fx = property(lambda s: s._blk("Fx"))
    Fu = property(lambda s: s._blk("Fu"))
    Lx = property(lambda s: s._blk("Lx"))
    Lu = property(lambda s: s._blk("Lu"))
    Lxx = property(lambda s: s._blk("Lxx"))
    Lxu = property(lambda s: s._blk("Lxu"))
    Luu = property(lambda s: s._blk("Luu"))


class _DataList(list):
    def tolist(self):
        return list(self)


class ShootingProblem(object):
    """crocoddyl.ShootingProblem(x0, runningModels, terminalModel) (examples/two_dof_sea.py:66).

    Batched form: `ShootingProblem(x0s[B, nx], runningModels, terminalModel, frame_refs=...)`.
    With `rank`/`world_size` the batch is sharded in contiguous blocks, one shard per GPU
    (SURVEY.md 8(e)); every trajectory is solved independently, so results do not depend on the
    sharding.
    """

    def __init__(self, x0, runningModels, terminalModel, frame_refs=None, rank=0, world_size=1, device=None):
        x0 = np.atleast_2d(np.asarray(x0, dtype=np.float64))
        self._x0_all = x0
        self.batch_total = x0.shape[0]
        self.rank, self.world_size = rank, world_size
        lo, hi = shard_rows(self.batch_total, rank, world_size)
        self.rows = (lo, hi)
        self.runningModels = list(runningModels)
        self.terminalModel = terminalModel
        self.T = len(self.runningModels)
        self.nthreads = 1
        fr = None if frame_refs is None else list(frame_refs)[lo:hi]
        self._lowered = lower_problem(x0[lo:hi], self.runningModels, terminalModel, fr)
        self.batch = hi - lo
        self.nx, self.nu = self._lowered.nx, self._lowered.nu_user
        self._device = device
        self._engine = None

    @property
    def x0(self):
        return self._x0_all[self.rows[0]] if self.batch == 1 else self._x0_all[self.rows[0]:self.rows[1]]

    @property
    def lowered(self):
        return self._lowered

    @property
    def engine(self):
        if self._engine is None:
            from .engine import Engine
            self._engine = Engine(self._lowered, self._device)
        return self._engine

    # -- Crocoddyl API --
    def calc(self, xs, us):
        e = self.engine
        e.set_candidate(xs, us)
        e.calc()
        return self._total_cost()

    def calcDiff(self, xs, us):
        e = self.engine
        e.set_candidate(xs, us)
        e.calc_diff()
        return self._total_cost()

    def _total_cost(self):
        c = self.engine.region(_abi.R_COST).sum(dim=0)
        return float(c[0].item()) if self.batch == 1 else c

    def rollout(self, us):
        """xs with xs[0] = x0 and xs[t+1] = xnext(xs[t], us[t]): T sequential calc sweeps are avoided by
        running the forward kernel with zero gains (K = 0, k = 0, alpha = 1)."""
        import torch
        e = self.engine
        # the solver's state the forward kernel reads is saved and put back: rollout() has no side effect on a
        # solve in progress (gains, candidate, feasibility flags)
        rids = (_abi.R_XS, _abi.R_US, _abi.R_KGAIN, _abi.R_KFF)
        keep = [(r, e.region(r).clone()) for r in rids]
        feas = e.region(_abi.R_TRAJ_I)[_abi.TI_FEASIBLE].clone()
        e.set_candidate(None, us)
        e.region(_abi.R_KGAIN).zero_()
        e.region(_abi.R_KFF).zero_()
        e.region(_abi.R_TRAJ_I)[_abi.TI_FEASIBLE].fill_(1)
        e.forward_pass(_abi.default_solver_params(_abi.SOLVER_DDP))
        xs = e.region(_abi.R_XS_TRY)[0].permute(1, 0, 2)
        out = [x for x in xs[0].cpu().numpy()] if self.batch == 1 else xs.clone()
        for r, t in keep:
            e.region(r).copy_(t)
        e.region(_abi.R_TRAJ_I)[_abi.TI_FEASIBLE].copy_(feas)
        torch.cuda.synchronize(e.device)
        return out

    def quasiStatic(self, xs, maxiter=100, tol=1e-9):
        """us[t] holding xs[t] still under node t's model (examples/two_dof_sea.py:78); xs has T entries."""
        import torch
        e = self.engine
        xs = np.asarray(xs, dtype=np.float64) if not torch.is_tensor(xs) else xs
        x = torch.as_tensor(xs, dtype=torch.float64, device=e.device)
        if x.dim() == 2:
            x = x.unsqueeze(0).expand(self.batch, -1, -1)
        X = e.region(_abi.R_XS)
        X[:self.T].copy_(x[:, :self.T].permute(1, 0, 2))
        e.region(_abi.R_US).zero_()
        e.quasi_static(maxiter, tol)
        torch.cuda.synchronize(e.device)
        U = e.cut_u(e.region(_abi.R_US).permute(1, 0, 2))
        return [u for u in U[0].cpu().numpy()] if self.batch == 1 else U.clone()

    @property
    def runningDatas(self):
        return _DataList(_NodeData(self, t) for t in range(self.T))

    @property
    def terminalData(self):
        return _NodeData(self, self.T)


class CallbackLogger(object):
    """crocoddyl.CallbackLogger (examples/double_pendulum.py:77-79, examples/two_dof_sea.py:75): per-iteration
    `costs, u_regs, x_regs, grads, stops, steps, iters` and the solver's latest `xs, us, fs`.
    B = 1: plain Python lists, one entry per iteration, as in Crocoddyl.  A batch: after solve() the same names hold
    numpy arrays [iterations, B] (NaN where a trajectory had already stopped) and `iters` the per-trajectory
    iteration counts; they come from the device-resident log in one transfer."""

    def __init__(self):
        self.xs, self.us, self.fs = [], [], []
        self.costs, self.u_regs, self.x_regs, self.grads, self.stops, self.steps = [], [], [], [], [], []
        self.iters = []

    def __call__(self, solver):
        self.xs, self.us, self.fs = solver.xs, solver.us, solver.fs
        self.iters.append(solver.iter)
        self.costs.append(solver.cost)
        self.u_regs.append(solver.u_reg)
        self.x_regs.append(solver.x_reg)
        self.grads.append(-solver.d[1])   # Crocoddyl: -expectedImprovement()[1]
        self.stops.append(solver.stop)
        self.steps.append(solver.stepLength)

    def from_batch_log(self, solver, log, iters):
        """log: numpy [n, LOG_COUNT, B]; iters: per-trajectory iteration counts."""
        self.xs, self.us, self.fs = solver.xs, solver.us, solver.fs
        self.costs, self.stops = log[:, _abi.LOG_COST], log[:, _abi.LOG_STOP]
        self.x_regs = self.u_regs = log[:, _abi.LOG_XREG]
        self.grads, self.steps = -log[:, _abi.LOG_D2], log[:, _abi.LOG_STEP]
        self.iters = iters


class CallbackVerbose(object):
    """crocoddyl.CallbackVerbose: one table row per iteration -- iter, cost, stop, grad (= -d[1]), xreg, ureg, step,
    feas -- with the header repeated every 10 iterations (the layout of the Crocoddyl 1.x generation the reference
    was written against; unverifiable here).  For a batch one row per lock-step iteration with the
    active-trajectory count, the summed cost and the largest stop / regularisation of the active trajectories."""

    def __init__(self, out=None):
        self.out = out or sys.stdout

    def __call__(self, solver):
        if solver.iter % 10 == 0:
            self.out.write("iter \t cost \t      stop \t    grad \t  xreg \t      ureg \t step \t feas\n")
        self.out.write("%4d  %.5e  %.5e  %.5e  %.5e  %.5e  %.4f     %d\n" % (
            solver.iter, solver.cost, solver.stop, -solver.d[1], solver.x_reg, solver.u_reg, solver.stepLength,
            1 if solver.isFeasible else 0))

    def from_batch_log(self, solver, log, iters):
        for i in range(log.shape[0]):
            on = ~np.isnan(log[i, _abi.LOG_COST])
            if not on.any():
                break
            if i % 10 == 0:
                self.out.write("iter  active   sum cost     max stop     max grad     max xreg   mean step\n")
            self.out.write("%4d  %6d  %.5e  %.5e  %.5e  %.5e  %.4f\n" % (
                i, int(on.sum()), log[i, _abi.LOG_COST][on].sum(), log[i, _abi.LOG_STOP][on].max(),
                (-log[i, _abi.LOG_D2][on]).max(), log[i, _abi.LOG_XREG][on].max(), log[i, _abi.LOG_STEP][on].mean()))


class _IterationView(object):
    """The solver as a callback of iteration i saw it (B = 1), rebuilt from row i of the device log."""

    def __init__(self, solver, row, i):
        self._s = solver
        self.iter = i
        self.cost, self.stop = float(row[_abi.LOG_COST]), float(row[_abi.LOG_STOP])
        self.x_reg = self.u_reg = float(row[_abi.LOG_XREG])
        self.stepLength = float(row[_abi.LOG_STEP])
        self.d = [float(row[_abi.LOG_D1]), float(row[_abi.LOG_D2])]
        self.dV, self.dVexp = float(row[_abi.LOG_DV]), float(row[_abi.LOG_DVEXP])
        self.isFeasible = int(row[_abi.LOG_FEASIBLE])
        self.status = int(row[_abi.LOG_STATUS])

    def __getattr__(self, name):  # xs, us, fs, problem, K ...: the solver's (final) ones
        return getattr(self._s, name)


class SolverDDP(object):
    """crocoddyl.SolverDDP on the batched GPU engine.  All of `x_reg, alpha, feasible, cost, stop,
    iter, status` are per trajectory; scalar properties report trajectory 0 for a
    B = 1 problem and tensors for a batch."""
    _solver = _abi.SOLVER_DDP

    def __init__(self, problem):
        self.problem = problem
        sp = _abi.default_solver_params(self._solver)
        self._sp = sp
        self._callbacks = []
        self.keep_log = False   # record the per-iteration log even without callbacks (iteration_log())
        self.poll_every = 4
        self.batch_iters = 0

    # -- parameters (crocoddyl member names) --
    def _param(name):  # noqa: N805
        return property(lambda s: getattr(s._sp, name), lambda s, v: setattr(s._sp, name, v))

    th_stop = _param("th_stop")
    th_grad = _param("th_grad")
    th_gaptol = _param("th_gaptol")
    th_stepdec = _param("th_stepdec")
    th_stepinc = _param("th_stepinc")
    th_acceptstep = _param("th_acceptstep")
    th_acceptnegstep = _param("th_acceptnegstep")
    reg_min = _param("reg_min")
    reg_max = _param("reg_max")
    reg_incfactor = _param("reg_incfactor")
    reg_decfactor = _param("reg_decfactor")
    del _param

    @property
    def alphas(self):
        return [1.0 / 2 ** j for j in range(_abi.NALPHA)]

    def setCallbacks(self, callbacks):
        """Callbacks are NOT called during the solve: the line-search kernel records what they read (cost, stop, regularisation,
        step length, d, dV, feasibility, status) into a device-resident log and they are replayed from it when solve()
        returns (B = 1: once per iteration with an _IterationView; a batch: `from_batch_log`).  Consequences, different
        from Crocoddyl: (1) inside a callback `xs / us / fs / K / k` are the solver's FINAL values, not those of that
        iteration (a display callback such as examples/double_pendulum.py:60 CallbackDisplay would show the final
        trajectory every time); (2) CallbackVerbose prints after the solve, not while it runs; (3) a callback cannot
        stop the solve.  For per-iteration trajectories drive the solver one iteration at a time: `solve(xs, us, 1,
        isFeasible=...)` in a loop keeps every semantic of the reference at the price of a host round trip per iteration."""
        self._callbacks = list(callbacks)

    def getCallbacks(self):
        return self._callbacks

    # -- solve --
    def solve(self, init_xs=None, init_us=None, maxiter=100, isFeasible=False, regInit=None):
        """solver.solve(init_xs=[], init_us=[], maxiter=100, isFeasible=False, regInit=nan) -> bool
        (all trajectories converged)."""
        import torch
        e = self.problem.engine
        sp = self._sp
        sp.maxiter = int(maxiter)
        sp.is_feasible = 1 if isFeasible else 0
        sp.reg_init = float("nan") if regInit is None else float(regInit)
        e.set_candidate(init_xs, init_us)
        # callbacks: the line-search kernel records what they read into a device-resident log; the solve runs
        # without a host round trip per iteration and the callbacks are replayed from the log afterwards
        e.enable_iteration_log(sp.maxiter if (self._callbacks or self.keep_log) and sp.maxiter > 0 else 0)
        self.batch_iters = e.solve(sp, self.poll_every)
        torch.cuda.synchronize(e.device)
        st = e.traj_i(_abi.TI_STATUS)
        if self._callbacks:
            self._replay_callbacks()
        return bool(((st & _abi.ST_CONVERGED) != 0).all().item())

    def solve_pool(self, x0s, frame_refs=None, maxiter=100, isFeasible=False, regInit=None, refill_every=4,
                   poll_every=16, init_xs=None, init_us=None):
        """Solve MANY problems of this solver's structure (per-problem x0 and, optionally, frame-placement targets as
        pinocchio.SE3 or [P, 12] arrays), each from a cold start to its own stop, streaming them through the problem's
        trajectory slots (the loop `for x0 in x0s: solver.solve([], [], maxiter)` of a script, run on the device:
        aslr_solve_pool).  -> dict of torch tensors: xs [P, T+1, nx], us [P, T, nu], cost, stop, x_reg, step, iters,
        status [P], and batch_iters."""
        sp = self._sp
        sp.maxiter = int(maxiter)
        sp.is_feasible = 1 if isFeasible else 0
        sp.reg_init = float("nan") if regInit is None else float(regInit)
        x0s = np.atleast_2d(np.asarray(x0s, dtype=np.float64))
        fr = None
        if frame_refs is not None:
            fr = np.array([f.as12() if hasattr(f, "as12") else np.asarray(f, dtype=np.float64).reshape(12) for f in frame_refs])
        return self.problem.engine.solve_pool(x0s, fr, sp, refill_every, poll_every, init_xs, init_us)

    def iteration_log(self):
        """numpy [iterations, LOG_COUNT, B] of the last solve (needs callbacks or `keep_log = True`), trimmed to the
        iterations some trajectory ran; NaN where a trajectory had stopped.  Fields: _abi.LOG_*."""
        lg = self.problem.engine.iteration_log()
        if lg is None:
            return None
        n = int(self.problem.engine.traj_i(_abi.TI_ITER).max().item())
        return lg[:min(n, lg.shape[0])].cpu().numpy()

    def _replay_callbacks(self):
        log = self.iteration_log()
        iters = self.problem.engine.traj_i(_abi.TI_ITER).cpu().numpy()
        if self._single():
            for i in range(log.shape[0]):
                row = log[i, :, 0]
                # Crocoddyl returns from solve() before the callbacks when the regularisation hits its maximum
                if int(row[_abi.LOG_STATUS]) & _abi.ST_REG_MAX:
                    break
                view = _IterationView(self, row, i)
                for cb in self._callbacks:
                    cb(view)
        else:
            for cb in self._callbacks:
                if hasattr(cb, "from_batch_log"):
                    cb.from_batch_log(self, log, iters)
                else:
                    raise TypeError("callback %r cannot consume a batch log (needs from_batch_log)" % (cb,))

    def export_solution(self, path, trajectory=0):
        """The arrays examples/two_dof_vsa_boxddp.py:104-127 saves to .mat files, as one .npz: t [T], q [T+1, nj]
        (link positions), u [T, nj] (motor commands), stiffness [T, nj] (VSA models; empty otherwise), xs, us.
        trajectory: index in the batch, or None for all ([B, ...])."""
        e = self.problem.engine
        X = self.xs if not self._single() else np.asarray(self.xs)[None]
        U = self.us if not self._single() else np.asarray(self.us)[None]
        X = X.cpu().numpy() if hasattr(X, "cpu") else np.asarray(X)
        U = U.cpu().numpy() if hasattr(U, "cpu") else np.asarray(U)
        nj = e.nx // 4
        dt = float(self.problem.runningModels[0].dt)
        sel = slice(None) if trajectory is None else int(trajectory)
        vsa = e.nu_user == 2 * nj
        np.savez(path, t=np.arange(self.problem.T) * dt, q=X[sel][..., :nj], u=U[sel][..., :nj],
                 stiffness=U[sel][..., nj:] if vsa else np.zeros(U[sel].shape[:-1] + (0,)), xs=X[sel], us=U[sel])
        return path

    # -- results --
    def _single(self):
        return self.problem.batch == 1

    def _current_xu(self):
        """xs/us including a not-yet-committed accepted candidate (only matters inside callbacks)."""
        import torch
        e = self.problem.engine
        acc = e.traj_i(_abi.TI_ACCEPTED)
        X, U = e.region(_abi.R_XS), e.region(_abi.R_US)
        if bool((acc >= 0).any().item()):
            X, U = X.clone(), U.clone()
            XT, UT = e.region(_abi.R_XS_TRY), e.region(_abi.R_US_TRY)
            for b in torch.nonzero(acc >= 0).flatten().tolist():
                a = int(acc[b].item())
                X[:, b] = XT[a, :, b]
                U[:, b] = UT[a, :, b]
        return X.permute(1, 0, 2), U.permute(1, 0, 2)

    @property
    def xs(self):
        X, _ = self._current_xu()
        return [x for x in X[0].cpu().numpy()] if self._single() else X

    @property
    def us(self):
        _, U = self._current_xu()
        U = self.problem.engine.cut_u(U)
        return [u for u in U[0].cpu().numpy()] if self._single() else U

    def _tf(self, row):
        v = self.problem.engine.traj_f(row)
        return float(v[0].item()) if self._single() else v

    def _ti(self, row):
        v = self.problem.engine.traj_i(row)
        return int(v[0].item()) if self._single() else v

    cost = property(lambda s: s._tf(_abi.TF_COST))
    stop = property(lambda s: s._tf(_abi.TF_STOP))
    x_reg = property(lambda s: s._tf(_abi.TF_XREG))
    u_reg = property(lambda s: s._tf(_abi.TF_XREG))
    stepLength = property(lambda s: s._tf(_abi.TF_STEP))
    dV = property(lambda s: s._tf(_abi.TF_DV))
    dVexp = property(lambda s: s._tf(_abi.TF_DVEXP))
    d = property(lambda s: [s._tf(_abi.TF_D1), s._tf(_abi.TF_D2)])
    status = property(lambda s: s._ti(_abi.TI_STATUS))
    isFeasible = property(lambda s: s._ti(_abi.TI_FEASIBLE))

    @property
    def iter(self):
        """Crocoddyl's iter_: index of the last iteration when solve() returned from inside the loop
        (converged / regularisation at its maximum), maxiter otherwise."""
        n, st = self._ti(_abi.TI_ITER), self._ti(_abi.TI_STATUS)
        inside = (st & (_abi.ST_CONVERGED | _abi.ST_REG_MAX)) != 0
        if self._single():
            return n - 1 if inside and n > 0 else n
        return n - (inside & (n > 0)).to(n.dtype)

    @property
    def iterations(self):
        """number of completed iterations per trajectory"""
        return self._ti(_abi.TI_ITER)

    def _gain(self, rid):
        import torch
        e = self.problem.engine
        torch.cuda.synchronize(e.device)
        v = e.region(rid)
        if rid in (_abi.R_KGAIN, _abi.R_KFF, _abi.R_QU):  # [T, B, nu(, nx)]: the models' own controls
            v = e.cut_u(v, axis=2)
        return [g for g in v[:, 0].cpu().numpy()] if self._single() else v.transpose(0, 1)

    def _value(self, rid):
        """Vx / Vxx are not kept by the solve loop (212 MB per 4096-trajectory shard, written for nobody): they are
        recomputed here by one stand-alone backward pass on the CURRENT iterate and regularisation (Crocoddyl
        holds those of its last iteration's backward pass, i.e. one step earlier); K, k, Qu are left untouched."""
        import torch
        e = self.problem.engine
        keep = [(r, e.region(r).clone()) for r in (_abi.R_KGAIN, _abi.R_KFF, _abi.R_QU)]
        e.calc_diff()
        e.backward_pass(self._sp)
        torch.cuda.synchronize(e.device)
        v = e.region(rid).clone()
        for r, t in keep:
            e.region(r).copy_(t)
        return [g for g in v[:, 0].cpu().numpy()] if self._single() else v.transpose(0, 1)

    Vx = property(lambda s: s._value(_abi.R_VX))
    Vxx = property(lambda s: s._value(_abi.R_VXX))
    K = property(lambda s: s._gain(_abi.R_KGAIN))
    k = property(lambda s: s._gain(_abi.R_KFF))
    Qu = property(lambda s: s._gain(_abi.R_QU))
    fs = property(lambda s: s._gain(_abi.R_GAPS))


class SolverFDDP(SolverDDP):
    """crocoddyl.SolverFDDP (examples/two_dof_sea.py:69)."""
    _solver = _abi.SOLVER_FDDP


class SolverBoxDDP(SolverDDP):
    """crocoddyl.SolverBoxDDP (examples/two_dof_vsa_boxddp.py:69)."""
    _solver = _abi.SOLVER_BOXDDP


def reduce_stats(solver, group=None):
    """Global termination / reporting scalars over all shards; see dist.py."""
    from .dist import all_reduce_stats, local_stats
    return all_reduce_stats(local_stats(solver.problem.engine), group)
