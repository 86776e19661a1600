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


class _NodeData(object):
    """runningDatas[t] / terminalData: views of the engine's per-node results (B = 1)."""

    def __init__(self, problem, t):
        self._p, self._t = problem, t
        self.differential = self

    def _blk(self, name):
        return self._p.engine.deriv_block(name)[self._t, 0].cpu().numpy()

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
        self.nx, self.nu = self._lowered.nx, self._lowered.nu
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
        e.set_candidate(None, us)
        e.region(_abi.R_KGAIN).zero_()
        e.region(_abi.R_KFF).zero_()
        e.region(_abi.R_TRAJ_I)[_abi.TI_FEASIBLE].fill_(1)
        e.forward_pass(_abi.default_solver_params(_abi.SOLVER_DDP))
        xs = e.region(_abi.R_XS_TRY)[0].permute(1, 0, 2)
        torch.cuda.synchronize(e.device)
        return [x for x in xs[0].cpu().numpy()] if self.batch == 1 else xs.clone()

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
        U = e.region(_abi.R_US).permute(1, 0, 2)
        return [u for u in U[0].cpu().numpy()] if self.batch == 1 else U.clone()

    @property
    def runningDatas(self):
        return _DataList(_NodeData(self, t) for t in range(self.T))

    @property
    def terminalData(self):
        return _NodeData(self, self.T)


class CallbackLogger(object):
    """Records per-iteration scalars and the final xs/us (examples/double_pendulum.py:77-79)."""

    def __init__(self):
        self.xs, self.us = [], []
        self.costs, self.u_regs, self.x_regs, self.grads, self.stops, self.steps = [], [], [], [], [], []
        self.iters = []

    def __call__(self, solver):
        self.xs, self.us = solver.xs, solver.us
        self.iters.append(solver.iter)
        self.costs.append(solver.cost)
        self.u_regs.append(solver.u_reg)
        self.x_regs.append(solver.x_reg)
        self.grads.append(solver.d[0])
        self.stops.append(solver.stop)
        self.steps.append(solver.stepLength)


class CallbackVerbose(object):
    def __init__(self, out=None):
        self.out = out or sys.stdout
        self._n = 0

    def __call__(self, solver):
        if self._n % 10 == 0:
            self.out.write("iter     cost         stop         grad         xreg         ureg       step    ||ffeas||\n")
        self._n += 1
        self.out.write("%4d  %0.5e  %0.5e  %0.5e  %0.5e  %0.5e  %0.4f  %d\n" % (
            solver.iter, solver.cost, solver.stop, solver.d[0], solver.x_reg, solver.u_reg, solver.stepLength,
            1 if solver.isFeasible else 0))


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
        if self._callbacks and self.problem.batch == 1:
            # per-iteration callbacks need a host round trip per iteration
            done_iters = sp.maxiter
            for it in range(sp.maxiter):
                e.iterate(sp, it == 0)
                active = e.count_active()
                for cb in self._callbacks:
                    cb(self)
                if active == 0 and not sp.fixed_iterations:
                    done_iters = it + 1
                    break
            it = done_iters
            e.finalize()
            self.batch_iters = it
        else:
            self.batch_iters = e.solve(sp, self.poll_every)
        torch.cuda.synchronize(e.device)
        st = e.traj_i(_abi.TI_STATUS)
        return bool(((st & _abi.ST_CONVERGED) != 0).all().item())

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
