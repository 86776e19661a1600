"""aslr_to's model classes (state, actuation, residuals, costs, SEA/VSA differential models, Euler
integrator) as *descriptions* that lower to the POD structs of include/aslr_to_amd.h.

Names, constructor signatures and attributes mirror the reference (python/aslr_to/*.py) and the
Crocoddyl classes its examples use, so a script written against `aslr_to` + `crocoddyl` runs with

    import aslr_to_amd as aslr_to
    from aslr_to_amd import crocoddyl, pinocchio, example_robot_data

The arithmetic itself (calc / calcDiff) is NOT here: `model.calc(data, x, u)` evaluates on the GPU
through the C ABI (engine.py).  There is no CPU fallback.
"""
import numpy as np

from . import _abi
from .pinocchio import SE3


# --------------------------------------------------------------------------------------------
# state (python/aslr_to/statemultibody_aslr.py:13-109); revolute chains => vector space
# --------------------------------------------------------------------------------------------
class Jcomponent(object):
    both, first, second = 0, 1, 2


class StateMultibodyASR(object):
    """x = [q_l, q_m, v_l, v_m] (statemultibody_aslr.py:7-11)."""

    def __init__(self, pinocchioModel):
        self.pinocchio = pinocchioModel
        self.nx = 2 * (pinocchioModel.nq + pinocchioModel.nv)
        self.ndx = 4 * pinocchioModel.nv
        self.nv = self.ndx // 2
        self.nq = self.nx - self.nv

    def zero(self):
        return np.zeros(self.nx)

    def rand(self):
        n = self.pinocchio.nq
        return np.concatenate([np.random.uniform(-np.pi, np.pi, n), np.random.uniform(-np.pi, np.pi, n),
                               np.random.rand(n), np.random.rand(n)])

    def diff(self, x0, x1):
        return np.asarray(x1, dtype=float) - np.asarray(x0, dtype=float)

    def integrate(self, x, dx):
        return np.asarray(x, dtype=float) + np.asarray(dx, dtype=float)

    def Jdiff(self, x1, x2, firstsecond=Jcomponent.both):
        if firstsecond == Jcomponent.both:
            return [self.Jdiff(x1, x2, Jcomponent.first), self.Jdiff(x1, x2, Jcomponent.second)]
        return -np.eye(self.ndx) if firstsecond == Jcomponent.first else np.eye(self.ndx)

    def Jintegrate(self, x, dx, firstsecond=Jcomponent.both):
        if firstsecond == Jcomponent.both:
            return [np.eye(self.ndx), np.eye(self.ndx)]
        return np.eye(self.ndx)


# --------------------------------------------------------------------------------------------
# actuation (actuation_asr.py:5-13, actuation_vsa.py:5-13, __init__.py:262-290)
# --------------------------------------------------------------------------------------------
class ASRActuation(object):
    def __init__(self, state):
        self.state = state
        self.nu = state.nv // 2

    def motor_matrix(self):
        return np.eye(self.nu)


class VSAASRActuation(ASRActuation):
    pass


class ActuationModelDoublePendulum(object):
    def __init__(self, state, actLink, nu=None):
        if nu is None:
            # examples/double_pendulum.py:18 omits nu and raises TypeError in the reference; the
            # 2-vector control weights at :30 imply nu = 2
            nu = state.nv // 2
        self.state = state
        self.nu = int(nu)
        self.nv = state.nv
        self.actLink = actLink

    def motor_matrix(self):
        nj = self.nv // 2
        S = np.zeros((self.nv, self.nu))
        if self.actLink == 1:
            S[-1, -1] = 1.0
        else:
            S[nj, 0] = 1.0
        return S[nj:, :]


# --------------------------------------------------------------------------------------------
# activations / residuals / costs
# --------------------------------------------------------------------------------------------
class ActivationModelQuad(object):
    def __init__(self, nr):
        self.nr = int(nr)
        self.weights = np.ones(self.nr)


class ActivationModelWeightedQuad(object):
    def __init__(self, weights):
        self.weights = np.array(weights, dtype=float)
        self.nr = self.weights.size


class ResidualModelState(object):
    def __init__(self, state, xref=None, nu=None):
        if xref is not None and np.isscalar(xref):  # (state, nu)
            xref, nu = None, xref
        self.state = state
        self.xref = state.zero() if xref is None else np.array(xref, dtype=float)
        self.nu = nu
        self.nr = state.ndx


class ResidualModelControl(object):
    def __init__(self, state, uref_or_nu=None):
        self.state = state
        if uref_or_nu is None or np.isscalar(uref_or_nu):
            self.nu = None if uref_or_nu is None else int(uref_or_nu)
            self.uref = None
        else:
            self.uref = np.array(uref_or_nu, dtype=float)
            self.nu = self.uref.size
        self.nr = self.nu


class ResidualModelFramePlacementASR(object):
    """residual_frame_placement.py:7-24: r = log6(placement^-1 * oMf[frame_id])."""

    def __init__(self, state, frame_id=None, placement=None, nu=None):
        self.state = state
        self._frame_id = frame_id
        self._placement = placement if placement is not None else SE3()
        self.nu = nu
        self.nr = 6


class CostModelResidual(object):
    def __init__(self, state, activation_or_residual, residual=None):
        self.state = state
        if residual is None:
            self.residual = activation_or_residual
            self.activation = ActivationModelQuad(self.residual.nr)
        else:
            self.activation = activation_or_residual
            self.residual = residual
        self.nr = self.residual.nr


class CostModelDoublePendulum(object):
    """__init__.py:223-259"""

    def __init__(self, state, activation, nu):
        self.state = state
        self.activation = activation if activation is not None else ActivationModelQuad(6)
        self.nu = nu
        self.nr = 6


class CostModelStiffness(object):
    """stiffness_cost.py:6-22: cost = sum(lamda * (u[nu/2:] - Kref))."""

    def __init__(self, state, nu, lamda, Kref=None):
        self.state = state
        self.nu_ = int(nu)
        self.nu = int(nu)
        self.lamda = float(lamda)
        self.Kref = np.zeros(self.nu_ // 2) if Kref is None else np.array(Kref, dtype=float)
        self.nr = self.nu_ // 2


class _CostItem(object):
    def __init__(self, name, cost, weight):
        self.name, self.cost, self.weight, self.active = name, cost, float(weight), True


class CostModelSum(object):
    def __init__(self, state, nu=None):
        self.state = state
        self.nu = nu
        self.costs = {}
        self._order = []

    def addCost(self, name, cost, weight, active=True):
        item = _CostItem(name, cost, weight)
        item.active = active
        self.costs[name] = item
        self._order.append(name)

    def removeCost(self, name):
        del self.costs[name]
        self._order.remove(name)

    @property
    def nr(self):
        return sum((self.costs[n].cost.nr or 0) for n in self._order if self.costs[n].active)

    def order_residuals(self, r, nx, nu, nu_dev=None):
        """`r` holds the residual vectors of the active costs stacked in insertion order (the order of the lowered
        cost list, aslr_dam_residuals); returns them stacked the way Crocoddyl's CostModelSum does: its cost items
        live in a std::map keyed by name, so data.r follows the ALPHABETICAL order of the cost names.
        nu_dev > nu: the device works on controls padded to nu_dev (see _DifferentialBase.lower); the padded entries of
        a control residual are dropped here."""
        seg, off = {}, 0
        pad = 0 if nu_dev is None else int(nu_dev) - int(nu)
        for name in self._order:
            item = self.costs[name]
            if not item.active:
                continue
            n = int(item.cost.nr or 0)
            seg[name] = r[off:off + n]
            off += n
            if pad and isinstance(item.cost, CostModelResidual) and isinstance(item.cost.residual, ResidualModelControl):
                off += pad
        if off != len(r):
            raise ValueError("residual vector has %d entries, the cost stack %d" % (len(r), off))
        return np.concatenate([seg[k] for k in sorted(seg)]) if seg else np.zeros(0)

    def lower(self, nj, nx, nu, nu_dev=None):
        """-> list of _abi.Cost in insertion order (nu_dev: the padded control size of the device, default nu)."""
        out = []
        for name in self._order:
            item = self.costs[name]
            if not item.active:
                continue
            out.append(_lower_cost(item.cost, item.weight, nj, nx, nu, nu if nu_dev is None else nu_dev))
        if len(out) > _abi.MAX_COSTS:
            raise ValueError("at most %d cost terms per CostModelSum are supported" % _abi.MAX_COSTS)
        return out


def _lower_cost(cost, weight, nj, nx, nu, nu_dev=None):
    nu_dev = nu if nu_dev is None else nu_dev
    c = _abi.Cost()
    c.weight = weight
    if isinstance(cost, CostModelResidual):
        res, act = cost.residual, cost.activation
        w = np.asarray(act.weights, dtype=float)
        if isinstance(res, ResidualModelFramePlacementASR):
            if w.size != 6:
                raise ValueError("frame-placement activation needs 6 weights")
            model = res.state.pinocchio
            if res._frame_id is None or not (0 <= res._frame_id < len(model.frames)):
                raise ValueError("unknown frame id %r" % (res._frame_id,))
            frame = model.frames[res._frame_id]
            if frame.parent < 0:
                raise ValueError("frame %r is attached to the universe" % frame.name)
            c.type = _abi.COST_FRAME_PLACEMENT
            c.frame_joint = frame.parent
            ref = res._placement.as12()
            for k in range(12):
                c.ref[k] = ref[k]
            fR = frame.placement.rotation.reshape(9)
            for k in range(9):
                c.frame_R[k] = fR[k]
            for k in range(3):
                c.frame_p[k] = frame.placement.translation[k]
            for k in range(6):
                c.act_w[k] = w[k]
        elif isinstance(res, ResidualModelState):
            if w.size != nx or res.xref.size != nx:
                raise ValueError("state cost: expected %d weights / xref entries" % nx)
            c.type = _abi.COST_STATE
            for k in range(nx):
                c.act_w[k] = w[k]
                c.ref[k] = res.xref[k]
        elif isinstance(res, ResidualModelControl):
            uref = np.zeros(nu) if res.uref is None else res.uref
            if w.size != nu or uref.size != nu:
                raise ValueError("control cost: expected %d weights / uref entries (got %d)" % (nu, w.size))
            c.type = _abi.COST_CONTROL
            for k in range(nu):
                c.act_w[k] = w[k]
                c.ref[k] = uref[k]
            for k in range(nu, nu_dev):  # padded controls stay at zero: unit weight, zero reference (their own pivot in Quu)
                c.act_w[k] = 1.0
                c.ref[k] = 0.0
        else:
            raise TypeError("unsupported residual model %r" % type(res).__name__)
    elif isinstance(cost, CostModelDoublePendulum):
        w = np.asarray(cost.activation.weights, dtype=float)
        if w.size != 6:
            raise ValueError("CostModelDoublePendulum needs 6 activation weights")
        c.type = _abi.COST_PENDULUM
        for k in range(6):
            c.act_w[k] = w[k]
    elif isinstance(cost, CostModelStiffness):
        if cost.Kref.size != nu // 2:
            raise ValueError("CostModelStiffness: Kref needs nu/2 entries")
        c.type = _abi.COST_STIFFNESS
        c.lambda_ = cost.lamda
        for k in range(nu // 2):
            c.ref[k] = cost.Kref[k]
    else:
        raise TypeError("unsupported cost model %r" % type(cost).__name__)
    return c


# --------------------------------------------------------------------------------------------
# differential action models (free_fwddyn_asr.py:6-19, free_fwddyn_vsa.py:6-18) and data
# --------------------------------------------------------------------------------------------
class _PinocchioDataView(object):
    def __init__(self):
        self.oMf = {}
        self.M = None
        self.nle = None


class _Multibody(object):
    def __init__(self):
        self.pinocchio = _PinocchioDataView()


class DifferentialActionData(object):
    def __init__(self, model):
        nv, nx, nu = model.state.nv, model.state.ndx, model.nu
        self.xout = np.zeros(nv)
        self.cost = 0.0
        self.r = np.zeros(max(model.nr, 0))
        self.Fx = np.zeros((nv, nx))
        self.Fu = np.zeros((nv, nu))
        self.Lx = np.zeros(nx)
        self.Lu = np.zeros(nu)
        self.Lxx = np.zeros((nx, nx))
        self.Lxu = np.zeros((nx, nu))
        self.Luu = np.zeros((nu, nu))
        self.multibody = _Multibody()
        self.pinocchio = self.multibody.pinocchio


class _DifferentialBase(object):
    dam = None

    def __init__(self, state, actuationModel, costModel, nu):
        self.state = state
        self.actuation = actuationModel
        self.costs = costModel
        self.nu = int(nu)
        self.nr = costModel.nr
        self._evaluator = None

    def createData(self):
        return DifferentialActionData(self)

    def _default_u(self):
        raise NotImplementedError

    def lower(self, dt=0.0, u_lb=None, u_ub=None):
        """-> _abi.Model for IntegratedActionModelEulerASR(self, dt)."""
        nj = self.state.pinocchio.nv
        nx, nu = self.state.ndx, self.nu
        # ActuationModelDoublePendulum(state, actLink, nu=1) (python/aslr_to/__init__.py:262-290): fewer motor commands
        # than joints.  The kernels are built for nu = nj, so the model is lowered with the controls PADDED to nj: zero
        # columns in S (the padded commands drive nothing: zero Fu columns, zero gradient, zero gains), unit weight in a
        # control cost so that they keep a pivot of their own in Quu, unit box.  The Python layer pads / slices at its
        # boundary (Engine.set_candidate, solver.us / K / k / Qu, data.Fu ...).
        nu_dev = self.nu_dev
        m = _abi.Model()
        m.dam = self.dam
        m.nu = nu_dev
        m.dt = float(dt)
        K = np.zeros((nj, nj)) if self.dam == _abi.DAM_VSA else np.asarray(self.K, dtype=float).reshape(nj, nj)
        Bm = np.asarray(self.B, dtype=float).reshape(nj, nj)
        for k, v in enumerate(K.reshape(-1)):
            m.K[k] = v
        for k, v in enumerate(Bm.reshape(-1)):
            m.B[k] = v
        if self.dam == _abi.DAM_SEA:
            if nu > nj or nu < 1:
                raise ValueError("SEA models need 1 <= nu <= number of joints (got nu=%d, nj=%d)" % (nu, nj))
            S = np.zeros((nj, nu_dev))
            S[:, :nu] = np.asarray(self.actuation.motor_matrix(), dtype=float).reshape(nj, nu)
            for k, v in enumerate(S.reshape(-1)):
                m.S[k] = v
        costs = self.costs.lower(nj, nx, nu, nu_dev)
        m.ncosts = len(costs)
        for i, c in enumerate(costs):
            m.costs[i] = c
        has_lim = u_lb is not None and u_ub is not None and np.all(np.isfinite(u_lb)) and np.all(np.isfinite(u_ub))
        m.has_u_limits = 1 if has_lim else 0
        if has_lim:
            for k in range(nu):
                m.u_lb[k] = float(u_lb[k])
                m.u_ub[k] = float(u_ub[k])
            for k in range(nu, nu_dev):
                m.u_lb[k], m.u_ub[k] = -1.0, 1.0
        return m

    @property
    def nu_dev(self):
        """control size of the lowered model (SEA: one command per joint, padded when the actuation has fewer)"""
        if self.dam == _abi.DAM_SEA:
            return max(self.nu, self.state.pinocchio.nv)
        return self.nu

    # -- single-point evaluation on the GPU (the path the reference's unit tests exercise) --
    def calc(self, data, x, u=None):
        from .engine import point_evaluator
        if u is None:
            u = self._default_u()
        point_evaluator(self).dam(data, x, u, diff=False)

    def calcDiff(self, data, x, u=None):
        from .engine import point_evaluator
        if u is None:
            u = self._default_u()
        point_evaluator(self).dam(data, x, u, diff=True)


class DifferentialFreeASRFwdDynamicsModel(_DifferentialBase):
    """SEA free forward dynamics (free_fwddyn_asr.py:6-19)."""
    dam = _abi.DAM_SEA

    def __init__(self, state, actuationModel, costModel, K=None, B=None):
        _DifferentialBase.__init__(self, state, actuationModel, costModel, actuationModel.nu)
        n = state.nv // 2
        self.enable_force = True
        self.K = 1e-1 * np.eye(n) if K is None else np.array(K, dtype=float)
        self.B = 1e-3 * np.eye(n) if B is None else np.array(B, dtype=float)

    def _default_u(self):
        return np.zeros(self.nu)


class DifferentialFreeFwdDynamicsModelVSA(_DifferentialBase):
    """VSA free forward dynamics, stiffness is a control (free_fwddyn_vsa.py:6-18)."""
    dam = _abi.DAM_VSA

    def __init__(self, state, actuationModel, costModel, B=None):
        _DifferentialBase.__init__(self, state, actuationModel, costModel, 2 * actuationModel.nu)
        n = state.nv // 2
        self.B = 1e-3 * np.eye(n) if B is None else np.array(B, dtype=float)

    def _default_u(self):  # free_fwddyn_vsa.py:21-23
        u = np.zeros(self.nu)
        u[self.nu // 2:] = 3.0
        return u


# --------------------------------------------------------------------------------------------
# Euler integrator (integrated_action.py:6-52)
# --------------------------------------------------------------------------------------------
class IntegratedActionData(object):
    def __init__(self, model):
        nx, nu = model.state.ndx, model.nu
        self.differential = model.differential.createData()
        self.xnext = np.zeros(model.state.nx)
        self.dx = np.zeros(nx)
        self.cost = 0.0
        self.r = np.zeros(max(model.nr, 0))
        self.Fx = np.zeros((nx, nx))
        self.Fu = np.zeros((nx, nu))
        self.Lx = np.zeros(nx)
        self.Lu = np.zeros(nu)
        self.Lxx = np.zeros((nx, nx))
        self.Lxu = np.zeros((nx, nu))
        self.Luu = np.zeros((nu, nu))


class IntegratedActionModelEulerASR(object):
    def __init__(self, diffModel, timeStep=1e-3, withCostResiduals=True):
        self.differential = diffModel
        self.state = diffModel.state
        self.nu = diffModel.nu
        self.nr = diffModel.nr
        self.withCostResiduals = withCostResiduals
        self.dt = float(timeStep)
        self.u_lb = np.full(self.nu, -np.inf)
        self.u_ub = np.full(self.nu, np.inf)
        self._evaluator = None

    @property
    def has_control_limits(self):
        return bool(np.all(np.isfinite(self.u_lb)) and np.all(np.isfinite(self.u_ub)))

    def createData(self):
        return IntegratedActionData(self)

    def lower(self):
        return self.differential.lower(self.dt, np.asarray(self.u_lb, dtype=float), np.asarray(self.u_ub, dtype=float))

    def calc(self, data, x, u=None):
        from .engine import point_evaluator
        point_evaluator(self).integrated(data, x, u, diff=False)
        return data.xnext, data.cost

    def calcDiff(self, data, x, u=None):
        from .engine import point_evaluator
        point_evaluator(self).integrated(data, x, u, diff=True)


def u_squared(log):
    """Sum over knots of the squared controls, per control component (__init__.py:63-68)."""
    us = np.asarray(log.us, dtype=float)
    return (us ** 2).sum(axis=0)
