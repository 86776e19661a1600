"""The BASELINE.json configurations as concrete, seeded, synthetic batches.

Every scenario is a row of the SPECS table below -- robot, actuator kind, cost stack, actuator constants, step and
bounds, each value cited to the reference script it comes from -- turned into action models by one builder
(`_build`).  A scenario is returned as a dict: x0 [B, nx], running (list of T models), terminal, frame_refs [B, 12] or
None, solver name, maxiter, th_stop, name.  Pure host logic.
"""
import numpy as np

from . import example_robot_data
from .crocoddyl import (ActivationModelWeightedQuad, CostModelResidual, CostModelSum, ResidualModelControl,
                        ResidualModelState)
from .models import (ASRActuation, ActuationModelDoublePendulum, CostModelDoublePendulum, CostModelStiffness,
                     DifferentialFreeASRFwdDynamicsModel, DifferentialFreeFwdDynamicsModelVSA,
                     IntegratedActionModelEulerASR, ResidualModelFramePlacementASR, StateMultibodyASR,
                     VSAASRActuation)
from .pinocchio import SE3

# Cost entries: (name, kind, weight, parameters).  Kinds:
#   "reach"    frame-placement residual to SE3(I, target) of frame `frame`, quadratic activation
#   "state"    weighted-quadratic state regulariser around 0, weights per block [q_l, q_m, v_l, v_m]
#   "control"  weighted-quadratic control regulariser, explicit weights (None: plain quadratic)
#   "pendulum" CostModelDoublePendulum with the given activation weights
#   "stiffness" CostModelStiffness(lamda, Kref value)
SPECS = {
    # examples/two_dof_vsa_boxddp.py: gravity :16, target :22-23, weights :29-48, B :50, dt :52, bounds :59-60,
    # solver / th_stop / maxiter :69,79-81 (T = 100 per BASELINE.json; the script has 200)
    "two_dof_vsa_boxddp": dict(
        robot="asr_twodof", gravity=(9.81, 0.0, 0.0), actuator="vsa", frame="EE", target=(.01, .2, .18),
        running=[("gripperPose", "reach", 1e0, None), ("xReg", "state", 1e-1, (1.0, 1.0, 1.0, 1.0)),
                 ("uReg", "control", 1e-1, (1.0, 1.0, 1.0, 1.0))],
        terminal=[("gripperPose", "reach", 4e4, None)],
        motor_inertia=1e-3, dt=1e-2, u_lb=(-100, -100, 0, 0), u_ub=(100, 100, 100, 100),
        solver="SolverBoxDDP", maxiter=400, th_stop=1e-7, T=100),
    # examples/two_dof_vsa_modified.py:14-67: linear stiffness cost (stiffness_cost.py), torque-only control
    # regulariser, stiffness bounded below by 0.002
    "two_dof_vsa_modified": dict(
        robot="asr_twodof", gravity=(9.81, 0.0, 0.0), actuator="vsa", frame="EE", target=(.01, .2, .18),
        running=[("gripperPose", "reach", 1e0, None), ("xReg", "state", 1e-3, (1.0, 1.0, 1.0, 1.0)),
                 ("uReg", "control", 1e-2, (1.0, 1.0, 0.0, 0.0)), ("vsa", "stiffness", 1e-2, (10, 0.002))],
        terminal=[("gripperPose", "reach", 1e4, None)],
        motor_inertia=1e-3, dt=1e-2, u_lb=(-100, -100, 0.002, 0.002), u_ub=(100, 100, 100, 100),
        solver="SolverBoxDDP", maxiter=400, th_stop=1e-7, T=200),
    # examples/two_dof_sea.py: gravity :20, weights :27-47, K, B :50-51, dt :53, solver / th_stop / maxiter :69,79-81
    "two_dof_sea": dict(
        robot="asr_twodof", gravity=(9.81, 0.0, 0.0), actuator="sea", frame="EE",
        target=(0.01, 2.03063311e-01, 1.80000000e-01),
        running=[("gripperPose", "reach", 1e-1, None), ("xReg", "state", 1e-3, (1.0, 0.0, 1.0, 0.0)),
                 ("uReg", "control", 1e-2, None)],
        terminal=[("gripperPose", "reach", 1e4, None)],
        stiffness=1.0, motor_inertia=1e-2, dt=1e-2, solver="SolverDDP", maxiter=100, th_stop=1e-7, T=100),
    # examples/double_pendulum.py:13-53 (SolverDDP, T = 100 per BASELINE.json; nu = 2, see ActuationModelDoublePendulum)
    "double_pendulum": dict(
        robot="double_pendulum", gravity=None, actuator="pendulum", frame=None, target=None,
        running=[("uReg", "control", 1e-1, (1.0, 0.0)), ("xReg", "state", 1e-2, (1.0, 0.0, 1.0, 0.0)),
                 ("xGoalR", "pendulum", 1e-1, (1, 1, 1, 1, .1, .1))],
        terminal=[("xGoal", "pendulum", 1e4, (1, 1, 1, 1, .1, .1))],
        stiffness=1.0, motor_inertia=1e-3, dt=1e-2, solver="SolverDDP", maxiter=100, th_stop=1e-9, T=100),
    # the same problem with ActuationModelDoublePendulum(state, actLink=0, nu=1) (python/aslr_to/__init__.py:279-281): ONE
    # motor command; the control cost then has one weight
    "double_pendulum_nu1": dict(
        robot="double_pendulum", gravity=None, actuator="pendulum", pendulum_nu=1, frame=None, target=None,
        running=[("uReg", "control", 1e-1, (1.0,)), ("xReg", "state", 1e-2, (1.0, 0.0, 1.0, 0.0)),
                 ("xGoalR", "pendulum", 1e-1, (1, 1, 1, 1, .1, .1))],
        terminal=[("xGoal", "pendulum", 1e4, (1, 1, 1, 1, .1, .1))],
        stiffness=1.0, motor_inertia=1e-3, dt=1e-2, solver="SolverDDP", maxiter=100, th_stop=1e-9, T=100),
    # C5: 7-DoF arm + SEA actuation (nx = 28, nu = 7).  No example script in the reference: the model follows
    # unittest/test_asr_free_fwddyn.py:50-56 (default K = 0.1 I, B = 1e-3 I, free_fwddyn_asr.py:12-19) with C2's
    # cost stack and a 7-DoF end-effector target
    "talos_arm_sea": dict(
        robot="talos_arm", gravity=None, actuator="sea", frame="gripper_left_joint", target=(0.15, 0.35, -0.25),
        running=[("gripperPose", "reach", 1e-1, None), ("xReg", "state", 1e-3, (1.0, 0.0, 1.0, 0.0)),
                 ("uReg", "control", 1e-2, None)],
        terminal=[("gripperPose", "reach", 1e4, None)],
        stiffness=None, motor_inertia=None, dt=1e-2, solver="SolverDDP", maxiter=100, th_stop=1e-7, T=150),
}


def _cost_stack(entries, state, nu, nj, frame_id, target):
    stack = CostModelSum(state, nu)
    for name, kind, weight, par in entries:
        if kind == "reach":
            res = ResidualModelFramePlacementASR(state, frame_id, SE3(np.eye(3), np.array(target, dtype=float)), nu)
            cost = CostModelResidual(state, res)
        elif kind == "state":
            w = np.repeat(np.array(par, dtype=float), nj)
            cost = CostModelResidual(state, ActivationModelWeightedQuad(w), ResidualModelState(state, state.zero(), nu))
        elif kind == "control":
            res = ResidualModelControl(state, nu)
            cost = (CostModelResidual(state, res) if par is None else
                    CostModelResidual(state, ActivationModelWeightedQuad(np.array(par, dtype=float)), res))
        elif kind == "pendulum":
            cost = CostModelDoublePendulum(state, ActivationModelWeightedQuad(np.array(par, dtype=float)), nu)
        elif kind == "stiffness":
            cost = CostModelStiffness(state, nu, par[0], par[1] * np.ones(nu // 2))
        else:
            raise ValueError("unknown cost kind %r" % kind)
        stack.addCost(name, cost, weight)
    return stack


def _build(name, B, T, seed):
    spec = SPECS[name]
    model = example_robot_data.load(spec["robot"]).model
    if spec["gravity"] is not None:
        model.gravity.linear = np.array(spec["gravity"], dtype=float)
    state = StateMultibodyASR(model)
    nj = model.nv
    if spec["actuator"] == "vsa":
        actuation = VSAASRActuation(state)
        nu = 2 * actuation.nu
    elif spec["actuator"] == "sea":
        actuation = ASRActuation(state)
        nu = actuation.nu
    else:
        actuation = ActuationModelDoublePendulum(state, actLink=0, nu=spec.get("pendulum_nu", 2))
        nu = actuation.nu
    frame_id = model.getFrameId(spec["frame"]) if spec["frame"] else None
    stacks = [_cost_stack(spec[k], state, nu, nj, frame_id, spec["target"]) for k in ("running", "terminal")]

    def differential(costs):
        if spec["actuator"] == "vsa":
            return DifferentialFreeFwdDynamicsModelVSA(state, actuation, costs, spec["motor_inertia"] * np.eye(nj))
        if spec.get("stiffness") is None:  # the model's own defaults (free_fwddyn_asr.py:12-19)
            return DifferentialFreeASRFwdDynamicsModel(state, actuation, costs)
        return DifferentialFreeASRFwdDynamicsModel(state, actuation, costs, spec["stiffness"] * np.eye(nj),
                                                   spec["motor_inertia"] * np.eye(nj))

    running = IntegratedActionModelEulerASR(differential(stacks[0]), spec["dt"])
    terminal = IntegratedActionModelEulerASR(differential(stacks[1]), 0)
    if "u_lb" in spec:
        running.u_lb = np.array(spec["u_lb"], dtype=float)
        running.u_ub = np.array(spec["u_ub"], dtype=float)
    if spec["target"] is None:
        x0, refs = np.array([[3.14, 0., 0., 0., 0, 0, 0, 0]]), None   # examples/double_pendulum.py:52
    else:
        x0, refs = _batch_inputs(B, seed, nj, np.array(spec["target"], dtype=float))
    return dict(x0=x0, running=[running] * (spec["T"] if T is None else T), terminal=terminal, frame_refs=refs,
                solver=spec["solver"], maxiter=spec["maxiter"], th_stop=spec["th_stop"], name=name)


def _batch_inputs(B, seed, nj, p_nom):
    """q_l = q_m ~ U(-0.5, 0.5)^nj, v = 0; target p = p_nom + U(-0.03, 0.03) (1,1,0).
    Trajectory 0 is the example script's nominal problem (x0 = 0, nominal target)."""
    rng = np.random.default_rng(seed)
    q = rng.uniform(-0.5, 0.5, (B, nj))
    dp = rng.uniform(-0.03, 0.03, (B, 3)) * np.array([1.0, 1.0, 0.0])
    q[0] = 0.0
    dp[0] = 0.0
    x0 = np.concatenate([q, q, np.zeros((B, 2 * nj))], axis=1)
    refs = np.concatenate([np.tile(np.eye(3).reshape(9), (B, 1)), p_nom + dp], axis=1)
    return x0, refs


def two_dof_vsa_boxddp(B=1, T=100, seed=0):
    """C3 / C4."""
    return _build("two_dof_vsa_boxddp", B, T, seed)


def two_dof_vsa_modified(B=1, T=200, seed=0):
    return _build("two_dof_vsa_modified", B, T, seed)


def two_dof_sea(B=1, T=100, seed=0):
    """C2."""
    return _build("two_dof_sea", B, T, seed)


def double_pendulum(T=100):
    """C1."""
    return _build("double_pendulum", 1, T, 0)


def double_pendulum_nu1(T=100):
    """C1 with one motor command (nu = 1)."""
    return _build("double_pendulum_nu1", 1, T, 0)


def talos_arm_sea(B=1, T=150, seed=0):
    """C5."""
    return _build("talos_arm_sea", B, T, seed)


SCENARIOS = {"two_dof_vsa_boxddp": two_dof_vsa_boxddp, "two_dof_vsa_modified": two_dof_vsa_modified,
             "two_dof_sea": two_dof_sea,
             "double_pendulum": double_pendulum, "double_pendulum_nu1": double_pendulum_nu1, "talos_arm_sea": talos_arm_sea}


def lower(sc):
    from .lowering import lower_problem
    return lower_problem(sc["x0"], sc["running"], sc["terminal"], sc["frame_refs"])


def solver_params(sc, **overrides):
    from . import _abi
    kind = {"SolverDDP": _abi.SOLVER_DDP, "SolverFDDP": _abi.SOLVER_FDDP, "SolverBoxDDP": _abi.SOLVER_BOXDDP}[
        overrides.pop("solver", sc["solver"])]
    sp = _abi.default_solver_params(kind)
    sp.maxiter = sc["maxiter"]
    sp.th_stop = sc["th_stop"]
    for k, v in overrides.items():
        setattr(sp, k, v)
    return sp
