"""The BASELINE.json configurations as concrete, seeded, synthetic batches.

Each builder mirrors one of the reference's example scripts (cited) with the robot tables of
example_robot_data.py; it returns (x0s[B,nx], running_models (list of T), terminal_model,
frame_refs or None, solver_name, maxiter).  Pure host logic.
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
    """C3 / C4: examples/two_dof_vsa_boxddp.py:14-81 (T = 100 per BASELINE.json; the script has 200)."""
    robot_model = example_robot_data.load('asr_twodof').model
    robot_model.gravity.linear = np.array([9.81, 0, 0])
    state = StateMultibodyASR(robot_model)
    actuation = VSAASRActuation(state)
    nu = 2 * actuation.nu
    p_nom = np.array([.01, .2, .18])
    framePlacementResidual = ResidualModelFramePlacementASR(state, robot_model.getFrameId("EE"), SE3(np.eye(3), p_nom), nu)
    goalTrackingCost = CostModelResidual(state, framePlacementResidual)
    xActivation = ActivationModelWeightedQuad(np.array([1e0] * 2 + [1e0] * 2 + [1e0] * robot_model.nv + [1e0] * robot_model.nv))
    xRegCost = CostModelResidual(state, xActivation, ResidualModelState(state, state.zero(), nu))
    uActivation = ActivationModelWeightedQuad(np.array([1e0] + [1e0] + [1e0] * 2))
    uRegCost = CostModelResidual(state, uActivation, ResidualModelControl(state, nu))
    runningCostModel = CostModelSum(state, nu)
    terminalCostModel = CostModelSum(state, nu)
    runningCostModel.addCost("gripperPose", goalTrackingCost, 1e0)
    runningCostModel.addCost("xReg", xRegCost, 1e-1)
    runningCostModel.addCost("uReg", uRegCost, 1e-1)
    terminalCostModel.addCost("gripperPose", goalTrackingCost, 4e4)
    Bm = .001 * np.eye(int(state.nv / 2))
    dt = 1e-2
    runningModel = IntegratedActionModelEulerASR(DifferentialFreeFwdDynamicsModelVSA(state, actuation, runningCostModel, Bm), dt)
    terminalModel = IntegratedActionModelEulerASR(DifferentialFreeFwdDynamicsModelVSA(state, actuation, terminalCostModel, Bm), 0)
    runningModel.u_lb = np.array([-100, -100, 0, 0])
    runningModel.u_ub = np.array([100, 100, 100, 100])
    x0, refs = _batch_inputs(B, seed, 2, p_nom)
    return dict(x0=x0, running=[runningModel] * T, terminal=terminalModel, frame_refs=refs, solver="SolverBoxDDP",
                maxiter=400, th_stop=1e-7, name="two_dof_vsa_boxddp")


def two_dof_vsa_modified(B=1, T=200, seed=0):
    """examples/two_dof_vsa_modified.py:14-67: the VSA arm with the linear stiffness cost (stiffness_cost.py), a
    control regulariser on the motor torques only, and the stiffness bounded below by 0.002."""
    robot_model = example_robot_data.load('asr_twodof').model
    robot_model.gravity.linear = np.array([9.81, 0, 0])
    state = StateMultibodyASR(robot_model)
    actuation = VSAASRActuation(state)
    nu = 2 * actuation.nu
    p_nom = np.array([.01, .2, .18])
    framePlacementResidual = ResidualModelFramePlacementASR(state, robot_model.getFrameId("EE"), SE3(np.eye(3), p_nom), nu)
    goalTrackingCost = CostModelResidual(state, framePlacementResidual)
    xActivation = ActivationModelWeightedQuad(np.array([1e0] * 2 + [1e0] * 2 + [1e0] * robot_model.nv + [1e0] * robot_model.nv))
    xRegCost = CostModelResidual(state, xActivation, ResidualModelState(state, state.zero(), nu))
    uActivation = ActivationModelWeightedQuad(np.array([1e0] + [1e0] + [0] * 2))
    uRegCost = CostModelResidual(state, uActivation, ResidualModelControl(state, nu))
    lamda = 10
    Kref = 0.002 * np.ones(int(nu / 2))
    vsaCost = CostModelStiffness(state, nu, lamda, Kref)
    runningCostModel = CostModelSum(state, nu)
    terminalCostModel = CostModelSum(state, nu)
    runningCostModel.addCost("gripperPose", goalTrackingCost, 1e0)
    runningCostModel.addCost("xReg", xRegCost, 1e-3)
    runningCostModel.addCost("uReg", uRegCost, 1e-2)
    runningCostModel.addCost("vsa", vsaCost, 1e-2)
    terminalCostModel.addCost("gripperPose", goalTrackingCost, 1e4)
    Bm = .001 * np.eye(int(state.nv / 2))
    dt = 1e-2
    runningModel = IntegratedActionModelEulerASR(DifferentialFreeFwdDynamicsModelVSA(state, actuation, runningCostModel, Bm), dt)
    terminalModel = IntegratedActionModelEulerASR(DifferentialFreeFwdDynamicsModelVSA(state, actuation, terminalCostModel, Bm), 0)
    runningModel.u_lb = np.array([-100, -100, 0.002, 0.002])
    runningModel.u_ub = np.array([100, 100, 100, 100])
    x0, refs = _batch_inputs(B, seed, 2, p_nom)
    return dict(x0=x0, running=[runningModel] * T, terminal=terminalModel, frame_refs=refs, solver="SolverBoxDDP",
                maxiter=400, th_stop=1e-7, name="two_dof_vsa_modified")


def two_dof_sea(B=1, T=100, seed=0):
    """C2: examples/two_dof_sea.py:18-81."""
    robot_model = example_robot_data.load('asr_twodof').model
    robot_model.gravity.linear = np.array([9.81, 0, 0])
    state = StateMultibodyASR(robot_model)
    actuation = ASRActuation(state)
    nu = actuation.nu
    runningCostModel = CostModelSum(state, nu)
    terminalCostModel = CostModelSum(state, nu)
    xActivation = ActivationModelWeightedQuad(np.array([1e0] * 2 + [0] * 2 + [1e0] * robot_model.nv + [0] * robot_model.nv))
    xRegCost = CostModelResidual(state, xActivation, ResidualModelState(state, state.zero(), nu))
    uRegCost = CostModelResidual(state, ResidualModelControl(state, nu))
    p_nom = np.array([0.01, 2.03063311e-01, 1.80000000e-01])
    framePlacementResidual = ResidualModelFramePlacementASR(state, robot_model.getFrameId("EE"), SE3(np.eye(3), p_nom), nu)
    goalTrackingCost = CostModelResidual(state, framePlacementResidual)
    runningCostModel.addCost("gripperPose", goalTrackingCost, 1e-1)
    runningCostModel.addCost("xReg", xRegCost, 1e-3)
    runningCostModel.addCost("uReg", uRegCost, 1e-2)
    terminalCostModel.addCost("gripperPose", goalTrackingCost, 1e4)
    K = 1 * np.eye(int(state.nv / 2))
    Bm = .01 * np.eye(int(state.nv / 2))
    dt = 1e-2
    runningModel = IntegratedActionModelEulerASR(DifferentialFreeASRFwdDynamicsModel(state, actuation, runningCostModel, K, Bm), dt)
    terminalModel = IntegratedActionModelEulerASR(DifferentialFreeASRFwdDynamicsModel(state, actuation, terminalCostModel, K, Bm), 0)
    x0, refs = _batch_inputs(B, seed, 2, p_nom)
    return dict(x0=x0, running=[runningModel] * T, terminal=terminalModel, frame_refs=refs, solver="SolverDDP",
                maxiter=100, th_stop=1e-7, name="two_dof_sea")


def double_pendulum(T=100):
    """C1: examples/double_pendulum.py:13-53 (SolverDDP, T = 100 per BASELINE.json; nu = 2, see
    ActuationModelDoublePendulum)."""
    model = example_robot_data.load('double_pendulum').model
    state = StateMultibodyASR(model)
    actuation = ActuationModelDoublePendulum(state, actLink=0, nu=2)
    nu = actuation.nu
    runningCostModel = CostModelSum(state, nu)
    terminalCostModel = CostModelSum(state, nu)
    xActivation = ActivationModelWeightedQuad(np.array([1e0] * 2 + [0] * 2 + [1e0] * model.nv + [0] * model.nv))
    xRegCost = CostModelResidual(state, xActivation, ResidualModelState(state, state.zero(), nu))
    uRegCost = CostModelResidual(state, ActivationModelWeightedQuad(np.array([1., 0.])), ResidualModelControl(state, nu))
    xPendCost = CostModelDoublePendulum(state, ActivationModelWeightedQuad(np.array([1] * 4 + [.1] * 2)), nu)
    dt = 1e-2
    runningCostModel.addCost("uReg", uRegCost, 1e-1)
    runningCostModel.addCost("xReg", xRegCost, 1e-2)
    runningCostModel.addCost("xGoalR", xPendCost, 1e-1)
    terminalCostModel.addCost("xGoal", xPendCost, 1e4)
    K = 1 * np.eye(int(state.nv / 2))
    Bm = .001 * np.eye(int(state.nv / 2))
    runningModel = IntegratedActionModelEulerASR(DifferentialFreeASRFwdDynamicsModel(state, actuation, runningCostModel, K, Bm), dt)
    terminalModel = IntegratedActionModelEulerASR(DifferentialFreeASRFwdDynamicsModel(state, actuation, terminalCostModel, K, Bm), 0)
    x0 = np.array([[3.14, 0., 0., 0., 0, 0, 0, 0]])
    return dict(x0=x0, running=[runningModel] * T, terminal=terminalModel, frame_refs=None, solver="SolverDDP",
                maxiter=100, th_stop=1e-9, name="double_pendulum")


def talos_arm_sea(B=1, T=150, seed=0):
    """C5: 7-DoF arm + SEA actuation (nx = 28, nu = 7).  The reference has no example script for it;
    the model follows unittest/test_asr_free_fwddyn.py:50-56 (default K = 0.1 I, B = 1e-3 I,
    free_fwddyn_asr.py:12-19) with C2's cost stack and a 7-DoF EE target."""
    robot_model = example_robot_data.load('talos_arm').model
    state = StateMultibodyASR(robot_model)
    actuation = ASRActuation(state)
    nu = actuation.nu
    nv = robot_model.nv
    runningCostModel = CostModelSum(state, nu)
    terminalCostModel = CostModelSum(state, nu)
    xActivation = ActivationModelWeightedQuad(np.array([1e0] * nv + [0] * nv + [1e0] * nv + [0] * nv))
    xRegCost = CostModelResidual(state, xActivation, ResidualModelState(state, state.zero(), nu))
    uRegCost = CostModelResidual(state, ResidualModelControl(state, nu))
    p_nom = np.array([0.15, 0.35, -0.25])
    framePlacementResidual = ResidualModelFramePlacementASR(state, robot_model.getFrameId("gripper_left_joint"),
                                                            SE3(np.eye(3), p_nom), nu)
    goalTrackingCost = CostModelResidual(state, framePlacementResidual)
    runningCostModel.addCost("gripperPose", goalTrackingCost, 1e-1)
    runningCostModel.addCost("xReg", xRegCost, 1e-3)
    runningCostModel.addCost("uReg", uRegCost, 1e-2)
    terminalCostModel.addCost("gripperPose", goalTrackingCost, 1e4)
    dt = 1e-2
    runningModel = IntegratedActionModelEulerASR(DifferentialFreeASRFwdDynamicsModel(state, actuation, runningCostModel), dt)
    terminalModel = IntegratedActionModelEulerASR(DifferentialFreeASRFwdDynamicsModel(state, actuation, terminalCostModel), 0)
    x0, refs = _batch_inputs(B, seed, 7, p_nom)
    return dict(x0=x0, running=[runningModel] * T, terminal=terminalModel, frame_refs=refs, solver="SolverDDP",
                maxiter=100, th_stop=1e-7, name="talos_arm_sea")


SCENARIOS = {"two_dof_vsa_boxddp": two_dof_vsa_boxddp, "two_dof_vsa_modified": two_dof_vsa_modified,
             "two_dof_sea": two_dof_sea,
             "double_pendulum": double_pendulum, "talos_arm_sea": talos_arm_sea}


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
