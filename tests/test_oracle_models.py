"""The reference's own acceptance technique (unittest/test_vsa_freefwddyn.py:23-39,
unittest/test_stiffness_residual.py:12-56): analytic derivatives of the SEA / VSA models against finite
differences of calc at random (x, u) -- applied to the CPU oracle.  The reference's tolerances are
3e4 * sqrt(2 eps) ~ 6.3e-3 (Fx, Fu) and 3e-2 (Lx, Lu); central differences let us hold 1e-5.
"""
import numpy as np
import pytest

from aslr_to_amd import crocoddyl, example_robot_data, scenarios
from aslr_to_amd.lowering import lower_problem
from aslr_to_amd.models import (ASRActuation, CostModelStiffness, DifferentialFreeASRFwdDynamicsModel,
                                DifferentialFreeFwdDynamicsModelVSA, IntegratedActionModelEulerASR,
                                ResidualModelFramePlacementASR, StateMultibodyASR, VSAASRActuation)
from aslr_to_amd.pinocchio import SE3


def _numdiff(f, z, h=1e-6):
    f0 = np.atleast_1d(f(z))
    J = np.zeros((f0.size, z.size))
    for k in range(z.size):
        e = np.zeros(z.size)
        e[k] = h
        J[:, k] = (np.atleast_1d(f(z + e)) - np.atleast_1d(f(z - e))) / (2 * h)
    return J


def _vsa_model(robot, frame, target, with_stiffness):
    model = example_robot_data.load(robot).model
    state = StateMultibodyASR(model)
    actuation = VSAASRActuation(state)
    nu = 2 * actuation.nu
    costs = crocoddyl.CostModelSum(state, nu)
    res = ResidualModelFramePlacementASR(state, model.getFrameId(frame), SE3(np.eye(3), np.array(target)), nu)
    costs.addCost("gripperPose", crocoddyl.CostModelResidual(state, res), float(nu))
    costs.addCost("xReg", crocoddyl.CostModelResidual(state, crocoddyl.ResidualModelControl(state, nu)), 1e-2)
    if with_stiffness:
        costs.addCost("vsa", CostModelStiffness(state, nu, .1, np.zeros(nu // 2)), 1e0)
    return DifferentialFreeFwdDynamicsModelVSA(state, actuation, costs)


def _sea_model(robot, frame=None, target=None):
    model = example_robot_data.load(robot).model
    state = StateMultibodyASR(model)
    actuation = ASRActuation(state)
    nu = actuation.nu
    costs = crocoddyl.CostModelSum(state, nu)
    if frame is not None:
        res = ResidualModelFramePlacementASR(state, model.getFrameId(frame), SE3(np.eye(3), np.array(target)), nu)
        costs.addCost("gripperPose", crocoddyl.CostModelResidual(state, res), 1.0)
        w = np.linspace(0.5, 1.5, state.ndx)
        costs.addCost("xReg", crocoddyl.CostModelResidual(state, crocoddyl.ActivationModelWeightedQuad(w),
                                                          crocoddyl.ResidualModelState(state, 0.1 * np.ones(state.nx), nu)), 1e-1)
        costs.addCost("uReg", crocoddyl.CostModelResidual(state, crocoddyl.ResidualModelControl(state, nu)), 1e-2)
    return DifferentialFreeASRFwdDynamicsModel(state, actuation, costs)


CASES = {
    "sea_twodof_nocost (test_vsa_freefwddyn.py)": lambda: _sea_model("asr_twodof"),
    "vsa_twodof_stiffness (test_stiffness_residual.py)": lambda: _vsa_model("asr_twodof", "EE", [.0, .0, .4], True),
    "vsa_twodof": lambda: _vsa_model("asr_twodof", "EE", [.01, .2, .18], False),
    "sea_twodof_costs": lambda: _sea_model("asr_twodof", "EE", [.01, .2, .18]),
    "sea_talos_arm (test_asr_free_fwddyn.py)": lambda: _sea_model("talos_arm", "gripper_left_joint", [.15, .35, -.25]),
}


@pytest.mark.parametrize("case", list(CASES))
def test_dam_derivatives_match_finite_differences(oracle, case):
    dam = CASES[case]()
    iam = IntegratedActionModelEulerASR(dam, 1e-2)
    low = lower_problem(np.zeros(dam.state.nx), [iam], iam)
    nx, nu = low.nx, low.nu
    rng = np.random.default_rng(5)
    x = rng.uniform(-1, 1, nx)
    u = rng.uniform(0.1, 1.0, nu)
    d = oracle.dam(low, 0, x, u)
    Fx = _numdiff(lambda z: oracle.dam(low, 0, z, u)["xout"], x)
    Fu = _numdiff(lambda z: oracle.dam(low, 0, x, z)["xout"], u)
    Lx = _numdiff(lambda z: oracle.dam(low, 0, z, u)["cost"], x).ravel()
    Lu = _numdiff(lambda z: oracle.dam(low, 0, x, z)["cost"], u).ravel()
    tol = 1e-5
    assert np.abs(d["Fx"] - Fx).max() < tol
    assert np.abs(d["Fu"] - Fu).max() < tol
    assert np.abs(d["Lx"] - Lx).max() < tol * (1 + np.abs(Lx).max())
    assert np.abs(d["Lu"] - Lu).max() < tol * (1 + np.abs(Lu).max())
    # Gauss-Newton Hessians: symmetric PSD, and Lxu = 0 for every cost type of the package
    assert np.allclose(d["Lxx"], d["Lxx"].T, atol=1e-12) and np.linalg.eigvalsh(d["Lxx"]).min() > -1e-9
    assert np.allclose(d["Luu"], d["Luu"].T) and np.linalg.eigvalsh(d["Luu"]).min() > -1e-12
    assert not d["Lxu"].any()


@pytest.mark.parametrize("case", ["vsa_twodof", "sea_twodof_costs"])
def test_integrated_model_follows_the_semi_implicit_euler_step(oracle, case):
    """integrated_action.py:23-37: xnext = x + [v dt + a dt^2, a dt]; Fx, Fu by finite differences; cost NOT
    scaled by dt; the dt = 0 terminal model has Fx = I, Fu = 0, xnext = x."""
    dam = CASES[case]()
    dt = 1e-2
    iam, term = IntegratedActionModelEulerASR(dam, dt), IntegratedActionModelEulerASR(dam, 0)
    low = lower_problem(np.zeros(dam.state.nx), [iam], term)
    nx, nu, nv = low.nx, low.nu, low.nx // 2
    rng = np.random.default_rng(6)
    x, u = rng.uniform(-1, 1, nx), rng.uniform(0.1, 1.0, nu)
    k = oracle.knot(low, 0, x, u)
    d = oracle.dam(low, 0, x, u)
    xn = x.copy()
    xn[:nv] += x[nv:] * dt + d["xout"] * dt * dt
    xn[nv:] += d["xout"] * dt
    np.testing.assert_allclose(k["xnext"], xn, atol=1e-14)
    assert abs(k["cost"] - d["cost"]) < 1e-14 * (1 + abs(d["cost"]))
    Fx = _numdiff(lambda z: oracle.knot(low, 0, z, u, diff=False)["xnext"], x)
    Fu = _numdiff(lambda z: oracle.knot(low, 0, x, z, diff=False)["xnext"], u)
    assert np.abs(k["Fx"] - Fx).max() < 1e-6 and np.abs(k["Fu"] - Fu).max() < 1e-6
    for name in ("Lx", "Lu", "Lxx", "Lxu", "Luu"):
        np.testing.assert_array_equal(k[name], d[name])
    kt = oracle.knot(low, 1, x, u)
    np.testing.assert_array_equal(kt["xnext"], x)
    np.testing.assert_array_equal(kt["Fx"], np.eye(nx))
    assert not kt["Fu"].any()


def test_double_pendulum_cost_and_actuation_follow_the_reference(oracle):
    """__init__.py:228-249 (cost) and :262-290 (actuation: only motor 1 is driven, S[nv/2, 0] = 1)."""
    sc = scenarios.double_pendulum(T=3)
    low = scenarios.lower(sc)
    x = np.array([0.3, -0.7, 0.1, 0.2, 0.5, -0.4, 0.3, 0.9])
    u = np.array([0.7, -1.3])
    d = oracle.dam(low, 0, x, u)
    c1, c2, s1, s2 = np.cos(x[0]), np.cos(x[1]), np.sin(x[0]), np.sin(x[1])
    r = np.array([s1, s2, 1 + c1, 1 + c2, x[4], x[5]])
    w = np.array([1, 1, 1, 1, .1, .1])
    xw = np.array([1, 1, 0, 0, 1, 1, 0, 0])
    cost = 1e-1 * 0.5 * (1 * u[0] ** 2) + 1e-2 * 0.5 * np.sum(xw * x ** 2) + 1e-1 * 0.5 * np.sum(w * r ** 2)
    assert abs(d["cost"] - cost) < 1e-13
    assert d["Fu"][2, 0] == pytest.approx(1000.0) and not d["Fu"][:, 1].any() and not d["Fu"][:2].any()
    Lxx_pend = np.zeros(8)
    Lxx_pend[0] = (c1 ** 2 - s1 ** 2) * w[0] + (s1 ** 2 + (1 - c1) * c1) * w[2]
    Lxx_pend[1] = (c2 ** 2 - s2 ** 2) * w[1] + (s2 ** 2 + (1 - c2) * c2) * w[3]
    Lxx_pend[4], Lxx_pend[5] = w[4], w[5]
    np.testing.assert_allclose(np.diag(d["Lxx"]), 1e-1 * Lxx_pend + 1e-2 * xw, atol=1e-14)
