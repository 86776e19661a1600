"""GPU tests through the reference-shaped Python API (the drop-in surface of SURVEY.md 8(b)) and
size-independent properties at the BASELINE batch size.  Everything computes through the C ABI.
"""
import numpy as np
import pytest

import aslr_to_amd as aslr_to
from aslr_to_amd import _abi, crocoddyl, example_robot_data, pinocchio, scenarios

pytestmark = pytest.mark.gpu


def _numdiff(f, z, h=1e-6):
    f0 = np.atleast_1d(f(z))
    J = np.zeros((f0.size, z.size))
    for k in range(z.size):
        e = np.zeros(z.size)
        e[k] = h
        J[:, k] = (np.atleast_1d(f(z + e)) - np.atleast_1d(f(z - e))) / (2 * h)
    return J


def test_reference_unit_test_vsa_stiffness_residual_numdiff():
    """unittest/test_stiffness_residual.py:12-56 written against the drop-in package: analytic Fx, Fu, Lx, Lu
    of the VSA model vs finite differences of calc (the reference's tolerances: 6.3e-3 and 3e-2)."""
    two_dof = example_robot_data.load('asr_twodof')
    robot_model = two_dof.model
    state = aslr_to.StateMultibodyASR(robot_model)
    actuation = aslr_to.VSAASRActuation(state)
    nu = 2 * actuation.nu
    costs = crocoddyl.CostModelSum(state, nu)
    framePlacementResidual = aslr_to.ResidualModelFramePlacementASR(
        state, robot_model.getFrameId("EE"), pinocchio.SE3(np.eye(3), np.array([.0, .0, .4])), nu)
    costs.addCost("gripperPose", crocoddyl.CostModelResidual(state, framePlacementResidual), nu)
    costs.addCost("xReg", crocoddyl.CostModelResidual(state, crocoddyl.ResidualModelControl(state, nu)), 1e-2)
    costs.addCost("vsa", aslr_to.CostModelStiffness(state, nu, .1, np.zeros(int(nu / 2))), 1e0)
    model = aslr_to.DifferentialFreeFwdDynamicsModelVSA(state, actuation, costs)
    np.random.seed(0)
    x = model.state.rand()
    u = np.random.rand(model.nu)
    data = model.createData()
    model.calc(data, x, u)
    model.calcDiff(data, x, u)

    def xout(xx, uu):
        d = model.createData()
        model.calc(d, xx, uu)
        return d.xout.copy()

    def cost(xx, uu):
        d = model.createData()
        model.calc(d, xx, uu)
        return d.cost
    assert np.allclose(data.Fx, _numdiff(lambda z: xout(z, u), x), atol=6.3e-3)
    assert np.allclose(data.Fu, _numdiff(lambda z: xout(x, z), u), atol=6.3e-3)
    assert np.allclose(data.Lx, _numdiff(lambda z: cost(z, u), x).ravel(), atol=3e-2)
    assert np.allclose(data.Lu, _numdiff(lambda z: cost(x, z), u).ravel(), atol=3e-2)
    # and tighter than the reference asks
    assert np.abs(data.Fx - _numdiff(lambda z: xout(z, u), x)).max() < 1e-4


def test_dam_and_integrated_calc_match_oracle_through_the_python_api(oracle):
    sc = scenarios.two_dof_sea(B=1, T=2)
    iam = sc["running"][0]
    low = scenarios.lower(sc)
    rng = np.random.default_rng(2)
    x, u = rng.uniform(-1, 1, 8), rng.uniform(-1, 1, 2)
    d = iam.differential.createData()
    iam.differential.calc(d, x, u)
    iam.differential.calcDiff(d, x, u)
    ref = oracle.dam(low, 0, x, u, frame_ref=None)
    for k in ("xout", "Fx", "Fu", "Lx", "Lu", "Lxx", "Lxu", "Luu"):
        np.testing.assert_allclose(getattr(d, k), ref[k], rtol=1e-9, atol=1e-10)
    assert d.cost == pytest.approx(ref["cost"], rel=1e-12)
    di = iam.createData()
    xnext, c = iam.calc(di, x, u)
    iam.calcDiff(di, x, u)
    kr = oracle.knot(low, 0, x, u)
    np.testing.assert_allclose(xnext, kr["xnext"], atol=1e-13)
    for k in ("Fx", "Fu", "Lx", "Lu", "Lxx", "Luu"):
        np.testing.assert_allclose(getattr(di, k), kr[k], rtol=1e-9, atol=1e-10)


def test_vsa_boxddp_example_script_flow(oracle):
    """examples/two_dof_vsa_boxddp.py:14-87 with the stand-in namespaces (T = 100 per BASELINE.json)."""
    robot_model = example_robot_data.load('asr_twodof').model
    robot_model.gravity.linear = np.array([9.81, 0, 0])
    state = aslr_to.StateMultibodyASR(robot_model)
    actuation = aslr_to.VSAASRActuation(state)
    nu = 2 * actuation.nu
    framePlacementResidual = aslr_to.ResidualModelFramePlacementASR(
        state, robot_model.getFrameId("EE"), pinocchio.SE3(np.eye(3), np.array([.01, .2, .18])), nu)
    goalTrackingCost = crocoddyl.CostModelResidual(state, framePlacementResidual)
    xActivation = crocoddyl.ActivationModelWeightedQuad(np.array([1e0] * 2 + [1e0] * 2 + [1e0] * robot_model.nv + [1e0] * robot_model.nv))
    xRegCost = crocoddyl.CostModelResidual(state, xActivation, crocoddyl.ResidualModelState(state, state.zero(), nu))
    uActivation = crocoddyl.ActivationModelWeightedQuad(np.array([1e0] + [1e0] + [1e0] * 2))
    uRegCost = crocoddyl.CostModelResidual(state, uActivation, crocoddyl.ResidualModelControl(state, nu))
    runningCostModel = crocoddyl.CostModelSum(state, nu)
    terminalCostModel = crocoddyl.CostModelSum(state, nu)
    runningCostModel.addCost("gripperPose", goalTrackingCost, 1e0)
    runningCostModel.addCost("xReg", xRegCost, 1e-1)
    runningCostModel.addCost("uReg", uRegCost, 1e-1)
    terminalCostModel.addCost("gripperPose", goalTrackingCost, 4e4)
    B = .001 * np.eye(int(state.nv / 2))
    dt = 1e-2
    runningModel = aslr_to.IntegratedActionModelEulerASR(
        aslr_to.DifferentialFreeFwdDynamicsModelVSA(state, actuation, runningCostModel, B), dt)
    terminalModel = aslr_to.IntegratedActionModelEulerASR(
        aslr_to.DifferentialFreeFwdDynamicsModelVSA(state, actuation, terminalCostModel, B), 0)
    runningModel.u_lb = np.array([-100, -100, 0, 0])
    runningModel.u_ub = np.array([100, 100, 100, 100])
    T = 100
    x0 = np.concatenate([np.array([.0, .0]), np.zeros(2), pinocchio.utils.zero(state.nv)])
    problem = crocoddyl.ShootingProblem(x0, [runningModel] * T, terminalModel)
    solver = crocoddyl.SolverBoxDDP(problem)
    solver.setCallbacks([crocoddyl.CallbackLogger()])
    solver.th_stop = 1e-7
    converged = solver.solve([], [], 400)
    log = solver.getCallbacks()[0]
    assert converged and solver.stop < 1e-7
    assert len(solver.xs) == T + 1 and len(solver.us) == T and solver.xs[0].shape == (8,)
    assert len(log.costs) == solver.iter + 1 and log.costs[-1] == pytest.approx(solver.cost)
    assert all(b <= a + 1e-9 for a, b in zip(log.costs[1:], log.costs[2:]))  # accepted steps never raise the cost
    assert aslr_to.u_squared(log).shape == (4,)
    us = np.array(solver.us)
    assert (us >= runningModel.u_lb - 1e-12).all() and (us <= runningModel.u_ub + 1e-12).all()
    # same answer as the CPU oracle on the same ShootingProblem
    sp = _abi.default_solver_params(_abi.SOLVER_BOXDDP)
    sp.maxiter, sp.th_stop = 400, 1e-7
    ref = oracle.solve(problem.lowered, sp)
    assert np.abs(np.array(solver.xs) - ref["xs"][:, 0]).max() < 1e-6
    assert np.abs(us - ref["us"][:, 0]).max() < 1e-6
    assert abs(solver.cost - ref["traj_f"][_abi.TF_COST][0]) < 1e-4
    assert solver.iter + 1 == ref["traj_i"][_abi.TI_ITER][0]
    # terminal EE position, read as the script does (examples/two_dof_vsa_boxddp.py:83-84)
    ee = oracle.frame_placement(problem.lowered.desc.chain, solver.xs[-1][:2], 1, np.eye(3), [0.12, -2.03063311e-04, 0.0])[1]
    assert np.isfinite(ee).all()


def test_full_batch_properties_and_batch_size_independence():
    """B = 4096, T = 100 (the BASELINE configuration): invariants that need no oracle, and
    bit-identical per-trajectory results whatever the batch (no cross-trajectory arithmetic)."""
    import torch
    from aslr_to_amd.engine import Engine
    sc = scenarios.two_dof_vsa_boxddp(B=4096, T=100, seed=0)
    sp = scenarios.solver_params(sc, maxiter=25)
    e = Engine(scenarios.lower(sc))
    e.set_candidate(None, None)
    e.solve(sp, poll_every=0)
    torch.cuda.synchronize()
    X, U = e.region(_abi.R_XS).clone(), e.region(_abi.R_US).clone()
    x0 = torch.as_tensor(sc["x0"], device=X.device)
    assert torch.equal(X[0], x0)
    assert torch.isfinite(X).all() and torch.isfinite(U).all()
    lb = torch.tensor([-100., -100., 0., 0.], device=U.device, dtype=torch.float64)
    ub = torch.tensor([100., 100., 100., 100.], device=U.device, dtype=torch.float64)
    assert (U >= lb).all() and (U <= ub).all()
    cost_solver = e.traj_f(_abi.TF_COST).clone()
    iters = e.traj_i(_abi.TI_ITER).clone()
    e.calc()  # ShootingProblem.calc on the solution: dynamically feasible, cost = sum of node costs
    torch.cuda.synchronize()
    assert (e.region(_abi.R_XNEXT)[:-1] - X[1:]).abs().max().item() < 1e-11
    assert ((e.region(_abi.R_COST).sum(dim=0) - cost_solver).abs() / (1 + cost_solver.abs())).max().item() < 1e-12
    assert int(iters.min()) >= 1 and int(iters.max()) <= 25
    # the first 70 trajectories solved alone give the same bits
    sub = scenarios.two_dof_vsa_boxddp(B=4096, T=100, seed=0)
    sub["x0"], sub["frame_refs"] = sub["x0"][:70], sub["frame_refs"][:70]
    e2 = Engine(scenarios.lower(sub))
    e2.set_candidate(None, None)
    e2.solve(sp, poll_every=0)
    torch.cuda.synchronize()
    assert torch.equal(e2.region(_abi.R_XS), X[:, :70]) and torch.equal(e2.region(_abi.R_US), U[:, :70])
    assert torch.equal(e2.traj_i(_abi.TI_ITER), iters[:70])
    # re-solving from the solution (feasible warm start) stays there, within what th_stop = 1e-7 leaves
    # (|Qu| ~ 3e-4 per knot), and never raises the cost: idempotence at a stationary point
    conv = (e.traj_i(_abi.TI_STATUS) & _abi.ST_CONVERGED) != 0
    if conv.any():
        e.set_candidate(X.permute(1, 0, 2), U.permute(1, 0, 2))
        sp2 = scenarios.solver_params(sc, maxiter=3, is_feasible=1)
        e.solve(sp2, poll_every=0)
        torch.cuda.synchronize()
        d = (e.region(_abi.R_XS) - X).abs().amax(dim=(0, 2))
        assert d[conv].max().item() < 5e-3
        assert (e.traj_f(_abi.TF_COST)[conv] <= cost_solver[conv] + 1e-9).all()


def test_c5_shard_properties_and_batch_size_independence(oracle):
    """The C5 shard (7-DoF SEA, nx = 28, 512 trajectories x 150 knots) through the block / team kernels:
    invariants that need no oracle at full size, bit-identical per-trajectory results whatever the batch, and the
    oracle on the first two trajectories."""
    import torch
    from aslr_to_amd.engine import Engine
    sc = scenarios.talos_arm_sea(B=512, T=150, seed=0)
    sp = scenarios.solver_params(sc, solver="SolverFDDP", maxiter=6)
    e = Engine(scenarios.lower(sc))
    e.set_candidate(None, None)
    e.solve(sp, poll_every=0)
    torch.cuda.synchronize()
    X, U = e.region(_abi.R_XS).clone(), e.region(_abi.R_US).clone()
    feas = e.traj_i(_abi.TI_FEASIBLE) != 0   # FDDP closes the initial gap x0 - xs[0] only with a full step
    assert feas.any()
    assert torch.equal(X[0][feas], torch.as_tensor(sc["x0"], device=X.device)[feas])
    assert torch.isfinite(X).all() and torch.isfinite(U).all()
    cost_solver, iters = e.traj_f(_abi.TF_COST).clone(), e.traj_i(_abi.TI_ITER).clone()
    assert int(iters.min()) >= 1 and int(iters.max()) <= 6
    e.calc()   # cost = sum of node costs on the solver's iterate
    torch.cuda.synchronize()
    assert ((e.region(_abi.R_COST).sum(dim=0) - cost_solver).abs() / (1 + cost_solver.abs())).max().item() < 1e-12
    sub = scenarios.talos_arm_sea(B=512, T=150, seed=0)
    sub["x0"], sub["frame_refs"] = sub["x0"][:2], sub["frame_refs"][:2]
    low2 = scenarios.lower(sub)
    e2 = Engine(low2)
    e2.set_candidate(None, None)
    e2.solve(sp, poll_every=0)
    torch.cuda.synchronize()
    assert torch.equal(e2.region(_abi.R_XS), X[:, :2]) and torch.equal(e2.region(_abi.R_US), U[:, :2])
    assert torch.equal(e2.traj_i(_abi.TI_ITER), iters[:2])
    ref = oracle.solve(low2, sp)
    np.testing.assert_array_equal(iters[:2].cpu().numpy(), ref["traj_i"][_abi.TI_ITER])
    scale = max(1.0, np.abs(ref["xs"]).max(), np.abs(ref["us"]).max())
    assert np.abs(X[:, :2].cpu().numpy() - ref["xs"]).max() < 1e-6 * scale
    assert np.abs(U[:, :2].cpu().numpy() - ref["us"]).max() < 1e-6 * scale


def test_quasi_static_with_a_rank_deficient_fu(oracle):
    """VSA at q_l = q_m: the stiffness columns of Fu vanish; Crocoddyl's quasiStatic takes the SVD pseudo-inverse
    (minimum-norm update).  GPU against the oracle, the oracle against numpy's pinv."""
    sc = scenarios.two_dof_vsa_boxddp(B=1, T=4)
    problem = crocoddyl.ShootingProblem(sc["x0"][0], sc["running"], sc["terminal"])
    q = np.array([0.3, -0.2])
    x = np.concatenate([q, q, [0.1, -0.05], [0.4, 0.2]])     # q_l = q_m, moving motors
    us = problem.quasiStatic([x] * problem.T)
    u_ref, it = oracle.quasi_static(problem.lowered, 0, x)
    assert it >= 0
    np.testing.assert_allclose(us[0], u_ref, rtol=1e-8, atol=1e-9)
    uu = np.zeros(4)
    for _ in range(100):
        k = oracle.knot(problem.lowered, 0, x, uu)
        assert np.linalg.matrix_rank(k["Fu"]) < 4
        du = -np.linalg.pinv(k["Fu"]).dot(k["xnext"] - x)
        uu = uu + du
        if np.linalg.norm(du) <= 1e-9:
            break
    assert np.abs(uu).max() > 1e-7                     # a non-trivial update (motor inertia 1e-3: micro-newton-metres)
    np.testing.assert_allclose(u_ref, uu, rtol=1e-8, atol=1e-9)


def test_solver_value_function_getters():
    """solver.Vx / solver.Vxx: terminal values are the terminal cost's Lx / Lxx (+ x_reg on the
    diagonal), Vxx is symmetric, and asking for them leaves the gains alone."""
    sc = scenarios.two_dof_sea(B=1, T=12)
    problem = crocoddyl.ShootingProblem(sc["x0"][0], sc["running"], sc["terminal"])
    solver = crocoddyl.SolverDDP(problem)
    solver.th_stop = 1e-7
    solver.solve([], [], 30)
    K0 = [k.copy() for k in solver.K]
    Vxx, Vx = solver.Vxx, solver.Vx
    assert len(Vxx) == 13 and Vxx[0].shape == (8, 8) and Vx[0].shape == (8,)
    for V in Vxx:
        np.testing.assert_allclose(V, V.T, atol=1e-12)
    tdata = problem.terminalData
    np.testing.assert_allclose(Vxx[-1], tdata.Lxx + solver.x_reg * np.eye(8), rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(Vx[-1], tdata.Lx, rtol=1e-12, atol=1e-12)
    for a, b in zip(K0, solver.K):
        np.testing.assert_array_equal(a, b)


def test_rollout_and_problem_calc_api():
    sc = scenarios.two_dof_sea(B=1, T=20)
    problem = crocoddyl.ShootingProblem(sc["x0"][0], sc["running"], sc["terminal"])
    us = [np.array([0.05, -0.02])] * 20
    xs = problem.rollout(us)
    assert len(xs) == 21 and np.array_equal(xs[0], sc["x0"][0])
    total = problem.calc(xs, us)
    assert np.isfinite(total)
    datas = problem.runningDatas.tolist()
    np.testing.assert_allclose(datas[3].xnext, xs[4], atol=1e-13)
    problem.calcDiff(xs, us)
    assert datas[0].Fx.shape == (8, 8) and problem.terminalData.Lxx.shape == (8, 8)


def test_quasi_static_matches_oracle_and_holds_the_state(oracle):
    """ShootingProblem.quasiStatic (examples/two_dof_sea.py:77-78): the SEA example's warm start."""
    sc = scenarios.two_dof_sea(B=1, T=10)
    problem = crocoddyl.ShootingProblem(sc["x0"][0], sc["running"], sc["terminal"])
    rng = np.random.default_rng(3)
    q = rng.uniform(-0.5, 0.5, 2)
    x = np.concatenate([q, q + 0.05, np.zeros(4)])
    us = problem.quasiStatic([x] * problem.T)
    assert len(us) == 10
    u_ref, it = oracle.quasi_static(problem.lowered, 0, x)
    assert it >= 0
    np.testing.assert_allclose(us[0], u_ref, rtol=1e-8, atol=1e-10)
    # Gauss-Newton fixed point: Fu^T (xnext - x) = 0
    k = oracle.knot(problem.lowered, 0, x, us[0])
    assert np.abs(k["Fu"].T.dot(k["xnext"] - x)).max() < 1e-8
    # SEA example flow: warm start, then FDDP (examples/two_dof_sea.py:77-81)
    x0 = sc["x0"][0]
    xs0 = [x0] * (problem.T + 1)
    us0 = problem.quasiStatic([x0] * problem.T)
    solver = crocoddyl.SolverFDDP(problem)
    solver.th_stop = 1e-7
    assert solver.solve(xs0, us0, 100)
    sp = _abi.default_solver_params(_abi.SOLVER_FDDP)
    sp.maxiter, sp.th_stop = 100, 1e-7
    ref = oracle.solve(problem.lowered, sp, xs=np.array(xs0)[:, None, :], us=np.array(us0)[:, None, :])
    assert np.abs(np.array(solver.xs) - ref["xs"][:, 0]).max() < 1e-6
    assert abs(solver.cost - ref["traj_f"][_abi.TF_COST][0]) < 1e-4
