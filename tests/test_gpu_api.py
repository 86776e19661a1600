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


def test_vsa_model_with_stiffness_cost_passes_the_numdiff_acceptance_check():
    """The reference's acceptance technique for its VSA model with a stiffness cost in the stack
    (unittest/test_stiffness_residual.py: analytic Fx, Fu, Lx, Lu against central differences of calc, tolerances
    6.3e-3 and 3e-2), applied to the drop-in classes on the GPU."""
    model = scenarios.two_dof_vsa_modified(B=1, T=1)["running"][0].differential
    rng = np.random.default_rng(0)
    x, u = rng.uniform(-1.0, 1.0, model.state.nx), rng.uniform(0.0, 1.0, model.nu)
    analytic = model.createData()
    model.calc(analytic, x, u)
    model.calcDiff(analytic, x, u)

    def evaluate(field):
        def f(xx, uu):
            d = model.createData()
            model.calc(d, xx, uu)
            return np.array(getattr(d, field), dtype=float, copy=True)
        return f

    acc, cost = evaluate("xout"), evaluate("cost")
    checks = [("Fx", analytic.Fx, _numdiff(lambda z: acc(z, u), x), 6.3e-3),
              ("Fu", analytic.Fu, _numdiff(lambda z: acc(x, z), u), 6.3e-3),
              ("Lx", analytic.Lx, _numdiff(lambda z: cost(z, u), x).ravel(), 3e-2),
              ("Lu", analytic.Lu, _numdiff(lambda z: cost(x, z), u).ravel(), 3e-2)]
    for name, got, fd, tol in checks:
        assert np.allclose(got, fd, atol=tol), name
    assert np.abs(analytic.Fx - checks[0][2]).max() < 1e-4   # and far tighter than the reference asks


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


def test_vsa_boxddp_example_end_to_end_through_the_solver_api(oracle):
    """What examples/two_dof_vsa_boxddp.py does with its solver (T = 100 per BASELINE.json): logger callback, cold
    start, th_stop 1e-7, 400 iterations at most, then the end-effector position and the squared controls -- through
    the drop-in SolverBoxDDP, against the CPU oracle on the same ShootingProblem."""
    sc = scenarios.two_dof_vsa_boxddp(B=1, T=100)
    running = sc["running"][0]
    problem = crocoddyl.ShootingProblem(sc["x0"][0], sc["running"], sc["terminal"])
    solver = crocoddyl.SolverBoxDDP(problem)
    log = crocoddyl.CallbackLogger()
    solver.setCallbacks([log])
    solver.th_stop = sc["th_stop"]
    assert solver.solve([], [], sc["maxiter"]) and solver.stop < sc["th_stop"]
    T = problem.T
    assert len(solver.xs) == T + 1 and len(solver.us) == T and solver.xs[0].shape == (8,)
    assert solver.getCallbacks()[0] is log
    assert len(log.costs) == solver.iter + 1 and log.costs[-1] == pytest.approx(solver.cost)
    assert all(later <= earlier + 1e-9 for earlier, later in zip(log.costs[1:], log.costs[2:]))  # accepted steps only lower it
    assert aslr_to.u_squared(log).shape == (4,)
    us = np.array(solver.us)
    assert (us >= running.u_lb - 1e-12).all() and (us <= running.u_ub + 1e-12).all()
    sp = scenarios.solver_params(sc)
    ref = oracle.solve(problem.lowered, sp, log_cap=sc["maxiter"])
    assert np.abs(np.array(solver.xs) - ref["xs"][:, 0]).max() < 1e-6
    assert np.abs(us - ref["us"][:, 0]).max() < 1e-6
    assert abs(solver.cost - ref["traj_f"][_abi.TF_COST][0]) < 1e-4
    assert solver.iter + 1 == ref["traj_i"][_abi.TI_ITER][0]
    # the logger's series against the oracle's per-iteration values
    n = len(log.costs)
    np.testing.assert_allclose(log.costs, ref["log"][:n, _abi.LOG_COST, 0], rtol=1e-7)
    np.testing.assert_array_equal(log.steps, ref["log"][:n, _abi.LOG_STEP, 0])
    np.testing.assert_array_equal(log.x_regs, ref["log"][:n, _abi.LOG_XREG, 0])
    np.testing.assert_allclose(log.stops, ref["log"][:n, _abi.LOG_STOP, 0], rtol=1e-4, atol=1e-12)
    # the script's last print: where the end effector ended up (examples/two_dof_vsa_boxddp.py:83-84)
    model = running.state.pinocchio
    fid = model.getFrameId("EE")
    reached = solver.problem.terminalData.differential.multibody.pinocchio.oMf[fid].translation
    fr = model.frames[fid]
    expect = oracle.frame_placement(problem.lowered.desc.chain, solver.xs[-1][:2], fr.parent, fr.placement.rotation,
                                    fr.placement.translation)[1]
    np.testing.assert_allclose(reached, expect, atol=1e-12)
    assert np.abs(reached - np.array([.01, .2, .18])).max() < 0.1   # towards the target, as far as the regularisers let it


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


def test_frame_placements_and_residuals_of_node_data_match_the_oracle(oracle):
    """runningDatas[t] / terminalData: .differential.multibody.pinocchio.oMf[frame] (examples/two_dof_sea.py:82-86) and
    data.r (integrated_action.py:17-18) come from the GPU; checked against the oracle on the 2-DoF and 7-DoF chains,
    single problem and batch."""
    import torch
    for name, kw in (("two_dof_sea", dict(B=1, T=6)), ("talos_arm_sea", dict(B=3, T=4, seed=2))):
        sc = scenarios.SCENARIOS[name](**kw)
        B = sc["x0"].shape[0]
        problem = crocoddyl.ShootingProblem(sc["x0"][0] if B == 1 else sc["x0"], sc["running"], sc["terminal"],
                                            frame_refs=None if B == 1 else sc["frame_refs"])
        rng = np.random.default_rng(4)
        nj = problem.nx // 4
        xs = rng.uniform(-0.8, 0.8, (B, problem.T + 1, problem.nx))
        us = rng.uniform(-0.5, 0.5, (B, problem.T, problem.nu))
        problem.calc(xs[0] if B == 1 else torch.as_tensor(xs), us[0] if B == 1 else torch.as_tensor(us))
        model = sc["running"][0].state.pinocchio
        chain = problem.lowered.desc.chain
        for fid, fr in enumerate(model.frames):
            for t in (0, problem.T):
                node = problem.runningDatas.tolist()[t] if t < problem.T else problem.terminalData
                M = node.differential.multibody.pinocchio.oMf[fid]
                if fr.parent < 0:   # the universe frame does not move
                    np.testing.assert_array_equal(M.translation, fr.placement.translation)
                    continue
                for b in range(B):
                    R = M.rotation if B == 1 else M.rotation[b].cpu().numpy()
                    p = M.translation if B == 1 else M.translation[b].cpu().numpy()
                    Rr, pr = oracle.frame_placement(chain, xs[b, t, :nj], fr.parent, fr.placement.rotation,
                                                    fr.placement.translation)
                    np.testing.assert_allclose(R, Rr, atol=1e-12)
                    np.testing.assert_allclose(p, pr, atol=1e-12)
        # data.r of a running node and of the terminal node (trajectory 0), in Crocoddyl's name order
        lowm = problem.lowered
        for t in (1, problem.T):
            node = problem.runningDatas.tolist()[t] if t < problem.T else problem.terminalData
            iam = sc["running"][t] if t < problem.T else sc["terminal"]
            u = us[0, t] if t < problem.T else iam.differential._default_u()
            fref = None if lowm.frame_ref is None else lowm.frame_ref[0]
            raw = oracle.dam_residuals(lowm, int(lowm.node_model[t]), xs[0, t], u, frame_ref=fref)
            expect = iam.differential.costs.order_residuals(raw, problem.nx, problem.nu)
            assert expect.size == iam.differential.costs.nr and expect.size > 0
            np.testing.assert_allclose(node.r, expect, rtol=1e-10, atol=1e-12)


def test_model_level_data_r_and_omf(oracle):
    """model.calc(data, x, u) fills data.r (stacked residuals, alphabetical cost-name order like Crocoddyl's
    CostModelSum) and data.multibody.pinocchio.oMf; the integrated model copies r (integrated_action.py:17-18)."""
    sc = scenarios.two_dof_vsa_modified(B=1, T=1)
    iam = sc["running"][0]
    dam = iam.differential
    rng = np.random.default_rng(8)
    x, u = rng.uniform(-1, 1, 8), rng.uniform(0.1, 1, 4)
    d = dam.createData()
    dam.calc(d, x, u)
    low = scenarios.lower(sc)
    raw = oracle.dam_residuals(low, 0, x, u, frame_ref=None)
    # insertion order gripperPose(6) xReg(8) uReg(4) vsa(2) -> name order gripperPose, uReg, vsa, xReg
    expect = np.concatenate([raw[0:6], raw[14:18], raw[18:20], raw[6:14]])
    np.testing.assert_allclose(d.r, expect, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(raw[18:20], 10 * (u[2:] - 0.002), rtol=1e-14)      # stiffness_cost.py:15
    np.testing.assert_allclose(raw[6:14], x, rtol=0, atol=0)                        # state residual about 0
    model = dam.state.pinocchio
    fid = model.getFrameId("EE")
    fr = model.frames[fid]
    Rr, pr = oracle.frame_placement(low.desc.chain, x[:2], fr.parent, fr.placement.rotation, fr.placement.translation)
    np.testing.assert_allclose(d.multibody.pinocchio.oMf[fid].translation, pr, atol=1e-12)
    np.testing.assert_allclose(d.pinocchio.oMf[fid].rotation, Rr, atol=1e-12)
    di = iam.createData()
    iam.calc(di, x, u)
    np.testing.assert_allclose(di.r, expect, rtol=1e-10, atol=1e-12)


def test_batched_logger_matches_the_oracle_iteration_by_iteration(oracle, tmp_path):
    """CallbackLogger / CallbackVerbose on a batch of 8: the per-iteration series come from the device-resident log
    (no host round trip per iteration) and equal the oracle's per-iteration values; export_solution writes the arrays
    examples/two_dof_vsa_boxddp.py:104-127 saves."""
    import io
    sc = scenarios.two_dof_vsa_boxddp(B=8, T=40, seed=1)
    problem = crocoddyl.ShootingProblem(sc["x0"], sc["running"], sc["terminal"], frame_refs=sc["frame_refs"])
    solver = crocoddyl.SolverBoxDDP(problem)
    solver.th_stop = sc["th_stop"]
    log, table = crocoddyl.CallbackLogger(), io.StringIO()
    solver.setCallbacks([log, crocoddyl.CallbackVerbose(table)])
    solver.solve([], [], 60)
    ref = oracle.solve(problem.lowered, scenarios.solver_params(sc, maxiter=60), log_cap=60)
    iters = ref["traj_i"][_abi.TI_ITER]
    np.testing.assert_array_equal(log.iters, iters)
    n = int(iters.max())
    assert log.costs.shape == (n, 8)
    rl = ref["log"][:n]
    np.testing.assert_array_equal(np.isnan(log.costs), np.isnan(rl[:, _abi.LOG_COST]))
    for b in range(8):
        k = int(iters[b])
        assert np.isnan(log.costs[k:, b]).all() and not np.isnan(log.costs[:k, b]).any()
    on = ~np.isnan(rl[:, _abi.LOG_COST])
    np.testing.assert_allclose(log.costs[on], rl[:, _abi.LOG_COST][on], rtol=1e-7)
    np.testing.assert_array_equal(log.steps[on], rl[:, _abi.LOG_STEP][on])
    np.testing.assert_array_equal(log.x_regs[on], rl[:, _abi.LOG_XREG][on])
    np.testing.assert_allclose(log.stops[on], rl[:, _abi.LOG_STOP][on], rtol=1e-4, atol=1e-12)
    np.testing.assert_allclose(log.grads[on], -rl[:, _abi.LOG_D2][on], rtol=1e-4, atol=1e-12)
    full = solver.iteration_log()
    np.testing.assert_array_equal(full[:, _abi.LOG_ACCEPTED][on], rl[:, _abi.LOG_ACCEPTED][on])
    import _parity
    # (the logged status word is cumulative: the decision / outcome bits iteration by iteration, the overflow note of
    #  _parity.assert_status_words_match on the final words)
    keep = ~np.int64(_abi.ST_FORWARD_ERR)
    np.testing.assert_array_equal(full[:, _abi.LOG_STATUS][on].astype(np.int64) & keep,
                                  rl[:, _abi.LOG_STATUS][on].astype(np.int64) & keep)
    _parity.assert_status_words_match(solver.status.cpu().numpy(), ref["traj_i"][_abi.TI_STATUS])
    np.testing.assert_array_equal(full[:, _abi.LOG_FEASIBLE][on], rl[:, _abi.LOG_FEASIBLE][on])
    rows = table.getvalue().splitlines()
    assert rows[0].split()[:2] == ["iter", "active"] and len([r for r in rows if r.split()[0].isdigit()]) == n
    # export: q, u, stiffness, t of one trajectory and of the batch
    path = solver.export_solution(str(tmp_path / "sol.npz"), trajectory=3)
    z = np.load(path)
    X, U = solver.xs.cpu().numpy(), solver.us.cpu().numpy()
    np.testing.assert_array_equal(z["q"], X[3][:, :2])
    np.testing.assert_array_equal(z["u"], U[3][:, :2])
    np.testing.assert_array_equal(z["stiffness"], U[3][:, 2:])
    np.testing.assert_allclose(z["t"], np.arange(40) * 1e-2)
    assert np.load(solver.export_solution(str(tmp_path / "all.npz"), trajectory=None))["q"].shape == (8, 41, 2)


def test_verbose_table_of_a_single_problem_and_logger_grads(oracle):
    """B = 1: Crocoddyl's verbose table (header every 10 iterations; iter, cost, stop, grad = -d[1], xreg, ureg, step,
    feas) and CallbackLogger.grads = -expectedImprovement()[1], row by row against the oracle's log."""
    import io
    sc = scenarios.two_dof_sea(B=1, T=30)
    problem = crocoddyl.ShootingProblem(sc["x0"][0], sc["running"], sc["terminal"])
    solver = crocoddyl.SolverFDDP(problem)
    solver.th_stop = sc["th_stop"]
    out, log = io.StringIO(), crocoddyl.CallbackLogger()
    solver.setCallbacks([log, crocoddyl.CallbackVerbose(out)])
    solver.solve([], [], 25)
    ref = oracle.solve(problem.lowered, scenarios.solver_params(sc, solver="SolverFDDP", maxiter=25), log_cap=25)
    n = int(ref["traj_i"][_abi.TI_ITER][0])
    assert log.iters == list(range(n))
    np.testing.assert_allclose(log.grads, -ref["log"][:n, _abi.LOG_D2, 0], rtol=1e-6, atol=1e-14)
    np.testing.assert_allclose(log.costs, ref["log"][:n, _abi.LOG_COST, 0], rtol=1e-9)
    lines = out.getvalue().splitlines()
    headers = [i for i, l in enumerate(lines) if l.startswith("iter")]
    assert headers == [11 * k for k in range((n + 9) // 10)]
    rows = [l.split() for l in lines if not l.startswith("iter")]
    assert len(rows) == n and [int(r[0]) for r in rows] == list(range(n))
    for r, k in zip(rows, range(n)):
        assert float(r[1]) == pytest.approx(ref["log"][k, _abi.LOG_COST, 0], rel=2e-5)
        assert float(r[6]) == pytest.approx(ref["log"][k, _abi.LOG_STEP, 0], abs=1e-4)
        assert int(r[7]) == int(ref["log"][k, _abi.LOG_FEASIBLE, 0])


def test_rollout_leaves_a_solve_in_progress_untouched():
    """ShootingProblem.rollout() uses the forward kernel with zero gains: the gains, the candidate and the feasibility
    flags it overwrites are put back."""
    import torch
    sc = scenarios.two_dof_sea(B=1, T=15)
    problem = crocoddyl.ShootingProblem(sc["x0"][0], sc["running"], sc["terminal"])
    solver = crocoddyl.SolverDDP(problem)
    solver.solve([], [], 4)
    e = problem.engine
    before = {r: e.region(r).clone() for r in (_abi.R_XS, _abi.R_US, _abi.R_KGAIN, _abi.R_KFF)}
    feas = e.traj_i(_abi.TI_FEASIBLE).clone()
    xs = problem.rollout([np.array([0.01, 0.02])] * 15)
    assert len(xs) == 16
    for r, t in before.items():
        assert torch.equal(e.region(r), t)
    assert torch.equal(e.traj_i(_abi.TI_FEASIBLE), feas)


def test_engine_runs_on_its_own_device_whatever_the_current_one():
    """Every ABI call is made with the engine's device current (the library launches on the calling thread's device)."""
    import torch
    from aslr_to_amd.engine import Engine
    sc = scenarios.two_dof_sea(B=4, T=5)
    e = Engine(scenarios.lower(sc), "cuda:0")
    assert e.device == torch.device("cuda", 0)
    e.set_candidate(None, None)
    e.calc_diff()
    torch.cuda.synchronize()
    assert torch.isfinite(e.region(_abi.R_COST)).all()


@pytest.mark.parametrize("name, solver, T", [("two_dof_vsa_boxddp", "SolverBoxDDP", 100), ("two_dof_sea", "SolverFDDP", 100),
                                             ("two_dof_sea", "SolverFDDP", 17), ("two_dof_vsa_boxddp", "SolverBoxDDP", 33)])
def test_the_two_launch_forward_pass_changes_the_schedule_not_the_results(monkeypatch, name, solver, T):
    """Default: the rollout stops at mid-horizon and a second launch continues it while its other blocks evaluate the trial
    costs of the first half (rollout_and_cost_kernel).  ASLR_PIPELINE=0 (read when the handle is created) is the plain
    sequence; both must give the same bits -- states, controls, gains, solver state -- with and without sub-shards."""
    import torch
    from aslr_to_amd.engine import Engine
    res = {}
    for pl in ("0", "1", "3"):   # plain sequence, two segments (default), three
        monkeypatch.setenv("ASLR_PIPELINE", pl)
        sc = scenarios.SCENARIOS[name](B=300, T=T, seed=2)   # (odd horizons: the seam is at T // 2)
        e = Engine(scenarios.lower(sc))
        e.set_candidate(None, None)
        e.set_subshards(2 if pl == "1" else 1)
        sp = scenarios.solver_params(sc, solver=solver, fixed_iterations=1, maxiter=12)
        e.iterate_n(sp, True, 12)
        torch.cuda.synchronize()
        res[pl] = [e.region(r).clone() for r in (_abi.R_XS, _abi.R_US, _abi.R_KGAIN, _abi.R_TRAJ_F, _abi.R_TRAJ_I, _abi.R_COST_TRY)]
    for other in ("1", "3"):
        for a, b in zip(res["0"], res[other]):  # (bit patterns: failed candidates carry NaN)
            ia, ib = (t.view(torch.int64) if t.dtype == torch.float64 else t for t in (a, b))
            assert torch.equal(ia, ib)
    assert int(res["1"][4][_abi.TI_ITER].min()) == 12


def test_subshards_change_the_schedule_not_the_results():
    """aslr_set_subshards: the shard iterated as 1, 2, 3, 4 sub-shards on internal streams gives the same bits
    (trajectories are independent), for a batch that does not divide into 64-trajectory blocks evenly, through
    aslr_iterate_n and through aslr_solve with convergence polling."""
    import torch
    from aslr_to_amd.engine import Engine
    sc = scenarios.two_dof_vsa_boxddp(B=333, T=25, seed=9)
    low = scenarios.lower(sc)
    sp_fixed = scenarios.solver_params(sc, fixed_iterations=1, maxiter=12)
    sp_conv = scenarios.solver_params(sc, maxiter=40)
    ref = None
    for n in (1, 2, 3, 4):
        e = Engine(low)
        e.set_subshards(n)
        e.set_candidate(None, None)
        e.iterate_n(sp_fixed, True, 5)
        e.iterate_n(sp_fixed, False, 7)
        e.finalize()
        torch.cuda.synchronize()
        a = (e.region(_abi.R_XS).clone(), e.region(_abi.R_US).clone(), e.region(_abi.R_TRAJ_I).clone(),
             e.region(_abi.R_TRAJ_F).clone())
        e.set_candidate(None, None)
        iters = e.solve(sp_conv, poll_every=3)
        torch.cuda.synchronize()
        b = (e.region(_abi.R_XS).clone(), e.region(_abi.R_US).clone(), e.traj_i(_abi.TI_ITER).clone(),
             e.traj_i(_abi.TI_STATUS).clone(), iters)
        if ref is None:
            ref = (a, b)
            assert int(a[2][_abi.TI_ITER].min()) == 12
            continue
        same = lambda x, y: torch.equal(torch.nan_to_num(x), torch.nan_to_num(y)) if x.is_floating_point() else torch.equal(x, y)
        for x, y in zip(ref[0], a):   # (TRAJ_F holds NaN for failed line-search trials)
            assert same(x, y), n
        for x, y in zip(ref[1][:4], b[:4]):
            assert same(x, y), n
        assert ref[1][4] == b[4]


def test_vsa_on_the_seven_joint_chain_at_the_model_level(oracle):
    """VSA actuation on the 7-joint arm (nx = 28, nu = 14): calc / calcDiff of the model and of a shooting problem run on
    the GPU and match the oracle; the analytic Fx, Fu, Lx, Lu pass the reference's numdiff acceptance technique
    (unittest/test_free_placementcost_free_fwddyn.py:34-46); the solver entry points decline with a message."""
    model = example_robot_data.load('talos_arm').model
    state = aslr_to.StateMultibodyASR(model)
    actuation = aslr_to.VSAASRActuation(state)
    nu = 2 * actuation.nu
    assert nu == 14
    costs = crocoddyl.CostModelSum(state, nu)
    reach = aslr_to.ResidualModelFramePlacementASR(state, model.getFrameId("gripper_left_joint"),
                                                   pinocchio.SE3(np.eye(3), np.array([0.1, 0.2, -0.3])), nu)
    costs.addCost("reach", crocoddyl.CostModelResidual(state, reach), 2.0)
    costs.addCost("effort", crocoddyl.CostModelResidual(state, crocoddyl.ResidualModelControl(state, nu)), 1e-2)
    dam = aslr_to.DifferentialFreeFwdDynamicsModelVSA(state, actuation, costs)
    running = aslr_to.IntegratedActionModelEulerASR(dam, 1e-2)
    terminal = aslr_to.IntegratedActionModelEulerASR(dam, 0.0)
    rng = np.random.default_rng(21)
    x = rng.uniform(-0.6, 0.6, 28)
    u = np.concatenate([rng.uniform(-1, 1, 7), rng.uniform(0.5, 3.0, 7)])
    d = dam.createData()
    dam.calc(d, x, u)
    dam.calcDiff(d, x, u)
    problem = crocoddyl.ShootingProblem(np.zeros(28), [running] * 3, terminal)
    ref = oracle.dam(problem.lowered, 0, x, u)
    for k in ("xout", "Fx", "Fu", "Lx", "Lu", "Lxx", "Luu"):
        scale = max(1.0, np.abs(ref[k]).max())
        assert np.abs(getattr(d, k) - ref[k]).max() < 1e-9 * scale, k
    assert d.cost == pytest.approx(ref["cost"], rel=1e-11)

    def acc(xx, uu):
        dd = dam.createData()
        dam.calc(dd, xx, uu)
        return dd.xout.copy(), dd.cost
    assert np.allclose(d.Fx, _numdiff(lambda z: acc(z, u)[0], x), atol=6.3e-3)
    assert np.allclose(d.Fu, _numdiff(lambda z: acc(x, z)[0], u), atol=6.3e-3)
    assert np.allclose(d.Lx, _numdiff(lambda z: acc(z, u)[1], x).ravel(), atol=3e-2)
    assert np.allclose(d.Lu, _numdiff(lambda z: acc(x, z)[1], u).ravel(), atol=3e-2)
    # the shooting-problem sweeps
    xs = rng.uniform(-0.3, 0.3, (4, 28))
    us = np.concatenate([rng.uniform(-1, 1, (3, 7)), rng.uniform(0.5, 3.0, (3, 7))], axis=1)
    total = problem.calcDiff(xs, us)
    kr = [oracle.knot(problem.lowered, 0, xs[t], us[t]) for t in range(3)]
    assert total == pytest.approx(sum(k["cost"] for k in kr) + oracle.knot(problem.lowered, 1, xs[3], None)["cost"], rel=1e-11)
    datas = problem.runningDatas.tolist()
    for t in range(3):
        np.testing.assert_allclose(datas[t].xnext, kr[t]["xnext"], atol=1e-11)
        assert np.abs(datas[t].Fx - kr[t]["Fx"]).max() < 1e-9 * max(1.0, np.abs(kr[t]["Fx"]).max())
        assert np.abs(datas[t].Fu - kr[t]["Fu"]).max() < 1e-9 * max(1.0, np.abs(kr[t]["Fu"]).max())
    with pytest.raises(_abi.AslrError, match="model-level"):
        crocoddyl.SolverDDP(problem).solve([], [], 3)


def test_pool_solve_gives_every_problem_the_solve_it_would_get_in_a_batch():
    """aslr_solve_pool: 300 problems streamed through 64 slots (stopped slots flushed and refilled on the device), each
    cold-started and iterated to its own stop, against ONE lock-step batch solve of the same 300: identical bits per
    problem, whatever the refill period and the sub-shard count; fewer lock-step iterations than 5 waves of 64 would
    need back to back."""
    import torch
    from aslr_to_amd.engine import Engine
    P, T, maxiter = 300, 30, 80
    sc = scenarios.two_dof_vsa_boxddp(B=P, T=T, seed=4)
    sp = scenarios.solver_params(sc, maxiter=maxiter)
    full = Engine(scenarios.lower(sc))
    full.set_candidate(None, None)
    full.solve(sp, poll_every=8)
    torch.cuda.synchronize()
    X = full.region(_abi.R_XS).permute(1, 0, 2).contiguous()
    U = full.region(_abi.R_US).permute(1, 0, 2).contiguous()
    iters, status = full.traj_i(_abi.TI_ITER).clone(), full.traj_i(_abi.TI_STATUS).clone()
    cost = full.traj_f(_abi.TF_COST).clone()
    assert int(iters.min()) < int(iters.max())          # the problems really need different numbers of iterations
    slots = dict(sc)
    slots["x0"], slots["frame_refs"] = sc["x0"][:64], sc["frame_refs"][:64]
    e = Engine(scenarios.lower(slots))
    for refill_every, nsub in ((1, 1), (4, 1), (3, 2)):
        e.set_subshards(nsub)
        r = e.solve_pool(sc["x0"], sc["frame_refs"], sp, refill_every=refill_every, poll_every=8)
        assert torch.equal(r["iters"], iters) and torch.equal(r["status"], status)
        assert torch.equal(r["xs"], X) and torch.equal(r["us"], U)
        assert torch.equal(r["cost"], cost)
        waves = -(-P // 64)
        assert r["batch_iters"] < waves * int(iters.max())
    # warm-started pool (per-problem initial guesses) against the same warm start as one batch
    rng = np.random.default_rng(3)
    xs0 = np.repeat(sc["x0"][:, None, :], T + 1, axis=1) + 1e-2 * rng.standard_normal((P, T + 1, 8))
    us0 = np.abs(1e-1 * rng.standard_normal((P, T, 4)))
    full.set_candidate(xs0, us0)
    full.solve(sp, poll_every=8)
    torch.cuda.synchronize()
    rw = e.solve_pool(sc["x0"], sc["frame_refs"], sp, refill_every=3, poll_every=9, xs_init=xs0, us_init=us0)
    assert torch.equal(rw["xs"], full.region(_abi.R_XS).permute(1, 0, 2).contiguous())
    assert torch.equal(rw["iters"], full.traj_i(_abi.TI_ITER))
    # a pool smaller than the slots, and a solver without bounds on the same engine afterwards
    r = e.solve_pool(sc["x0"][:10], sc["frame_refs"][:10], sp)
    assert torch.equal(r["xs"], X[:10]) and torch.equal(r["iters"], iters[:10])
    # a short maxiter (fewer iterations than the polling period) still terminates with every problem flushed
    sp3 = scenarios.solver_params(sc, maxiter=3)
    r3 = e.solve_pool(sc["x0"][:100], sc["frame_refs"][:100], sp3, refill_every=4, poll_every=16)
    assert int(r3["iters"].max()) == 3 and r3["batch_iters"] <= 3 * (3 + 4)
    # the handle's OWN problems (x0 / targets it was created with) are back after a pool: a plain solve on it equals the
    # same solve on a fresh engine, bit for bit (aslr_solve_pool used to leave the last pool problems in the slots)
    e.set_subshards(1)
    e.set_candidate(None, None)
    e.solve(sp, poll_every=8)
    fresh = Engine(scenarios.lower(slots))
    fresh.set_candidate(None, None)
    fresh.solve(sp, poll_every=8)
    torch.cuda.synchronize()
    assert torch.equal(e.region(_abi.R_X0), fresh.region(_abi.R_X0))
    assert torch.equal(e.region(_abi.R_XS), fresh.region(_abi.R_XS)) and torch.equal(e.region(_abi.R_US), fresh.region(_abi.R_US))
    assert torch.equal(e.traj_i(_abi.TI_ITER), fresh.traj_i(_abi.TI_ITER))


@pytest.mark.parametrize("name,solver,P,slots,T,maxiter", [("two_dof_sea", "SolverFDDP", 150, 40, 25, 60),
                                                          ("talos_arm_sea", "SolverFDDP", 14, 6, 20, 12)])
def test_pool_solve_on_the_other_models_and_solvers(name, solver, P, slots, T, maxiter):
    """The pool solve with gap-aware FDDP iterations (slots start infeasible) on the 2-DoF SEA arm and on the 7-joint chain
    (block / team kernels, plain candidate slabs): bit-identical to one batch solve of the same problems."""
    import torch
    from aslr_to_amd.engine import Engine
    sc = scenarios.SCENARIOS[name](B=P, T=T, seed=6)
    sp = scenarios.solver_params(sc, solver=solver, maxiter=maxiter)
    full = Engine(scenarios.lower(sc))
    full.set_candidate(None, None)
    full.solve(sp, poll_every=4)
    torch.cuda.synchronize()
    sub = dict(sc)
    sub["x0"], sub["frame_refs"] = sc["x0"][:slots], sc["frame_refs"][:slots]
    e = Engine(scenarios.lower(sub))
    r = e.solve_pool(sc["x0"], sc["frame_refs"], sp, refill_every=2, poll_every=4)
    assert torch.equal(r["iters"], full.traj_i(_abi.TI_ITER)) and torch.equal(r["status"], full.traj_i(_abi.TI_STATUS))
    assert torch.equal(r["xs"], full.region(_abi.R_XS).permute(1, 0, 2).contiguous())
    assert torch.equal(r["us"], full.region(_abi.R_US).permute(1, 0, 2).contiguous())
    assert torch.equal(r["cost"], full.traj_f(_abi.TF_COST))


def test_pendulum_with_one_motor_command_solves_like_the_two_command_model():
    """ActuationModelDoublePendulum(state, actLink=0, nu=1) (python/aslr_to/__init__.py:279-281) next to the nu = 2 form
    the C1 scenario uses (second command drives nothing, zero weight): the same xs, the same motor command, the same
    iteration count -- and every control-sized quantity the solver / the data objects expose has ONE column."""
    import torch
    res = {}
    for name in ("double_pendulum", "double_pendulum_nu1"):
        sc = scenarios.SCENARIOS[name](T=100)
        problem = crocoddyl.ShootingProblem(sc["x0"][0], sc["running"], sc["terminal"])
        solver = crocoddyl.SolverDDP(problem)
        solver.th_stop = 1e-9
        solver.solve([], [], 60)
        torch.cuda.synchronize()
        res[name] = dict(xs=np.array(solver.xs), us=np.array(solver.us), K=np.array(solver.K), k=np.array(solver.k),
                         Qu=np.array(solver.Qu), iters=solver.iterations, cost=solver.cost, problem=problem, solver=solver)
    a, b = res["double_pendulum"], res["double_pendulum_nu1"]
    assert b["us"].shape == (100, 1) and b["K"].shape == (100, 1, 8) and b["k"].shape == (100, 1) and b["Qu"].shape == (100, 1)
    assert a["us"].shape == (100, 2) and np.abs(a["us"][:, 1]).max() == 0.0      # the idle command of the nu = 2 form stays at zero
    assert a["iters"] == b["iters"]
    np.testing.assert_allclose(b["xs"], a["xs"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(b["us"][:, 0], a["us"][:, 0], rtol=0, atol=1e-9)
    np.testing.assert_allclose(b["K"][:, 0], a["K"][:, 0], rtol=0, atol=1e-7 * (1 + np.abs(a["K"]).max()))
    assert abs(a["cost"] - b["cost"]) < 1e-9 * max(1.0, abs(a["cost"]))
    # warm start with nu = 1 controls, node data, model-level calc / calcDiff
    p = b["problem"]
    c = p.calc(list(b["xs"]), list(b["us"]))
    assert abs(c - b["cost"]) < 1e-8 * max(1.0, abs(c))
    p.calcDiff(list(b["xs"]), list(b["us"]))
    d = p.runningDatas[3]
    assert d.Fu.shape == (8, 1) and d.Lu.shape == (1,) and d.Luu.shape == (1, 1) and d.Lxu.shape == (8, 1)
    pa = a["problem"]
    pa.calcDiff(list(a["xs"]), list(a["us"]))
    np.testing.assert_allclose(d.Fu[:, 0], pa.runningDatas[3].Fu[:, 0], rtol=0, atol=1e-9)
    m = p.runningModels[0]
    data = m.createData()
    m.calc(data, b["xs"][5], b["us"][5])
    m.calcDiff(data, b["xs"][5], b["us"][5])
    assert data.Fu.shape == (8, 1) and np.isfinite(data.Fu).all() and data.r.shape[0] == m.nr
    roll = p.rollout(list(b["us"]))
    np.testing.assert_allclose(np.array(roll), b["xs"], rtol=0, atol=1e-8)


def test_pool_smaller_than_the_slots_on_a_fresh_handle_and_a_regularisation_that_cannot_grow():
    """Two ways a backward sweep could spin for ever, both found as a hang of the test suite in round 3: (1) a pool with fewer
    problems than slots on a handle that has never solved anything -- the slots that never receive a problem must be
    marked idle by the first refill even when their block starts after the counter has passed P (they used to keep
    DONE = 0 and were iterated with x_reg = 0); (2) reg_init = 0 with an indefinite Quu: the retry loop multiplies the
    regularisation by reg_incfactor, which never reaches reg_max from zero -- it now counts as the ceiling."""
    import torch
    from aslr_to_amd.engine import Engine
    sc = scenarios.two_dof_vsa_boxddp(B=70, T=5, seed=1)
    for P, maxiter in ((20, 80), (20, 12), (3, 30)):
        e = Engine(scenarios.lower(sc))                     # fresh: TRAJ_F / TRAJ_I all zero
        r = e.solve_pool(sc["x0"][:P], sc["frame_refs"][:P], scenarios.solver_params(sc, maxiter=maxiter))
        torch.cuda.synchronize()
        assert int(r["iters"].min()) >= 1 and int(r["iters"].max()) <= maxiter
        full = Engine(scenarios.lower(dict(sc, x0=sc["x0"][:P], frame_refs=sc["frame_refs"][:P])))
        full.set_candidate(None, None)
        full.solve(scenarios.solver_params(sc, maxiter=maxiter), poll_every=4)
        torch.cuda.synchronize()
        assert torch.equal(r["iters"], full.traj_i(_abi.TI_ITER)) and torch.equal(r["xs"], full.region(_abi.R_XS).permute(1, 0, 2).contiguous())
    # (2) a regularisation that cannot grow gives up at once instead of spinning: zero weights everywhere make Quu = 0
    sc0 = scenarios.two_dof_vsa_boxddp(B=8, T=5, seed=1)
    for m in (sc0["running"][0], sc0["terminal"]):
        for name in list(m.differential.costs.costs):
            m.differential.costs.costs[name].weight = 0.0
    e = Engine(scenarios.lower(sc0))
    sp = scenarios.solver_params(sc0, maxiter=5)
    sp.reg_init = 0.0
    sp.reg_min = 0.0
    e.set_candidate(None, None)
    e.solve(sp, poll_every=1)
    torch.cuda.synchronize()
    st = e.traj_i(_abi.TI_STATUS).cpu().numpy()
    assert ((st & _abi.ST_REG_MAX) != 0).all() and int(e.traj_i(_abi.TI_ITER).max()) <= 5
