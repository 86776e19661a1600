"""GPU parity tests proper: the HIP path (through the C ABI) against the CPU oracle on the same
seeded inputs.  Float64 throughout; tolerances are written at each assert.

north_star tolerance for solver results: 1e-6 on xs/us, 1e-4 on final cost.  Per-kernel outputs are
held to ~1e-9 relative (they differ from the oracle only by FMA contraction / libm rounding).
"""
import ctypes as C

import numpy as np
import pytest

from aslr_to_amd import _abi, scenarios

pytestmark = pytest.mark.gpu


def _engine(low):
    from aslr_to_amd.engine import Engine
    return Engine(low)


def _sync():
    import torch
    torch.cuda.synchronize()


def _np(t):
    return t.detach().cpu().numpy()


def _relerr(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return np.max(np.abs(a - b) / (1.0 + np.abs(b))) if a.size else 0.0


CASES = [("two_dof_vsa_boxddp", dict(B=70, T=5)), ("two_dof_sea", dict(B=70, T=5)),
         ("double_pendulum", dict(T=6)), ("talos_arm_sea", dict(B=5, T=3))]


def _random_candidate(low, seed):
    rng = np.random.default_rng(seed)
    xs = rng.uniform(-0.8, 0.8, (low.T + 1, low.B, low.nx))
    us = rng.uniform(-1.0, 1.0, (low.T, low.B, low.nu))
    if low.dam == _abi.DAM_VSA:
        us[..., low.nu // 2:] = rng.uniform(0.1, 5.0, (low.T, low.B, low.nu // 2))
    return xs, us


@pytest.mark.parametrize("name,kw", CASES)
def test_calc_and_calcdiff_match_oracle(oracle, name, kw):
    import torch
    sc = scenarios.SCENARIOS[name](**kw)
    low = scenarios.lower(sc)
    e = _engine(low)
    xs, us = _random_candidate(low, 1)
    e.region(_abi.R_XS).copy_(torch.as_tensor(xs))
    e.region(_abi.R_US).copy_(torch.as_tensor(us))
    e.calc_diff()
    _sync()
    xnext, cost, deriv = oracle.calc_diff(low, xs, us)
    assert _relerr(_np(e.region(_abi.R_XNEXT)), xnext) < 1e-11
    assert _relerr(_np(e.region(_abi.R_COST)), cost) < 1e-11
    g = _np(e.region(_abi.R_DERIV))
    err = _relerr(g, deriv)
    assert err < 1e-9, "DERIV record mismatch %g" % err
    # calc alone writes the same xnext / cost
    e.region(_abi.R_XNEXT).zero_()
    e.region(_abi.R_COST).zero_()
    e.calc()
    _sync()
    assert _relerr(_np(e.region(_abi.R_XNEXT)), xnext) < 1e-11
    assert _relerr(_np(e.region(_abi.R_COST)), cost) < 1e-11
    # a second sweep at another point: the record chunks that are structural zeros or depend on the model only are
    # not rewritten by it (DERIV was zero-filled at creation, the first sweep put the cost-weight diagonals in place)
    xs2, us2 = _random_candidate(low, 5)
    e.region(_abi.R_XS).copy_(torch.as_tensor(xs2))
    e.region(_abi.R_US).copy_(torch.as_tensor(us2))
    e.calc_diff()
    _sync()
    _, _, deriv2 = oracle.calc_diff(low, xs2, us2)
    assert np.abs(deriv2 - deriv).max() > 1e-3        # the point really changed
    err = _relerr(_np(e.region(_abi.R_DERIV)), deriv2)
    assert err < 1e-9, "DERIV record mismatch on the second sweep %g" % err


def _backward_inputs(oracle, low, seed):
    xs, us = _random_candidate(low, seed)
    _, _, deriv = oracle.calc_diff(low, xs, us)
    rng = np.random.default_rng(seed + 7)
    gaps = rng.uniform(-0.05, 0.05, (low.T + 1, low.B, low.nx))
    return xs, us, deriv, gaps


@pytest.mark.parametrize("name,kw", CASES)
@pytest.mark.parametrize("feasible", [0, 1])
@pytest.mark.parametrize("hs", [0, 1, 2, 4])
def test_backward_pass_matches_oracle(oracle, monkeypatch, name, kw, feasible, hs):
    import torch
    if hs:   # 0: the default decomposition of the size (block-per-trajectory LDS kernel at nx = 28)
        monkeypatch.setenv("ASLR_BWD_HS", str(hs))
    else:
        monkeypatch.delenv("ASLR_BWD_HS", raising=False)
    sc = scenarios.SCENARIOS[name](**kw)
    low = scenarios.lower(sc)
    sp = scenarios.solver_params(sc)
    e = _engine(low)
    xs, us, deriv, gaps = _backward_inputs(oracle, low, 3)
    xreg = 1e-3
    e.region(_abi.R_US).copy_(torch.as_tensor(us))
    e.region(_abi.R_DERIV).copy_(torch.as_tensor(deriv))
    e.region(_abi.R_GAPS).copy_(torch.as_tensor(gaps))
    e.region(_abi.R_KFF).zero_()
    e.region(_abi.R_TRAJ_F)[_abi.TF_XREG].fill_(xreg)
    e.region(_abi.R_TRAJ_I)[_abi.TI_FEASIBLE].fill_(feasible)
    e.region(_abi.R_TRAJ_I)[_abi.TI_STATUS].fill_(0)
    e.backward_pass(sp)
    _sync()
    ref = oracle.backward_pass(low, sp, deriv, gaps, us, xreg, feasible)
    assert not ref["fail"].any()
    assert (_np(e.traj_i(_abi.TI_STATUS)) & _abi.ST_BACKWARD_ERR == 0).all()
    tol = 1e-8
    assert _relerr(_np(e.region(_abi.R_KGAIN)), ref["K"]) < tol
    assert _relerr(_np(e.region(_abi.R_KFF)), ref["k"]) < tol
    assert _relerr(_np(e.region(_abi.R_QU)), ref["Qu"]) < tol
    assert _relerr(_np(e.region(_abi.R_VX)), ref["Vx"]) < tol
    assert _relerr(_np(e.region(_abi.R_VXX)), ref["Vxx"]) < tol
    assert _relerr(_np(e.traj_f(_abi.TF_D1)), ref["d1"]) < tol
    assert _relerr(_np(e.traj_f(_abi.TF_D2)), ref["d2"]) < tol
    assert _relerr(_np(e.traj_f(_abi.TF_STOP)), ref["stop"]) < tol


@pytest.mark.parametrize("name,kw", CASES)
def test_forward_pass_matches_oracle_for_every_alpha(oracle, name, kw):
    import torch
    sc = scenarios.SCENARIOS[name](**kw)
    low = scenarios.lower(sc)
    sp = scenarios.solver_params(sc)
    e = _engine(low)
    xs, us, deriv, gaps = _backward_inputs(oracle, low, 5)
    ref_b = oracle.backward_pass(low, sp, deriv, gaps, us, 1e-3, 1)
    K, k = 0.05 * ref_b["K"], 0.05 * ref_b["k"]  # mild gains keep every alpha's rollout finite
    e.region(_abi.R_XS).copy_(torch.as_tensor(xs))
    e.region(_abi.R_US).copy_(torch.as_tensor(us))
    e.region(_abi.R_KGAIN).copy_(torch.as_tensor(K))
    e.region(_abi.R_KFF).copy_(torch.as_tensor(k))
    e.region(_abi.R_TRAJ_I)[_abi.TI_FEASIBLE].fill_(1)
    e.forward_pass(sp)
    _sync()
    XT, UT = _np(e.region(_abi.R_XS_TRY)), _np(e.region(_abi.R_US_TRY))
    for a in range(_abi.NALPHA):
        xs_try, us_try, cost_try, fail = oracle.forward_pass(low, sp, 0.5 ** a, xs, us, K, k)
        ok = fail == 0
        assert ok.any()
        assert _relerr(XT[a][:, ok], xs_try[:, ok]) < 1e-9
        assert _relerr(UT[a][:, ok], us_try[:, ok]) < 1e-9
        got = _np(e.traj_f(_abi.TF_COST_TRY0 + a))
        assert _relerr(got[ok], cost_try[ok]) < 1e-9
        assert np.isnan(got[~ok]).all()


SOLVE_CASES = [
    ("two_dof_vsa_boxddp", dict(B=12, T=100), "SolverBoxDDP"),
    ("two_dof_sea", dict(B=12, T=100), "SolverDDP"),
    ("two_dof_sea", dict(B=6, T=100), "SolverFDDP"),
    ("double_pendulum", dict(T=10), "SolverDDP"),
    ("double_pendulum", dict(T=10), "SolverFDDP"),
    ("talos_arm_sea", dict(B=3, T=30), "SolverDDP"),   # C5 model at a size the oracle solves in seconds
    ("talos_arm_sea", dict(B=2, T=30), "SolverFDDP"),
]


def _with_box(sc, lb, ub):
    """the scenario's running model with control limits (-> has_control_limits, SolverBoxDDP's QP at every knot)"""
    sc = dict(sc)
    sc["running"][0].u_lb = np.full(sc["running"][0].nu, float(lb))
    sc["running"][0].u_ub = np.full(sc["running"][0].nu, float(ub))
    return sc


@pytest.mark.gpu
@pytest.mark.parametrize("hs", [0, 2])
def test_boxddp_on_the_seven_joint_arm_matches_oracle(oracle, monkeypatch, hs):
    """SolverBoxDDP at nx = 28 / nu = 7 (north_star: batched DDP / BoxDDP; config 5's model): motor commands boxed tightly
    enough that the QP clamps at many knots.  hs = 0: the block-per-trajectory kernel with the box QP in its gains phase
    (the default since round 3); hs = 2: the register-column kernel it replaces.  Backward pass from identical inputs
    against the oracle, then full solves with the oracle's iteration counts."""
    import torch
    if hs:
        monkeypatch.setenv("ASLR_BWD_HS", str(hs))
    else:
        monkeypatch.delenv("ASLR_BWD_HS", raising=False)
    sc = _with_box(scenarios.talos_arm_sea(B=5, T=12, seed=2), -0.6, 0.6)
    low = scenarios.lower(sc)
    sp = scenarios.solver_params(sc, solver="SolverBoxDDP")
    e = _engine(low)
    xs, us, deriv, gaps = _backward_inputs(oracle, low, 3)
    us = np.clip(us, -0.6, 0.6)
    rng = np.random.default_rng(5)
    k0 = rng.uniform(-0.5, 0.5, us.shape)       # stored k = the QP's warm start
    xreg = 1e-3
    e.region(_abi.R_US).copy_(torch.as_tensor(us))
    e.region(_abi.R_DERIV).copy_(torch.as_tensor(deriv))
    e.region(_abi.R_GAPS).zero_()
    e.region(_abi.R_KFF).copy_(torch.as_tensor(k0))
    e.region(_abi.R_TRAJ_F)[_abi.TF_XREG].fill_(xreg)
    e.region(_abi.R_TRAJ_I)[_abi.TI_FEASIBLE].fill_(1)
    e.region(_abi.R_TRAJ_I)[_abi.TI_STATUS].fill_(0)
    e.backward_pass(sp)
    _sync()
    ref = oracle.backward_pass(low, sp, deriv, np.zeros_like(gaps), us, xreg, 1, kff0=k0)
    assert not ref["fail"].any()
    assert (_np(e.traj_i(_abi.TI_STATUS)) & _abi.ST_BACKWARD_ERR == 0).all()
    clamped = (ref["Qu"] == 0.0).mean()
    assert 0.05 < clamped < 0.95, clamped       # the box is really active at a share of the (knot, control) pairs
    tol = 1e-8
    assert _relerr(_np(e.region(_abi.R_KGAIN)), ref["K"]) < tol
    assert _relerr(_np(e.region(_abi.R_KFF)), ref["k"]) < tol
    assert _relerr(_np(e.region(_abi.R_QU)), ref["Qu"]) < tol
    assert _relerr(_np(e.region(_abi.R_VX)), ref["Vx"]) < tol
    assert _relerr(_np(e.region(_abi.R_VXX)), ref["Vxx"]) < tol
    for fld, name in ((_abi.TF_D1, "d1"), (_abi.TF_D2, "d2"), (_abi.TF_STOP, "stop")):
        assert _relerr(_np(e.traj_f(fld)), ref[name]) < tol
    # full solves
    sc2 = _with_box(scenarios.talos_arm_sea(B=3, T=30, seed=4), -1.0, 1.0)
    low2 = scenarios.lower(sc2)
    sp2 = scenarios.solver_params(sc2, solver="SolverBoxDDP", maxiter=40)
    ref2 = oracle.solve(low2, sp2)
    e2 = _engine(low2)
    e2.set_candidate(None, None)
    e2.solve(sp2, poll_every=4)
    _sync()
    np.testing.assert_array_equal(_np(e2.traj_i(_abi.TI_ITER)), ref2["traj_i"][_abi.TI_ITER])
    import _parity
    _parity.assert_status_words_match(_np(e2.traj_i(_abi.TI_STATUS)), ref2["traj_i"][_abi.TI_STATUS])
    U = _np(e2.region(_abi.R_US))
    assert U.min() >= -1.0 and U.max() <= 1.0 and (np.abs(U) == 1.0).any()    # the solution rides the bounds somewhere
    dx = np.abs(_np(e2.region(_abi.R_XS)) - ref2["xs"]).max()
    du = np.abs(U - ref2["us"]).max()
    dc = np.abs(_np(e2.traj_f(_abi.TF_COST)) - ref2["traj_f"][_abi.TF_COST]).max()
    print("7-joint BoxDDP hs=%d: iterations %s, dx %.2e du %.2e dcost %.2e" % (hs, ref2["traj_i"][_abi.TI_ITER], dx, du, dc))
    assert dx < 1e-6 and du < 1e-6 and dc < 1e-4


@pytest.mark.parametrize("name,kw,solver", SOLVE_CASES)
def test_solve_matches_oracle(oracle, name, kw, solver):
    """north_star: xs/us within 1e-6, final cost within 1e-4 of the CPU solver on identical inputs."""
    sc = scenarios.SCENARIOS[name](**kw)
    low = scenarios.lower(sc)
    sp = scenarios.solver_params(sc, solver=solver)
    ref = oracle.solve(low, sp)
    e = _engine(low)
    e.set_candidate(None, None)
    e.solve(sp, poll_every=4)
    _sync()
    it_g, it_r = _np(e.traj_i(_abi.TI_ITER)), ref["traj_i"][_abi.TI_ITER]
    st_g, st_r = _np(e.traj_i(_abi.TI_STATUS)), ref["traj_i"][_abi.TI_STATUS]
    conv = (st_r & _abi.ST_CONVERGED) != 0
    assert conv.any()
    assert ((st_g & _abi.ST_CONVERGED) != 0)[conv].all()
    X, U = _np(e.region(_abi.R_XS)), _np(e.region(_abi.R_US))
    dx = np.abs(X - ref["xs"])[:, conv].max()
    du = np.abs(U - ref["us"])[:, conv].max()
    dc = np.abs(_np(e.traj_f(_abi.TF_COST)) - ref["traj_f"][_abi.TF_COST])[conv].max()
    print(name, solver, "iters gpu", it_g, "oracle", it_r, "dx %.2e du %.2e dcost %.2e" % (dx, du, dc))
    assert dx < 1e-6 and du < 1e-6, (dx, du)
    assert dc < 1e-4, dc
    assert (it_g == it_r)[conv].all()


def test_vsa_modified_example_first_iterations_match_oracle(oracle):
    """examples/two_dof_vsa_modified.py (linear stiffness cost, sigma >= 0.002).  Its Quu is
    singular along the stiffness directions (no control regulariser there): one ill-conditioned BoxQP step
    amplifies rounding differences by ~1e6 (tools/diverge.py: 7e-12 -> 2e-5 at iteration 9 of trajectory 0 while
    status, iteration counts and regularisation keep matching through iteration 25), and the
    solve runs into its iteration cap.  Parity of the iterates is therefore checked on the first 8 iterations,
    the solver state on the first 25."""
    sc = scenarios.two_dof_vsa_modified(B=6, T=60)
    low = scenarios.lower(sc)
    e = _engine(low)
    for maxiter in (25, 8):
        sp = scenarios.solver_params(sc, maxiter=maxiter)
        ref = oracle.solve(low, sp)
        e.set_candidate(None, None)
        e.solve(sp, poll_every=5)
        _sync()
        np.testing.assert_array_equal(_np(e.traj_i(_abi.TI_ITER)), ref["traj_i"][_abi.TI_ITER])
        np.testing.assert_array_equal(_np(e.traj_i(_abi.TI_STATUS)), ref["traj_i"][_abi.TI_STATUS])
        np.testing.assert_allclose(_np(e.traj_f(_abi.TF_XREG)), ref["traj_f"][_abi.TF_XREG], rtol=0)
        if maxiter == 8:
            np.testing.assert_allclose(_np(e.traj_f(_abi.TF_STEP)), ref["traj_f"][_abi.TF_STEP], rtol=0)
    scale = max(1.0, np.abs(ref["xs"]).max(), np.abs(ref["us"]).max())
    dx = np.abs(_np(e.region(_abi.R_XS)) - ref["xs"]).max()
    du = np.abs(_np(e.region(_abi.R_US)) - ref["us"]).max()
    print("vsa_modified: dx %.2e du %.2e scale %.2e" % (dx, du, scale))
    assert dx < 1e-6 * scale and du < 1e-6 * scale
    assert (_np(e.region(_abi.R_US))[..., 2:] >= 0.002).all()   # the stiffness bound holds


def test_three_action_models_along_the_horizon_match_oracle(oracle):
    """A horizon made of three distinct action models (control limits and dt = 1e-2 on the first 15 knots, no limits,
    dt = 5e-3 and a heavier control regulariser on the next 15, then the terminal model): the per-node model switch
    of every kernel (constants reloaded mid-sweep, limits per node) against the oracle."""
    sc = scenarios.two_dof_vsa_boxddp(B=6, T=30)
    import copy
    rm = sc["running"][0]
    memo = {id(rm.state): rm.state, id(rm.state.pinocchio): rm.state.pinocchio}   # same robot model object
    other = copy.deepcopy(rm, memo)
    other.dt = 5e-3
    other.u_lb = np.full(4, -np.inf)
    other.u_ub = np.full(4, np.inf)
    other.differential.costs.costs["uReg"].weight = 0.5
    sc["running"] = sc["running"][:15] + [other] * 15
    low = scenarios.lower(sc)
    assert low.desc.nmodels == 3 and list(low.node_model) == [0] * 15 + [1] * 15 + [2]
    sp = scenarios.solver_params(sc, maxiter=40)
    ref = oracle.solve(low, sp)
    e = _engine(low)
    e.set_candidate(None, None)
    e.solve(sp, poll_every=4)
    _sync()
    np.testing.assert_array_equal(_np(e.traj_i(_abi.TI_ITER)), ref["traj_i"][_abi.TI_ITER])
    np.testing.assert_array_equal(_np(e.traj_i(_abi.TI_STATUS)), ref["traj_i"][_abi.TI_STATUS])
    scale = max(1.0, np.abs(ref["xs"]).max(), np.abs(ref["us"]).max())
    assert np.abs(_np(e.region(_abi.R_XS)) - ref["xs"]).max() < 1e-6 * scale
    assert np.abs(_np(e.region(_abi.R_US)) - ref["us"]).max() < 1e-6 * scale
    U = _np(e.region(_abi.R_US))
    assert (U[:15, :, 2:] >= 0.0).all()          # the first model's stiffness bound holds on its knots


def test_solve_matches_oracle_on_a_larger_batch(oracle):
    """128 BoxDDP trajectories of another seed, full solves (the oracle runs them on the host cores): every
    trajectory must take the same number of iterations and end with the same status word; the converged ones agree
    within north_star's tolerances."""
    import os
    sc = scenarios.two_dof_vsa_boxddp(B=128, T=100, seed=11)
    low = scenarios.lower(sc)
    sp = scenarios.solver_params(sc)
    ref = oracle.solve(low, sp, nthreads=min(16, len(os.sched_getaffinity(0))))
    e = _engine(low)
    e.set_candidate(None, None)
    e.solve(sp, poll_every=4)
    _sync()
    np.testing.assert_array_equal(_np(e.traj_i(_abi.TI_ITER)), ref["traj_i"][_abi.TI_ITER])
    np.testing.assert_array_equal(_np(e.traj_i(_abi.TI_STATUS)), ref["traj_i"][_abi.TI_STATUS])
    conv = (ref["traj_i"][_abi.TI_STATUS] & _abi.ST_CONVERGED) != 0
    assert conv.sum() > 100
    dx = np.abs(_np(e.region(_abi.R_XS)) - ref["xs"])[:, conv].max()
    du = np.abs(_np(e.region(_abi.R_US)) - ref["us"])[:, conv].max()
    dc = np.abs(_np(e.traj_f(_abi.TF_COST)) - ref["traj_f"][_abi.TF_COST])[conv].max()
    print("128 trajectories: converged %d, dx %.2e du %.2e dcost %.2e" % (conv.sum(), dx, du, dc))
    assert dx < 1e-6 and du < 1e-6 and dc < 1e-4


@pytest.mark.parametrize("name,solver", [("two_dof_vsa_boxddp", "SolverBoxDDP"), ("two_dof_sea", "SolverDDP")])
def test_large_shard_kernel_variants_match_oracle(oracle, name, solver):
    """Large shards: a short-horizon batch of 2051 trajectories (not a multiple of the 4 teams of a wave) over the first
    iterations of a cold start, and two iterations at 8200 trajectories, where the launcher picks the backward sweep
    with one lane set per column (HS = 1), against the oracle."""
    import os
    nth = min(16, len(os.sched_getaffinity(0)))
    sc = scenarios.SCENARIOS[name](B=2051, T=12, seed=5)
    low = scenarios.lower(sc)
    sp = scenarios.solver_params(sc, solver=solver, maxiter=8)
    ref = oracle.solve(low, sp, nthreads=nth)
    e = _engine(low)
    e.set_candidate(None, None)
    e.solve(sp, poll_every=0)
    _sync()
    # status words: decision / outcome bits exactly; the "a rejected trial overflowed" note may differ on a trajectory or
    # two (an unstable rollout amplifies the 1e-13 difference of the gains as much as the state: _parity.py, and
    # profiles/r02/forward_err_probe_traj1005.txt).  The rollouts test |xnext|_inf like Crocoddyl's raiseIfNaN.
    import _parity
    np.testing.assert_array_equal(_np(e.traj_i(_abi.TI_ITER)), ref["traj_i"][_abi.TI_ITER])
    _parity.assert_status_words_match(_np(e.traj_i(_abi.TI_STATUS)), ref["traj_i"][_abi.TI_STATUS])
    scale = np.maximum(1.0, np.abs(ref["xs"]).max(axis=(0, 2)))
    assert (np.abs(_np(e.region(_abi.R_XS)) - ref["xs"]).max(axis=(0, 2)) < 1e-6 * scale).all()
    uscale = np.maximum(1.0, np.abs(ref["us"]).max(axis=(0, 2)))
    assert (np.abs(_np(e.region(_abi.R_US)) - ref["us"]).max(axis=(0, 2)) < 1e-6 * uscale).all()
    # one iteration at a batch that selects the HS = 1 backward sweep
    sc = scenarios.SCENARIOS[name](B=8200, T=6, seed=6)
    low = scenarios.lower(sc)
    sp = scenarios.solver_params(sc, solver=solver, maxiter=2)
    ref = oracle.solve(low, sp, nthreads=nth)
    e = _engine(low)
    e.set_candidate(None, None)
    e.solve(sp, poll_every=0)
    _sync()
    _parity.assert_status_words_match(_np(e.traj_i(_abi.TI_STATUS)), ref["traj_i"][_abi.TI_STATUS])
    scale = np.maximum(1.0, np.abs(ref["xs"]).max(axis=(0, 2)))
    assert (np.abs(_np(e.region(_abi.R_XS)) - ref["xs"]).max(axis=(0, 2)) < 1e-6 * scale).all()


@pytest.mark.parametrize("name,kw,solver,maxiter", [
    ("double_pendulum", dict(T=100), "SolverDDP", 15),        # C1 as BASELINE.json states it (T = 100, SolverDDP)
    ("double_pendulum", dict(T=100), "SolverFDDP", 15),
    ("two_dof_vsa_boxddp", dict(B=8, T=100), "SolverDDP", 15),   # the VSA models under the unconstrained solvers
    ("two_dof_vsa_boxddp", dict(B=8, T=100), "SolverFDDP", 15),
    # cold-started plain DDP on the 7-DoF arm: its first full step throws the iterates to ~6e3 and one of the four
    # trajectories then creeps back with step lengths of 1/16: rounding differences (3e-10 relative after the
    # first iteration) grow ~2x per iteration on it (tools/diverge.py); 6 iterations are compared
    ("talos_arm_sea", dict(B=4, T=60), "SolverDDP", 6),
])
def test_first_iterations_match_oracle(oracle, name, kw, solver, maxiter):
    """The first iterations from the cold start (these runs do not all converge within their cap): solver state
    exactly, iterates relative to their size."""
    sc = scenarios.SCENARIOS[name](**kw)
    low = scenarios.lower(sc)
    sp = scenarios.solver_params(sc, solver=solver, maxiter=maxiter)
    ref = oracle.solve(low, sp)
    e = _engine(low)
    e.set_candidate(None, None)
    e.solve(sp, poll_every=4)
    _sync()
    np.testing.assert_array_equal(_np(e.traj_i(_abi.TI_ITER)), ref["traj_i"][_abi.TI_ITER])
    np.testing.assert_array_equal(_np(e.traj_i(_abi.TI_STATUS)), ref["traj_i"][_abi.TI_STATUS])
    np.testing.assert_allclose(_np(e.traj_f(_abi.TF_XREG)), ref["traj_f"][_abi.TF_XREG], rtol=0)
    np.testing.assert_allclose(_np(e.traj_f(_abi.TF_STEP)), ref["traj_f"][_abi.TF_STEP], rtol=0)
    scale = max(1.0, np.abs(ref["xs"]).max(), np.abs(ref["us"]).max())
    dx = np.abs(_np(e.region(_abi.R_XS)) - ref["xs"]).max()
    du = np.abs(_np(e.region(_abi.R_US)) - ref["us"]).max()
    dc = np.abs(_np(e.traj_f(_abi.TF_COST)) - ref["traj_f"][_abi.TF_COST]).max()
    print(name, solver, "dx %.2e du %.2e dcost %.2e scale %.2e" % (dx, du, dc, scale))
    assert dx < 1e-6 * scale and du < 1e-6 * scale


def _indefinite_sea(B, T, cost_name, weight):
    """SEA problem with one NEGATIVE cost weight: Quu / Vxx turn indefinite, so backward passes fail
    (Cholesky "backward_error" -> increaseRegularization -> retry without recalc) and the
    solve ends at reg_max or keeps iterating at a raised regularisation."""
    sc = scenarios.two_dof_sea(B=B, T=T)
    sc["running"][0].differential.costs.costs[cost_name].weight = weight
    return sc


@pytest.mark.parametrize("cost_name,weight,maxiter,hits_reg_max",
                         [("uReg", -5e-3, 12, False), ("xReg", -1e-2, 12, True), ("uReg", -1e12, 30, True)])
def test_backward_error_recovery_and_reg_max_match_oracle(oracle, cost_name, weight, maxiter, hits_reg_max):
    # (a cost that is unbounded below makes the iterates run away exponentially: rounding differences of 1e-13
    #  reach O(1) after ~15 iterations on either side, so the recovery cases stop at 12 iterations)
    sc = _indefinite_sea(6, 20, cost_name, weight)
    low = scenarios.lower(sc)
    sp = scenarios.solver_params(sc, maxiter=maxiter)
    ref = oracle.solve(low, sp)
    e = _engine(low)
    e.set_candidate(None, None)
    e.solve(sp, poll_every=3)
    _sync()
    st_g, st_r = _np(e.traj_i(_abi.TI_STATUS)), ref["traj_i"][_abi.TI_STATUS]
    assert ((st_r & _abi.ST_BACKWARD_ERR) != 0).all()          # the path under test was taken
    assert ((st_r & _abi.ST_REG_MAX) != 0).any() == hits_reg_max
    np.testing.assert_array_equal(st_g, st_r)
    np.testing.assert_array_equal(_np(e.traj_i(_abi.TI_ITER)), ref["traj_i"][_abi.TI_ITER])
    np.testing.assert_allclose(_np(e.traj_f(_abi.TF_XREG)), ref["traj_f"][_abi.TF_XREG], rtol=0)
    # run-away iterates (the negative weight makes the cost unbounded below) amplify rounding differences
    # without bound: values are compared per trajectory relative to its size, on those that stayed below 1e6;
    # status, iteration count and regularisation above are compared on all of them
    mag = np.maximum(np.abs(ref["xs"]).max(axis=(0, 2)), np.abs(ref["us"]).max(axis=(0, 2)))
    tame = mag < 1e6
    assert tame.any()
    scale = np.maximum(1.0, mag)[tame]
    tol = 1e-6 if not hits_reg_max else 1e-4   # (the reg_max cases amplify by ~1e3 per iteration near the end)
    assert (np.abs(_np(e.region(_abi.R_XS)) - ref["xs"])[:, tame].max(axis=(0, 2)) < tol * scale).all()
    assert (np.abs(_np(e.region(_abi.R_US)) - ref["us"])[:, tame].max(axis=(0, 2)) < tol * scale).all()


def test_forward_error_is_skipped_like_crocoddyl(oracle):
    """A rollout that overflows (NaN / Inf / >= 1e30) makes that step length a "forward_error": it is skipped and
    the next alpha is tried.  Huge feed-forward terms provoke it."""
    import torch
    sc = scenarios.two_dof_sea(B=4, T=40)
    low = scenarios.lower(sc)
    sp = scenarios.solver_params(sc)
    e = _engine(low)
    xs = np.zeros((low.T + 1, low.B, low.nx))
    us = np.zeros((low.T, low.B, low.nu))
    K = np.zeros((low.T, low.B, low.nu, low.nx))
    k = np.full((low.T, low.B, low.nu), -1e200)  # u = us - alpha k: overflows to inf in a few steps for large alpha
    e.region(_abi.R_XS).copy_(torch.as_tensor(xs)); e.region(_abi.R_US).copy_(torch.as_tensor(us))
    e.region(_abi.R_KGAIN).copy_(torch.as_tensor(K)); e.region(_abi.R_KFF).copy_(torch.as_tensor(k))
    e.region(_abi.R_TRAJ_I)[_abi.TI_FEASIBLE].fill_(1)
    e.forward_pass(sp)
    _sync()
    for a in range(_abi.NALPHA):
        _, _, cost_try, fail = oracle.forward_pass(low, sp, 0.5 ** a, xs, us, K, k)
        got = _np(e.traj_f(_abi.TF_COST_TRY0 + a))
        np.testing.assert_array_equal(np.isnan(got), fail != 0)
        assert fail.all()


def test_edge_sizes_single_knot_single_trajectory(oracle):
    for B, T in ((1, 1), (3, 2), (65, 1)):
        sc = scenarios.two_dof_vsa_boxddp(B=B, T=T)
        low = scenarios.lower(sc)
        sp = scenarios.solver_params(sc, maxiter=15)
        ref = oracle.solve(low, sp)
        e = _engine(low)
        e.set_candidate(None, None)
        e.solve(sp, poll_every=1)
        _sync()
        # 15 iterations of a cold start: mid-descent iterates of size ~1e2, compared relative to their size
        scale = max(1.0, np.abs(ref["xs"]).max(), np.abs(ref["us"]).max())
        assert np.abs(_np(e.region(_abi.R_XS)) - ref["xs"]).max() < 1e-6 * scale
        assert np.abs(_np(e.region(_abi.R_US)) - ref["us"]).max() < 1e-6 * scale
        np.testing.assert_array_equal(_np(e.traj_i(_abi.TI_ITER)), ref["traj_i"][_abi.TI_ITER])
        np.testing.assert_array_equal(_np(e.traj_i(_abi.TI_STATUS)), ref["traj_i"][_abi.TI_STATUS])
    # maxiter = 0 leaves the candidate untouched
    sc = scenarios.two_dof_sea(B=2, T=5)
    e = _engine(scenarios.lower(sc))
    xs0 = np.random.default_rng(0).normal(size=(2, 6, 8))
    e.set_candidate(xs0, None)
    e.solve(scenarios.solver_params(sc, maxiter=0))
    _sync()
    np.testing.assert_array_equal(_np(e.xs), xs0)


def test_c5_horizon_full_solves_match_oracle(oracle):
    """The C5 problem at its full horizon (7-DoF SEA, nx = 28, T = 150): 16 trajectories solved to convergence by
    SolverFDDP (the solver the SEA example uses, examples/two_dof_sea.py:69) through the block / team kernels, against
    the oracle: iteration counts, decision bits of the status words, and xs / us / cost of the converged ones within
    the north_star tolerances (relative to the size of the iterates: a cold-started arm swings through ~1e2 rad/s)."""
    import os
    import _parity
    nth = min(16, len(os.sched_getaffinity(0)))
    sc = scenarios.talos_arm_sea(B=16, T=150, seed=0)
    low = scenarios.lower(sc)
    sp = scenarios.solver_params(sc, solver="SolverFDDP")
    ref = oracle.solve(low, sp, nthreads=nth)
    e = _engine(low)
    e.set_candidate(None, None)
    e.solve(sp, poll_every=4)
    _sync()
    it_g, it_r = _np(e.traj_i(_abi.TI_ITER)), ref["traj_i"][_abi.TI_ITER]
    st_r = ref["traj_i"][_abi.TI_STATUS]
    conv = (st_r & _abi.ST_CONVERGED) != 0
    assert conv.sum() >= 12
    # trajectories that converge take the same number of iterations; the one that does not (and one that stops at the
    # regularisation ceiling after a single iteration) must end with the same outcome bits
    np.testing.assert_array_equal(it_g[conv], it_r[conv])
    _parity.assert_status_words_match(_np(e.traj_i(_abi.TI_STATUS)), st_r, max_note_flips=2)
    X, U = _np(e.region(_abi.R_XS)), _np(e.region(_abi.R_US))
    scale = np.maximum(1.0, np.maximum(np.abs(ref["xs"]).max(axis=(0, 2)), np.abs(ref["us"]).max(axis=(0, 2))))
    dx = np.abs(X - ref["xs"]).max(axis=(0, 2)) / scale
    du = np.abs(U - ref["us"]).max(axis=(0, 2)) / scale
    dc = np.abs(_np(e.traj_f(_abi.TF_COST)) - ref["traj_f"][_abi.TF_COST])
    print("C5 horizon: converged %d / 16, max rel |dx| %.2e |du| %.2e, |dcost| %.2e" % (conv.sum(), dx[conv].max(), du[conv].max(), dc[conv].max()))
    assert dx[conv].max() < 1e-6 and du[conv].max() < 1e-6
    assert (dc[conv] < 1e-4 * np.maximum(1.0, np.abs(ref["traj_f"][_abi.TF_COST][conv]))).all()


@pytest.mark.gpu
def test_opt_in_closed_form_reach_residual_matches_the_general_log_map_on_the_gpu(oracle, monkeypatch):
    """ASLR_PLANAR_REACH=1 (closed-form frame-placement residual of planar chains in cost-only evaluations, off by
    default): the trial costs of every step length agree with the general SE(3) log path to 5e-9 relative (1e-12 typically;
    Pinocchio's t sin t / (2 (1 - cos t)) in the general path loses digits at small rotation angles) -- on the GPU,
    not only in the numpy derivation of tests/test_planar_reach_closed_form.py -- and a pool with its own targets falls back
    to the general path (aslr_solve_pool) instead of mixing the two formulas."""
    import torch
    sc = scenarios.two_dof_vsa_boxddp(B=70, T=5)
    low = scenarios.lower(sc)
    sp = scenarios.solver_params(sc)
    xs, us, deriv, gaps = _backward_inputs(oracle, low, 5)
    ref_b = oracle.backward_pass(low, sp, deriv, gaps, us, 1e-3, 1)
    K, k = 0.05 * ref_b["K"], 0.05 * ref_b["k"]
    costs = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("ASLR_PLANAR_REACH", flag)   # (read when the problem handle is created)
        e = _engine(low)
        e.region(_abi.R_XS).copy_(torch.as_tensor(xs))
        e.region(_abi.R_US).copy_(torch.as_tensor(us))
        e.region(_abi.R_KGAIN).copy_(torch.as_tensor(K))
        e.region(_abi.R_KFF).copy_(torch.as_tensor(k))
        e.region(_abi.R_TRAJ_I)[_abi.TI_FEASIBLE].fill_(1)
        e.forward_pass(sp)
        _sync()
        costs[flag] = np.stack([_np(e.traj_f(_abi.TF_COST_TRY0 + a)) for a in range(_abi.NALPHA)])
        if flag == "1":
            r1 = e.solve_pool(sc["x0"][:20], sc["frame_refs"][:20], scenarios.solver_params(sc, maxiter=6))
    ok = np.isfinite(costs["0"])
    assert ok.mean() > 0.9 and (np.isfinite(costs["1"]) == ok).all()
    assert _relerr(costs["1"][ok], costs["0"][ok]) < 5e-9
    assert (costs["1"][ok] != costs["0"][ok]).any()      # the closed form really ran (it rounds differently)
    monkeypatch.setenv("ASLR_PLANAR_REACH", "0")
    r0 = _engine(low).solve_pool(sc["x0"][:20], sc["frame_refs"][:20], scenarios.solver_params(sc, maxiter=6))
    assert torch.equal(r1["xs"], r0["xs"]) and torch.equal(r1["cost"], r0["cost"])


@pytest.mark.gpu
def test_c2_as_baseline_states_it_1024_sea_ddp_full_solves(oracle):
    """BASELINE.json configs[1] at its full size: 2-DoF SEA, SolverDDP, 1024 trajectories x T = 100, full solves
    (th_stop 1e-7, maxiter 100; the oracle needs ~1 s on the host cores).  Every trajectory must take the oracle's number
    of iterations and end with its status word; iterates within 1e-6, costs within 1e-4 (measured: 6e-13)."""
    import os
    sc = scenarios.two_dof_sea(B=1024, T=100, seed=0)
    low = scenarios.lower(sc)
    sp = scenarios.solver_params(sc, solver="SolverDDP")
    ref = oracle.solve(low, sp, nthreads=min(16, len(os.sched_getaffinity(0))))
    e = _engine(low)
    e.set_candidate(None, None)
    e.solve(sp, poll_every=4)
    _sync()
    np.testing.assert_array_equal(_np(e.traj_i(_abi.TI_ITER)), ref["traj_i"][_abi.TI_ITER])
    np.testing.assert_array_equal(_np(e.traj_i(_abi.TI_STATUS)), ref["traj_i"][_abi.TI_STATUS])
    conv = (ref["traj_i"][_abi.TI_STATUS] & _abi.ST_CONVERGED) != 0
    assert conv.sum() > 1000
    dx = np.abs(_np(e.region(_abi.R_XS)) - ref["xs"]).max()
    du = np.abs(_np(e.region(_abi.R_US)) - ref["us"]).max()
    dc = np.abs(_np(e.traj_f(_abi.TF_COST)) - ref["traj_f"][_abi.TF_COST]).max()
    print("C2 1024 x 100: converged %d, iterations %d..%d, dx %.2e du %.2e dcost %.2e"
          % (conv.sum(), ref["traj_i"][_abi.TI_ITER].min(), ref["traj_i"][_abi.TI_ITER].max(), dx, du, dc))
    assert dx < 1e-6 and du < 1e-6 and dc < 1e-4


@pytest.mark.gpu
def test_vsa_modified_examples_own_problem_full_solve(oracle):
    """examples/two_dof_vsa_modified.py:59-81 as the script runs it: ITS problem (trajectory 0 of the scenario: x0 = 0, the
    script's target), T = 200, solve([], [], 400).  The stiffness block of Quu has no curvature of its own there (linear
    cost, u_reg 1e-9), which makes RANDOM variations of the problem part ways with any differently rounded solver
    (profiles/r02/parity_vsa_modified_64.txt) -- the example itself ends within north_star's tolerances of the oracle
    with the oracle's iteration count, which is what this test pins."""
    sc = scenarios.two_dof_vsa_modified(B=1, T=200, seed=0)
    low = scenarios.lower(sc)
    sp = scenarios.solver_params(sc)
    assert sp.maxiter == 400
    ref = oracle.solve(low, sp)
    e = _engine(low)
    e.set_candidate(None, None)
    e.solve(sp, poll_every=8)
    _sync()
    it_g, it_r = int(_np(e.traj_i(_abi.TI_ITER))[0]), int(ref["traj_i"][_abi.TI_ITER][0])
    st_g, st_r = int(_np(e.traj_i(_abi.TI_STATUS))[0]), int(ref["traj_i"][_abi.TI_STATUS][0])
    dx = np.abs(_np(e.region(_abi.R_XS)) - ref["xs"]).max()
    du = np.abs(_np(e.region(_abi.R_US)) - ref["us"]).max()
    dc = abs(float(_np(e.traj_f(_abi.TF_COST))[0]) - float(ref["traj_f"][_abi.TF_COST][0]))
    print("two_dof_vsa_modified T=200: iterations gpu %d oracle %d, status %d / %d, dx %.2e du %.2e dcost %.2e"
          % (it_g, it_r, st_g, st_r, dx, du, dc))
    assert it_g == it_r and (st_g & ~_abi.ST_FORWARD_ERR) == (st_r & ~_abi.ST_FORWARD_ERR)
    assert dx < 1e-6 and du < 1e-6 and dc < 1e-4
