"""Pin the CPU oracle's solver pieces with independent known answers:
time-varying LQR (numpy Riccati recursion), brute-force enumeration of active sets for the box QP,
and solver-level invariants.  Solver parity with Crocoddyl itself is "parity unpinned" (no golden
vectors exist in the reference and Crocoddyl cannot be installed here).
"""
import itertools

import numpy as np
import pytest

from aslr_to_amd import _abi, scenarios


def _random_lq_records(low, rng):
    """Random LQ blocks in DERIV layout: generic Fx, Fu, SPD Lxx/Luu, dense Lxu."""
    T, B, nx, nu = low.T, low.B, low.nx, low.nu
    o = _abi.record_offsets(nx, nu)
    D = np.zeros((T + 1, B, low.rec))
    blocks = {}
    for t in range(T + 1):
        for b in range(B):
            Fx = np.eye(nx) + 0.1 * rng.normal(size=(nx, nx))
            Fu = rng.normal(size=(nx, nu))
            H = rng.normal(size=(nx + nu, nx + nu))
            H = H.dot(H.T) + 0.5 * np.eye(nx + nu)
            Lxx, Lxu, Luu = H[:nx, :nx], H[:nx, nx:], H[nx:, nx:]
            Lx, Lu = rng.normal(size=nx), rng.normal(size=nu)
            for k, v in (("Fx", Fx), ("Fu", Fu), ("Lxx", Lxx), ("Lxu", Lxu), ("Luu", Luu), ("Lx", Lx), ("Lu", Lu)):
                D[t, b, o[k]:o[k] + v.size] = v.ravel()
            blocks[t, b] = (Fx, Fu, Lxx, Lxu, Luu, Lx, Lu)
    return D, blocks


@pytest.mark.parametrize("feasible", [1, 0])
def test_backward_pass_equals_numpy_riccati_on_random_lq_problems(oracle, feasible):
    sc = scenarios.two_dof_sea(B=3, T=7)
    low = scenarios.lower(sc)
    sp = scenarios.solver_params(sc)
    rng = np.random.default_rng(0)
    D, blocks = _random_lq_records(low, rng)
    T, B, nx, nu = low.T, low.B, low.nx, low.nu
    gaps = rng.normal(size=(T + 1, B, nx)) * 0.1
    xreg = 1e-4
    out = oracle.backward_pass(low, sp, D, gaps, np.zeros((T, B, nu)), xreg, feasible)
    assert not out["fail"].any()
    for b in range(B):
        _, _, Lxx, _, _, Lx, _ = blocks[T, b]
        Vxx = Lxx + xreg * np.eye(nx)
        Vx = Lx + (0 if feasible else Vxx.dot(gaps[T, b]))
        d1 = d2 = stop = 0.0
        for t in range(T - 1, -1, -1):
            Fx, Fu, Lxx, Lxu, Luu, Lx, Lu = blocks[t, b]
            Qxx, Qxu = Lxx + Fx.T.dot(Vxx).dot(Fx), Lxu + Fx.T.dot(Vxx).dot(Fu)
            Quu = Luu + Fu.T.dot(Vxx).dot(Fu) + xreg * np.eye(nu)
            Qx, Qu = Lx + Fx.T.dot(Vx), Lu + Fu.T.dot(Vx)
            K, k = np.linalg.solve(Quu, Qxu.T), np.linalg.solve(Quu, Qu)
            np.testing.assert_allclose(out["K"][t, b], K, rtol=1e-9, atol=1e-10)
            np.testing.assert_allclose(out["k"][t, b], k, rtol=1e-9, atol=1e-10)
            d1 += Qu.dot(k); d2 -= k.dot(Quu).dot(k); stop += Qu.dot(Qu)
            Vx = Qx - K.T.dot(Qu)
            Vxx = Qxx - Qxu.dot(K)
            Vxx = 0.5 * (Vxx + Vxx.T) + xreg * np.eye(nx)
            if not feasible:
                Vx = Vx + Vxx.dot(gaps[t, b])
            np.testing.assert_allclose(out["Vxx"][t, b], Vxx, rtol=1e-9, atol=1e-9)
            np.testing.assert_allclose(out["Vx"][t, b], Vx, rtol=1e-9, atol=1e-9)
        assert out["d1"][b] == pytest.approx(d1, rel=1e-10)
        assert out["d2"][b] == pytest.approx(d2, rel=1e-10)
        assert out["stop"][b] == pytest.approx(stop, rel=1e-10)


def _brute_force_qp(H, q, lb, ub):
    """Exact box-QP solution by enumerating every (lower / free / upper) assignment."""
    n = len(q)
    best, bx = np.inf, None
    for assign in itertools.product((0, 1, 2), repeat=n):
        x = np.where(np.array(assign) == 0, lb, ub).astype(float)
        free = [i for i in range(n) if assign[i] == 1]
        clamped = [i for i in range(n) if assign[i] != 1]
        if free:
            rhs = -q[free] - H[np.ix_(free, clamped)].dot(x[clamped]) if clamped else -q[free]
            x[free] = np.linalg.solve(H[np.ix_(free, free)], rhs)
        if np.any(x < lb - 1e-12) or np.any(x > ub + 1e-12):
            continue
        f = 0.5 * x.dot(H).dot(x) + q.dot(x)
        if f < best - 1e-14:
            best, bx = f, x
    return bx, best


@pytest.mark.parametrize("n", [2, 4])
def test_boxqp_matches_brute_force_enumeration(oracle, n):
    rng = np.random.default_rng(1)
    for trial in range(40):
        A = rng.normal(size=(n, n))
        H = A.dot(A.T) + 0.2 * np.eye(n)
        q = rng.normal(size=n) * 3
        lb = -rng.uniform(0.05, 1.0, n)
        ub = rng.uniform(0.05, 1.0, n)
        x_ref, f_ref = _brute_force_qp(H, q, lb, ub)
        r = oracle.boxqp(H, q, lb, ub, rng.uniform(-2, 2, n), th_grad=1e-9, reg=0.0)
        f = 0.5 * r["x"].dot(H).dot(r["x"]) + q.dot(r["x"])
        assert f <= f_ref + 1e-9 * (1 + abs(f_ref))
        np.testing.assert_allclose(r["x"], x_ref, atol=1e-7)
        fr = list(r["free"])
        if fr:
            np.testing.assert_allclose(r["Hff_inv"], np.linalg.inv(H[np.ix_(fr, fr)]), rtol=1e-8, atol=1e-10)
        assert sorted(fr + list(r["clamped"])) == list(range(n))


@pytest.mark.parametrize("name,kw,solver", [("two_dof_sea", dict(B=3, T=40), "SolverDDP"),
                                            ("two_dof_sea", dict(B=3, T=40), "SolverFDDP"),
                                            ("two_dof_vsa_boxddp", dict(B=3, T=40), "SolverBoxDDP")])
def test_solver_converges_to_a_stationary_feasible_trajectory(oracle, name, kw, solver):
    sc = scenarios.SCENARIOS[name](**kw)
    low = scenarios.lower(sc)
    sp = scenarios.solver_params(sc, solver=solver)
    r = oracle.solve(low, sp)
    st = r["traj_i"][_abi.TI_STATUS]
    assert ((st & _abi.ST_CONVERGED) != 0).all()
    assert (r["traj_f"][_abi.TF_STOP] < sp.th_stop).all()
    xs, us = r["xs"], r["us"]
    # dynamically feasible: xs[0] = x0 and xs[t+1] = xnext(xs[t], us[t])
    np.testing.assert_allclose(xs[0], sc["x0"], atol=0)
    xnext, cost, _ = oracle.calc_diff(low, xs, us, diff=False)
    np.testing.assert_allclose(xs[1:], xnext[:-1], atol=1e-12)
    np.testing.assert_allclose(cost.sum(axis=0), r["traj_f"][_abi.TF_COST], rtol=1e-12)
    if solver == "SolverBoxDDP":
        m = sc["running"][0]
        assert (us >= m.u_lb - 1e-12).all() and (us <= m.u_ub + 1e-12).all()
    # a cold-started solve improves on the do-nothing trajectory
    xs0, us0 = np.zeros_like(xs), np.zeros_like(us)
    _, c0, _ = oracle.calc_diff(low, np.broadcast_to(sc["x0"], xs.shape).copy(), us0, diff=False)
    assert (r["traj_f"][_abi.TF_COST] < c0.sum(axis=0)).all()


def test_ddp_solves_an_lq_like_problem_in_few_iterations_and_openmp_matches_serial(oracle):
    sc = scenarios.two_dof_sea(B=6, T=30)
    low = scenarios.lower(sc)
    sp = scenarios.solver_params(sc)
    a = oracle.solve(low, sp, nthreads=1)
    b = oracle.solve(low, sp, nthreads=4)
    np.testing.assert_array_equal(a["xs"], b["xs"])
    np.testing.assert_array_equal(a["traj_i"], b["traj_i"])
