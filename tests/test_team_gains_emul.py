"""The lane choreography of the team-distributed gains / box QP (aslr_to_amd/csrc/aslr_team_gains.hpp), emulated on the
host: the SAME template source the backward kernel instantiates with DPP row broadcasts runs on 64 emulated lanes
(tests/host/team_gains_emul.cpp) and is compared with the oracle's BoxQP (oracle/aslr_oracle.c:865-957) and with plain
Cholesky gains, four problems per emulated wavefront in lock-step.  CPU only; the GPU parity tests then check the
device instantiation (tests/test_gpu_parity.py::test_backward_pass_matches_oracle)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def emul(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("emul") / "libteam_gains_emul.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-Wall", "-Wno-unknown-pragmas", "-shared",
                           "-fPIC", "-o", so, os.path.join(HERE, "host", "team_gains_emul.cpp")])
    lib = C.CDLL(so)
    d, i = C.POINTER(C.c_double), C.POINTER(C.c_int)
    lib.emul_team_gains.argtypes = [C.c_int, d, d, d, d, d, i, d, C.c_int, C.c_double, C.c_double, C.c_double, d, d, d, i]
    lib.emul_team_gains.restype = None

    def run(box, H, q, lb, ub, k0, boxed, Qux, maxiter=100, th_acc=0.1, th_grad=1e-5, reg=0.0):
        H, q, lb, ub, k0, Qux = (np.ascontiguousarray(a, dtype=np.float64) for a in (H, q, lb, ub, k0, Qux))
        boxed = np.ascontiguousarray(boxed, dtype=np.int32)
        k, qz, K, bad = np.zeros((4, 4)), np.zeros((4, 4)), np.zeros((4, 4, 8)), np.zeros(4, dtype=np.int32)
        p = lambda a: a.ctypes.data_as(d)
        lib.emul_team_gains(box, p(H), p(q), p(lb), p(ub), p(k0), boxed.ctypes.data_as(i), p(Qux), maxiter, th_acc,
                            th_grad, reg, p(k), p(qz), p(K), bad.ctypes.data_as(i))
        return k, qz, K, bad
    return run


def _problem(rng, scale):
    A = rng.normal(size=(4, 4))
    H = A.dot(A.T) + 0.3 * np.eye(4)
    q = scale * rng.normal(size=4)
    lb = -np.abs(rng.normal(size=4)) * rng.choice([0.05, 0.5, 5.0])
    ub = np.abs(rng.normal(size=4)) * rng.choice([0.05, 0.5, 5.0])
    k0 = rng.normal(size=4) * rng.choice([0.0, 0.3, 3.0])
    return H, q, lb, ub, k0, rng.normal(size=(4, 8))


def _expected(oracle, H, q, lb, ub, k0, Qux, boxed, reg):
    if not boxed:
        return -np.linalg.solve(H, -q), q.copy(), np.linalg.solve(H, Qux), 0
    r = oracle.boxqp(H, q, lb, ub, k0, maxiter=100, th_acceptstep=0.1, th_grad=1e-5, reg=reg)
    Qinv = np.zeros((4, 4))
    f = r["free"]
    if len(f):
        Qinv[np.ix_(f, f)] = r["Hff_inv"]
    qz = q.copy()
    qz[r["clamped"]] = 0.0
    return -r["x"], qz, Qinv.dot(Qux), r["iters"]


def test_emulated_lanes_reproduce_the_oracle_boxqp_and_plain_gains(oracle, emul):
    rng = np.random.default_rng(7)
    stats = dict(iters=np.zeros(8, dtype=int), clamped_start=0, plain=0, n=0)
    for trial in range(600):
        probs = [_problem(rng, rng.choice([0.01, 1.0, 10.0])) for _ in range(4)]
        boxed = rng.integers(0, 4, size=4) > 0  # some rows of the wave are not boxed (infeasible trajectory / no limits)
        reg = 0.0 if trial % 5 else 1e-9
        args = [np.stack([p[i] for p in probs]) for i in range(6)]
        k, qz, K, bad = emul(1, args[0], args[1], args[2], args[3], args[4], boxed, args[5], reg=reg)
        assert not bad.any()
        for row, (H, q, lb, ub, k0, Qux) in enumerate(probs):
            ek, eqz, eK, it = _expected(oracle, H, q, lb, ub, k0, Qux, boxed[row], reg)
            tol = 1e-9 * (1.0 + np.linalg.cond(H) * 1e-2)
            np.testing.assert_allclose(k[row], ek, rtol=0, atol=tol * (1 + np.abs(ek).max()))
            np.testing.assert_allclose(K[row], eK, rtol=0, atol=tol * (1 + np.abs(eK).max()))
            np.testing.assert_array_equal(qz[row] == 0.0, eqz == 0.0)
            np.testing.assert_allclose(qz[row], eqz, rtol=0, atol=0)
            stats["n"] += 1
            if boxed[row]:
                stats["iters"][min(it, 7)] += 1
                x0 = np.clip(k0, lb, ub)
                g0 = q + H.dot(x0)
                stats["clamped_start"] += bool((((x0 == lb) & (g0 > 0)) | ((x0 == ub) & (g0 < 0))).any())
            else:
                stats["plain"] += 1
    # the cases must cover every outcome: returns at once, one / two / three and more projected-Newton iterations
    assert stats["iters"][0] > 5 and stats["iters"][1] > 100 and stats["iters"][2] > 100 and stats["iters"][3:].sum() > 20
    assert stats["clamped_start"] > 300 and stats["plain"] > 300


def test_emulated_lanes_plain_ddp_instantiation(emul):
    rng = np.random.default_rng(11)
    for _ in range(50):
        probs = [_problem(rng, 1.0) for _ in range(4)]
        args = [np.stack([p[i] for p in probs]) for i in range(6)]
        k, qz, K, bad = emul(0, args[0], args[1], args[2], args[3], args[4], np.zeros(4), args[5])
        assert not bad.any()
        for row, (H, q, lb, ub, k0, Qux) in enumerate(probs):
            np.testing.assert_allclose(k[row], np.linalg.solve(H, q), rtol=1e-9, atol=1e-12)
            np.testing.assert_allclose(K[row], np.linalg.solve(H, Qux), rtol=1e-9, atol=1e-11)
            np.testing.assert_array_equal(qz[row], q)


def test_emulated_lanes_flag_an_indefinite_quu(emul):
    rng = np.random.default_rng(3)
    probs = [_problem(rng, 1.0) for _ in range(4)]
    args = [np.stack([p[i] for p in probs]) for i in range(6)]
    args[0][2] = -args[0][2]  # row 2: negative definite
    for box in (0, 1):
        k, qz, K, bad = emul(box, args[0], args[1], args[2], args[3], args[4], np.ones(4), args[5])
        assert bad[2] & 1 and not (bad[[0, 1, 3]] & 1).any()
        # the other rows of the wave are not disturbed by the failing one
        for row in (0, 1, 3):
            assert np.isfinite(k[row]).all() and np.isfinite(K[row]).all()


def test_emulated_lanes_on_qp_instances_of_a_real_solve(oracle, emul):
    """Quu / Qu / bounds / warm starts logged by the oracle during BoxDDP iterations of the headline scenario."""
    from aslr_to_amd import scenarios
    sc = scenarios.two_dof_vsa_boxddp(B=6, T=100)
    low = scenarios.lower(sc)
    sp = scenarios.solver_params(sc)
    L = oracle.lib()
    cap = 4_000_000
    buf = np.zeros(cap)
    L.aslr_cpu_boxqp_dump(buf.ctypes.data_as(C.POINTER(C.c_double)), C.c_long(cap))
    try:
        sp.maxiter = 30
        oracle.solve(low, sp)
        n = L.aslr_cpu_boxqp_dump_len()
    finally:
        L.aslr_cpu_boxqp_dump(None, C.c_long(0))
    rec = 16 + 5 * 4 + 1
    inst = buf[:n].reshape(-1, rec)
    assert len(inst) > 5000
    inst = inst[: len(inst) // 4 * 4]
    multi = 0
    for g in range(0, min(len(inst), 6000), 4):
        blk = inst[g:g + 4]
        H, q, lb, ub, x0, xs = blk[:, :16], blk[:, 16:20], blk[:, 20:24], blk[:, 24:28], blk[:, 28:32], blk[:, 32:36]
        k, qz, K, bad = emul(1, H, q, lb, ub, x0, np.ones(4), np.zeros((4, 4, 8)))
        assert not bad.any()
        np.testing.assert_allclose(k, -xs, rtol=0, atol=1e-9 * (1.0 + np.abs(xs).max()))
        multi += int((blk[:, 36] >= 2).sum())
    assert multi > 50
