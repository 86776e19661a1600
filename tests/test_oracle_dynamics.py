"""Pin the CPU oracle's rigid-body pieces with independent substitutes:
a sympy Lagrangian derivation of the planar 2R arm, finite differences, and group identities.
No GPU, no product code paths.
"""
import numpy as np
import pytest

from aslr_to_amd import example_robot_data


@pytest.fixture(scope="module")
def arm2():
    m = example_robot_data.load("asr_twodof").model
    m.gravity.linear = np.array([9.81, 0.0, 0.0])
    return m


@pytest.fixture(scope="module")
def lagrangian_2r(arm2):
    """M(q), nle(q, v), d(tau)/dq, d(tau)/dv of the planar 2R arm from the Lagrangian, with sympy."""
    import sympy as sp
    q1, q2, v1, v2, a1, a2 = sp.symbols("q1 q2 v1 v2 a1 a2")
    j1, j2 = arm2.joints
    gx, gy = arm2.gravity.linear[:2]
    l1x, l1y = j2.placement.translation[:2]
    def rot(t):
        return sp.Matrix([[sp.cos(t), -sp.sin(t)], [sp.sin(t), sp.cos(t)]])
    p1 = rot(q1) * sp.Matrix(j1.com[:2])
    o2 = rot(q1) * sp.Matrix([l1x, l1y])
    p2 = o2 + rot(q1 + q2) * sp.Matrix(j2.com[:2])
    q, v, a = sp.Matrix([q1, q2]), sp.Matrix([v1, v2]), sp.Matrix([a1, a2])
    vc1, vc2 = p1.jacobian(q) * v, p2.jacobian(q) * v
    T = (sp.Rational(1, 2) * (j1.mass * vc1.dot(vc1) + j2.mass * vc2.dot(vc2))
         + sp.Rational(1, 2) * (j1.inertia[2, 2] * v1 ** 2 + j2.inertia[2, 2] * (v1 + v2) ** 2))
    g = sp.Matrix([gx, gy])
    U = -(j1.mass * g.dot(p1) + j2.mass * g.dot(p2))
    L = T - U
    dLdv = sp.Matrix([L]).jacobian(v).T
    tau = dLdv.jacobian(q) * v + dLdv.jacobian(v) * a - sp.Matrix([L]).jacobian(q).T
    M = dLdv.jacobian(v)
    nle = tau.subs({a1: 0, a2: 0})
    args = (q1, q2, v1, v2, a1, a2)
    return dict(M=sp.lambdify(args, M), nle=sp.lambdify(args, nle), tau=sp.lambdify(args, tau),
                dq=sp.lambdify(args, tau.jacobian(q)), dv=sp.lambdify(args, tau.jacobian(v)))


def test_2r_inertia_bias_and_rnea_derivatives_match_lagrangian(oracle, arm2, lagrangian_2r):
    c = arm2.to_struct()
    rng = np.random.default_rng(0)
    for _ in range(20):
        q, v, a = rng.uniform(-3, 3, 2), rng.uniform(-2, 2, 2), rng.uniform(-5, 5, 2)
        args = (*q, *v, *a)
        np.testing.assert_allclose(oracle.crba(c, q), np.array(lagrangian_2r["M"](*args), dtype=float), rtol=0, atol=1e-13)
        np.testing.assert_allclose(oracle.nle(c, q, v), np.array(lagrangian_2r["nle"](*args), dtype=float).ravel(), atol=1e-12)
        np.testing.assert_allclose(oracle.rnea(c, q, v, a), np.array(lagrangian_2r["tau"](*args), dtype=float).ravel(), atol=1e-12)
        dq, dv = oracle.rnea_derivatives(c, q, v, a)
        np.testing.assert_allclose(dq, np.array(lagrangian_2r["dq"](*args), dtype=float), atol=1e-11)
        np.testing.assert_allclose(dv, np.array(lagrangian_2r["dv"](*args), dtype=float), atol=1e-11)


@pytest.mark.parametrize("name", ["asr_twodof", "double_pendulum", "talos_arm"])
def test_rnea_derivatives_match_central_differences(oracle, name):
    c = example_robot_data.load(name).model.to_struct()
    nj = c.nj
    rng = np.random.default_rng(1)
    q, v, a = rng.uniform(-1.5, 1.5, nj), rng.uniform(-1, 1, nj), rng.uniform(-2, 2, nj)
    dq, dv = oracle.rnea_derivatives(c, q, v, a)
    h = 1e-6
    for j in range(nj):
        e = np.zeros(nj)
        e[j] = h
        fdq = (oracle.rnea(c, q + e, v, a) - oracle.rnea(c, q - e, v, a)) / (2 * h)
        fdv = (oracle.rnea(c, q, v + e, a) - oracle.rnea(c, q, v - e, a)) / (2 * h)
        np.testing.assert_allclose(dq[:, j], fdq, atol=2e-7)
        np.testing.assert_allclose(dv[:, j], fdv, atol=2e-7)


@pytest.mark.parametrize("name", ["asr_twodof", "double_pendulum", "talos_arm"])
def test_inertia_matrix_is_spd_and_consistent_with_rnea(oracle, name):
    c = example_robot_data.load(name).model.to_struct()
    nj = c.nj
    rng = np.random.default_rng(2)
    q, v, a = rng.uniform(-2, 2, nj), rng.uniform(-1, 1, nj), rng.uniform(-2, 2, nj)
    M = oracle.crba(c, q)
    assert np.allclose(M, M.T, atol=0)
    assert np.linalg.eigvalsh(M).min() > 0
    np.testing.assert_allclose(oracle.rnea(c, q, v, a), M.dot(a) + oracle.nle(c, q, v), atol=1e-11)


def test_frame_jacobian_matches_finite_differences_of_the_placement(oracle):
    model = example_robot_data.load("talos_arm").model
    c = model.to_struct()
    f = model.frames[model.getFrameId("gripper_left_joint")]
    fR, fp = f.placement.rotation, f.placement.translation
    rng = np.random.default_rng(3)
    q = rng.uniform(-1, 1, c.nj)
    J = oracle.frame_jacobian(c, q, f.parent, fR, fp)
    R0, p0 = oracle.frame_placement(c, q, f.parent, fR, fp)
    h = 1e-6
    for j in range(c.nj):
        e = np.zeros(c.nj)
        e[j] = h
        R1, p1 = oracle.frame_placement(c, q + e, f.parent, fR, fp)
        # LOCAL twist: [R0^T dp ; vee(R0^T dR)]
        lin = R0.T.dot(p1 - p0) / h
        W = R0.T.dot(R1 - R0) / h
        ang = np.array([W[2, 1] - W[1, 2], W[0, 2] - W[2, 0], W[1, 0] - W[0, 1]]) / 2
        np.testing.assert_allclose(J[:3, j], lin, atol=5e-6)
        np.testing.assert_allclose(J[3:, j], ang, atol=5e-6)


def test_log6_exp6_roundtrip_and_jlog6_is_the_right_jacobian(oracle):
    rng = np.random.default_rng(4)
    for scale in (1e-9, 1e-3, 0.5, 2.5, 3.13):
        r = rng.normal(size=6)
        r[3:] *= scale / np.linalg.norm(r[3:])
        R, p = oracle.exp6(r)
        np.testing.assert_allclose(R.dot(R.T), np.eye(3), atol=1e-13)
        np.testing.assert_allclose(oracle.log6(R, p), r, atol=2e-7 if scale > 3 else 1e-9)
        if 1e-3 <= scale <= 2.5:
            J = oracle.jlog6(R, p)
            h = 1e-7
            for k in range(6):
                d = np.zeros(6)
                d[k] = h
                Rd, pd = oracle.exp6(d)
                Rn, pn = R.dot(Rd), R.dot(pd) + p  # M * exp(d): right perturbation
                fd = (oracle.log6(Rn, pn) - oracle.log6(R, p)) / h
                np.testing.assert_allclose(J[:, k], fd, atol=5e-6)
