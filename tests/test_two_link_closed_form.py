"""Closed-form joint-space inertia and nonlinear effects of a planar two-link chain (ChainPlanar<2>::mass2 / nle2 in
aslr_to_amd/csrc/aslr_device.hpp, constants from fill_planar in aslr_abi.hip) against the planar Newton-Euler
recursion they replace, for random link offsets, joint placements, centres of mass and gravity directions.
The GPU parity tests compare the kernels with the oracle's general 3-D recursion; this test pins the derivation."""
import numpy as np


def _recursion(phi, p, m, com, izz, g, q, v, a):
    X = []
    for i in range(2):
        c, s = np.cos(q[i]), np.sin(q[i])
        X.append((np.cos(phi[i]) * c - np.sin(phi[i]) * s, np.sin(phi[i]) * c + np.cos(phi[i]) * s, p[i, 0], p[i, 1]))

    def actinv(Xi, mv):
        c, s, px, py = Xi
        w, x, y = mv
        vx, vy = x - py * w, y + px * w
        return np.array([w, c * vx + s * vy, c * vy - s * vx])

    def fact(Xi, f):
        c, s, px, py = Xi
        n, x, y = f
        lx, ly = c * x - s * y, s * x + c * y
        return np.array([n + px * ly - py * lx, lx, ly])

    def crm(a_, b_):
        return np.array([0.0, b_[0] * a_[2] - a_[0] * b_[2], a_[0] * b_[1] - b_[0] * a_[1]])

    def crf(a_, f):
        return np.array([a_[1] * f[2] - a_[2] * f[1], -a_[0] * f[2], a_[0] * f[1]])

    def inertia(i, vv):
        lx = m[i] * (vv[1] - com[i, 1] * vv[0])
        ly = m[i] * (vv[2] + com[i, 0] * vv[0])
        return np.array([izz[i] * vv[0] + com[i, 0] * ly - com[i, 1] * lx, lx, ly])

    vp, ap, f = np.zeros(3), np.array([0.0, -g[0], -g[1]]), [None, None]
    for i in range(2):
        vi = actinv(X[i], vp)
        vi[0] += v[i]
        ai = actinv(X[i], ap) + crm(vi, np.array([v[i], 0.0, 0.0]))
        ai[0] += a[i]
        f[i] = inertia(i, ai) + crf(vi, inertia(i, vi))
        vp, ap = vi, ai
    tau = np.zeros(2)
    for i in (1, 0):
        tau[i] = f[i][0]
        if i > 0:
            f[i - 1] = f[i - 1] + fact(X[i], f[i])
    return tau, X


def _closed_form(p, m, com, izz, g, X, v):
    J1 = izz[0] + m[0] * (com[0] @ com[0])
    J2 = izz[1] + m[1] * (com[1] @ com[1])
    p2 = p[1]
    two = [J1 + J2 + m[1] * (p2 @ p2), J2, m[1] * (p2 @ com[1]), m[1] * (p2[1] * com[1, 0] - p2[0] * com[1, 1]),
           -(m[0] * com[0, 1] + m[1] * p2[1]), m[0] * com[0, 0] + m[1] * p2[0], -m[1] * com[1, 1], m[1] * com[1, 0]]
    c1, s1, c2, s2 = X[0][0], X[0][1], X[1][0], X[1][1]
    E = two[2] * c2 + two[3] * s2
    Ep = two[3] * c2 - two[2] * s2
    M = np.array([[two[0] + 2.0 * E, two[1] + E], [two[1] + E, two[1]]])
    ux, uy = c1 * g[0] + s1 * g[1], c1 * g[1] - s1 * g[0]
    wx, wy = c2 * ux + s2 * uy, c2 * uy - s2 * ux
    G2 = -(wx * two[6] + wy * two[7])
    G1 = G2 - (ux * two[4] + uy * two[5])
    return M, np.array([Ep * ((2.0 * v[0] + v[1]) * v[1]) + G1, G2 - Ep * (v[0] * v[0])])


def test_closed_form_matches_the_recursion():
    rng = np.random.default_rng(7)
    for _ in range(50):
        phi, p = rng.uniform(-1, 1, 2), rng.uniform(-0.3, 0.3, (2, 2))
        m, com, izz = rng.uniform(0.2, 2, 2), rng.uniform(-0.2, 0.2, (2, 2)), rng.uniform(0.01, 0.1, 2)
        g, q, v = rng.uniform(-10, 10, 2), rng.uniform(-3, 3, 2), rng.uniform(-3, 3, 2)
        nle, X = _recursion(phi, p, m, com, izz, g, q, v, np.zeros(2))
        bias, _ = _recursion(phi, p, m, com, izz, g, q, np.zeros(2), np.zeros(2))
        M = np.column_stack([_recursion(phi, p, m, com, izz, g, q, np.zeros(2), e)[0] - bias for e in np.eye(2)])
        Mc, nc = _closed_form(p, m, com, izz, g, X, v)
        assert np.abs(M - Mc).max() < 1e-13
        assert np.abs(nle - nc).max() < 1e-12
