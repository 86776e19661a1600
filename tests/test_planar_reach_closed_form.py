"""Closed-form frame-placement residual of a planar chain (ChainPlanar::reach_residual in
aslr_to_amd/csrc/aslr_device.hpp, constants from fill_planar / fill_planar_reach in aslr_abi.hip) against the general
path it replaces in cost-only evaluations -- forward kinematics, Mref^-1 oMf and the SE(3) log map of the oracle
(residual_frame_placement.py:13-15) -- for random planar chains, frames turned about z, reference positions and joint
angles well outside (-pi, pi).  The GPU parity tests compare the kernel with the oracle; this pins the derivation."""
import numpy as np
import pytest

from aslr_to_amd import _abi


def _rz(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])


def _closed_form(phi, p, q, fj, Fp, phiF, pref):
    """numpy restatement of ChainPlanar::reach_residual (same operation order)."""
    th, Px, Py, Pz, c, s = 0.0, 0.0, 0.0, 0.0, 1.0, 0.0
    z = np.cumsum(p[:, 2])
    for i in range(fj + 1):
        Px += c * p[i, 0] - s * p[i, 1]
        Py += s * p[i, 0] + c * p[i, 1]
        Pz = z[i]
        th += phi[i] + q[i]
        c, s = np.cos(th), np.sin(th)
    px = (Px + (c * Fp[0] - s * Fp[1])) - pref[0]
    py = (Py + (s * Fp[0] + c * Fp[1])) - pref[1]
    pz = (Pz + Fp[2]) - pref[2]
    psi = th + phiF
    kk = np.rint(psi / (2 * np.pi))
    psi = (psi - kk * 6.28318530717958623200e+00) - kk * 2.44929359829470635445e-16
    cF, sF = np.cos(phiF), np.sin(phiF)
    cr, sr = c * cF - s * sF, s * cF + c * sF
    t, st, ct = abs(psi), abs(sr), cr
    alpha = 1.0 - t * t / 12.0 - t ** 4 / 720.0 if t < 1.220703125e-04 else t * st / (2.0 * (1.0 - ct))
    return np.array([alpha * px + 0.5 * psi * py, alpha * py - 0.5 * psi * px, pz, 0.0, 0.0, psi])


@pytest.mark.parametrize("nj", [2, 3])
def test_closed_form_matches_the_general_log_map(oracle, nj):
    rng = np.random.default_rng(11 + nj)
    worst = 0.0
    for trial in range(200):
        phi = rng.uniform(-1, 1, nj) * (trial % 3 > 0)          # a third of the chains have untilted joint frames
        p = rng.uniform(-0.3, 0.3, (nj, 3))
        chain = _abi.Chain()
        chain.nj = nj
        for i in range(nj):
            R = _rz(phi[i]).reshape(9)
            for k in range(9):
                chain.joint_R[i][k] = R[k]
                chain.inertia[i][k] = np.eye(3).reshape(9)[k]
            for k in range(3):
                chain.joint_p[i][k] = p[i, k]
                chain.axis[i][k] = [0.0, 0.0, 1.0][k]
            chain.mass[i] = 1.0
        scale = [3.0, 30.0, 1e-3, 1e-5][trial % 4]                # ordinary, many turns, near zero, Taylor branch
        q = rng.uniform(-1, 1, nj) * scale
        fj = int(rng.integers(0, nj))
        phiF = rng.uniform(-1, 1) * (trial % 2)
        if trial % 4 >= 2:                                        # small relative angle: cancel the constant part
            q[0] -= phi[:fj + 1].sum() + phiF + q[1:fj + 1].sum()
            q[0] += rng.uniform(-1, 1) * scale
        Fp, pref = rng.uniform(-0.2, 0.2, 3), rng.uniform(-0.3, 0.3, 3)
        Rf, pf = oracle.frame_placement(chain, q, fj, _rz(phiF), Fp)
        ref = oracle.log6(Rf, pf - pref)                           # Mref = (I, pref): Mref^-1 oMf = (R, p - pref)
        got = _closed_form(phi, p, q, fj, Fp, phiF, pref)
        err = np.abs(got - ref).max() / max(1.0, np.abs(ref).max())
        # alpha = t sin t / (2 (1 - cos t)) -- Pinocchio's formula, kept as it is -- loses digits as t -> 0 (the
        # subtraction 1 - cos t has an absolute rounding error of ~1e-16 on a result of t^2 / 2): just above the
        # Taylor switch at t = 1.2e-4 ANY two evaluations of it differ by ~1e-8 relative, the oracle's own included
        t = abs(ref[5])
        noise = 8 * 2.2e-16 / (1.0 - np.cos(t)) if t >= 1.220703125e-04 else 0.0
        worst = max(worst, err - noise)
        assert err < 1e-12 + noise, (trial, got, ref, noise)
    assert worst < 1e-12
