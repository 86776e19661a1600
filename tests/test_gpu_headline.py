"""The headline configuration (BASELINE.json: 2-DoF VSA BoxDDP, B = 4096, T = 100) solved to convergence on the GPU and
by the CPU oracle on the host cores, compared trajectory by trajectory (north_star: 1e-6 on xs / us, 1e-4 on the cost).

The kernels are not bit-identical to the oracle (FMA contraction, closed-form planar dynamics, rsqrt pivots), and a
BoxDDP solve takes ~100 iterations of discrete decisions (line-search accepts, BoxQP active sets, the th_stop exit), so
a few trajectories in a thousand meet a decision whose margin is below the rounding difference and take the other
branch.  The test therefore asserts
  * the bulk: identical iteration counts and status words, and every trajectory converged on both sides within the
    north_star tolerances, EXCEPT a bounded number of exceptions;
  * every exception is accounted for: the per-iteration logs of both sides (aslr_set_iteration_log /
    aslr_cpu_solve_log) agree up to an iteration where ONE discrete decision differs, or show the same decisions
    throughout (rounding drift through an ill-conditioned BoxQP step); and when both sides converged, both end points
    satisfy stop < th_stop and their costs agree to 1e-4 (two valid Crocoddyl-style answers of the same problem).
The per-trajectory table is written by tools/parity_headline.py to profiles/r02/parity_headline_4096.txt.
"""
import os

import numpy as np
import pytest

from aslr_to_amd import _abi, scenarios

import _parity

pytestmark = pytest.mark.gpu

MAX_EXCEPTIONS = 10   # of 4096 (0.25 %); measured 8 (round 2), 6-8 (round 3 builds)
MAX_BEYOND_TOLERANCE = 4  # converged on both sides but further apart than 1e-6 / 1e-4 (measured 2)


def classify(row, sp):
    """Every exception must be one of the understood kinds; anything else fails the test.
    -> "reg-max" | "drift" | "forward-err note" | "exit" | "line search" | None"""
    st_g, st_r = row["st_gpu"], row["st_oracle"]
    note = _abi.ST_FORWARD_ERR
    f, d = row["flip"], row.get("drift")
    if (st_g & _abi.ST_REG_MAX) and (st_r & _abi.ST_REG_MAX):
        # both sides gave up with the regularisation at its maximum (REG_MAX 2 + the FORWARD_ERR note 8 = status 10 in the measured cases), not converged on
        # either: their last accept tests fail by 1e-13 .. 1e-10, so they stop within two iterations of each other
        return "reg-max" if abs(row["it_gpu"] - row["it_oracle"]) <= 2 and not ((st_g | st_r) & _abi.ST_CONVERGED) else None
    if row["it_gpu"] == row["it_oracle"] and (st_g ^ st_r) == note:
        return "forward-err note" if row["dx"] < 1e-9 and row["du"] < 1e-9 else None
    conv = (st_g & _abi.ST_CONVERGED) and (st_r & _abi.ST_CONVERGED)
    if conv and row["stop_gpu"] < sp.th_stop and row["stop_oracle"] < sp.th_stop and row["dcost"] < 1e-4:
        # two converged answers of the same problem.  Rounding drift: the logged decisions are identical up to the
        # iteration where the costs have already parted by 1e-9 relative (an ill-conditioned BoxQP step amplified a
        # rounding difference); or ONE accept / exit test fell the other way by less than its own rounding error
        if d is not None and (f is None or d[0] <= f["iteration"]):
            return "drift"
        if f is None:
            return "drift"
        if f["kind"] == "exit":
            a, b = f["prev_gpu"][_abi.LOG_STOP], f["prev_oracle"][_abi.LOG_STOP]
            return "exit" if (min(a, b) < sp.th_stop <= max(a, b) or abs(a - b) <= 1e-3 * max(a, b)) else None
        if f["kind"] == "line search":
            g, r = f["gpu"], f["oracle"]
            m = min(abs(v[_abi.LOG_DV] - sp.th_acceptstep * v[_abi.LOG_DVEXP]) for v in (g, r))
            return "line search" if m <= 1e-9 * max(1.0, abs(f["prev_oracle"][_abi.LOG_COST])) else None
    return None


def test_headline_batch_full_solves_match_the_oracle_trajectory_by_trajectory(oracle):
    import torch
    from aslr_to_amd.engine import Engine
    B = 4096
    sc = scenarios.two_dof_vsa_boxddp(B=B, T=100, seed=0)
    low = scenarios.lower(sc)
    sp = scenarios.solver_params(sc)
    nth = min(16, len(os.sched_getaffinity(0)))
    ref = oracle.solve(low, sp, nthreads=nth, log_cap=sp.maxiter)
    e = Engine(low)
    e.set_candidate(None, None)
    e.enable_iteration_log(sp.maxiter)
    e.solve(sp, poll_every=4)
    torch.cuda.synchronize()
    gpu = dict(xs=e.region(_abi.R_XS).cpu().numpy(), us=e.region(_abi.R_US).cpu().numpy(),
               traj_f=e.region(_abi.R_TRAJ_F).cpu().numpy(), traj_i=e.region(_abi.R_TRAJ_I).cpu().numpy(),
               log=e.iteration_log().cpu().numpy())
    r = _parity.compare(gpu, ref, sp)
    kinds = [classify(row, sp) for row in r["exceptions"]]
    text = "\n".join("[%s] %s" % (k, _parity.describe(row, sp)) for k, row in zip(kinds, r["exceptions"]))
    print("same iteration count %d, same status %d, converged on both %d, within tolerance %d, exceptions %d\n%s"
          % (r["it_same"], r["st_same"], r["conv_both"], r["within"], len(r["exceptions"]), text))
    assert r["conv_both"] > 0.98 * B
    assert len(r["exceptions"]) <= MAX_EXCEPTIONS, text
    assert r["it_same"] >= B - MAX_EXCEPTIONS and r["st_same"] >= B - MAX_EXCEPTIONS
    # north_star's tolerance on everything that converged on both sides, but a handful (each named and classified below)
    assert r["conv_both"] - r["within"] <= MAX_BEYOND_TOLERANCE, text
    conv_both = ((gpu["traj_i"][_abi.TI_STATUS] & ref["traj_i"][_abi.TI_STATUS] & _abi.ST_CONVERGED) != 0)
    ok = conv_both & (r["dx"] < 1e-6) & (r["du"] < 1e-6) & (r["dc"] < 1e-4)
    assert int(ok.sum()) == r["within"] and np.median(r["dx"][ok]) < 1e-9   # (typical agreement is far inside the tolerance)
    # every exception is of an understood kind
    assert all(k is not None for k in kinds), text
    assert kinds.count("forward-err note") <= 1, text
