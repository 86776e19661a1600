"""Generate the golden fixtures of tests/golden/ from the CPU oracle (oracle/).

The reference holds no golden vectors and cannot be imported here (Crocoddyl / Pinocchio are not
installable), so these fixtures are numbers produced by this repository's own CPU
restatement, which tests/test_oracle_*.py pin with independent substitutes.  They freeze (i) the
synthetic robot tables (example_robot_data.TABLE_VERSION), (ii) per-knot calc / calcDiff records at
seeded points, (iii) converged solver outputs, so that a regression in either the oracle or the HIP
path is caught without the other.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from aslr_to_amd import _abi, example_robot_data, scenarios  # noqa: E402
from oracle import pyoracle  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

KNOT_CASES = [("two_dof_vsa_boxddp", dict(B=6, T=4)), ("two_dof_sea", dict(B=6, T=4)),
              ("double_pendulum", dict(T=4)), ("talos_arm_sea", dict(B=3, T=2))]
SOLVE_CASES = [("two_dof_vsa_boxddp", dict(B=4, T=100), "SolverBoxDDP"), ("two_dof_sea", dict(B=4, T=100), "SolverDDP"),
               ("two_dof_sea", dict(B=4, T=100), "SolverFDDP"), ("double_pendulum", dict(T=10), "SolverDDP")]


def candidate(low, seed):
    rng = np.random.default_rng(seed)
    xs = rng.uniform(-0.8, 0.8, (low.T + 1, low.B, low.nx))
    us = rng.uniform(-1.0, 1.0, (low.T, low.B, low.nu))
    if low.dam == _abi.DAM_VSA:
        us[..., low.nu // 2:] = rng.uniform(0.1, 5.0, (low.T, low.B, low.nu // 2))
    return xs, us


def main():
    out = {"table_version": np.array(example_robot_data.TABLE_VERSION)}
    for name, kw in KNOT_CASES:
        sc = scenarios.SCENARIOS[name](**kw)
        low = scenarios.lower(sc)
        xs, us = candidate(low, 11)
        xnext, cost, deriv = pyoracle.calc_diff(low, xs, us)
        out["knot/%s/xs" % name], out["knot/%s/us" % name] = xs, us
        out["knot/%s/xnext" % name], out["knot/%s/cost" % name], out["knot/%s/deriv" % name] = xnext, cost, deriv
    for name, kw, solver in SOLVE_CASES:
        sc = scenarios.SCENARIOS[name](**kw)
        low = scenarios.lower(sc)
        r = pyoracle.solve(low, scenarios.solver_params(sc, solver=solver))
        key = "solve/%s/%s" % (name, solver)
        out[key + "/xs"], out[key + "/us"] = r["xs"], r["us"]
        out[key + "/cost"] = r["traj_f"][_abi.TF_COST]
        out[key + "/iters"] = r["traj_i"][_abi.TI_ITER]
        out[key + "/status"] = r["traj_i"][_abi.TI_STATUS]
    np.savez_compressed(os.path.join(HERE, "golden_v%d.npz" % example_robot_data.TABLE_VERSION), **out)
    print("wrote", len(out), "arrays")


if __name__ == "__main__":
    main()
