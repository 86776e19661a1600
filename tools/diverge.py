"""Iteration at which a GPU solve and the CPU oracle part ways: re-solves with maxiter = 1, 2, ... and prints the
per-trajectory difference of the iterates and of the solver state.  Usage: diverge.py SCENARIO [B] [T] [KMAX]."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from aslr_to_amd import scenarios, _abi as A
from aslr_to_amd.engine import Engine
from oracle import pyoracle as po
name = sys.argv[1] if len(sys.argv) > 1 else "two_dof_vsa_boxddp"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 6
T = int(sys.argv[3]) if len(sys.argv) > 3 else 60
KMAX = int(sys.argv[4]) if len(sys.argv) > 4 else 25
sc = scenarios.SCENARIOS[name](B=B, T=T)
low = scenarios.lower(sc)
e = Engine(low)
np.set_printoptions(linewidth=200, precision=3)
for k in range(1, KMAX + 1):
    sp = scenarios.solver_params(sc, maxiter=k, **({"solver": sys.argv[5]} if len(sys.argv) > 5 else {}))
    r = po.solve(low, sp)
    e.set_candidate(None, None)
    e.solve(sp, poll_every=0)
    torch.cuda.synchronize()
    g = lambda row: e.traj_f(row).cpu().numpy()
    gi = lambda row: e.traj_i(row).cpu().numpy()
    bad = (gi(A.TI_STATUS) != r["traj_i"][A.TI_STATUS]) | (gi(A.TI_ITER) != r["traj_i"][A.TI_ITER]) | (g(A.TF_XREG) != r["traj_f"][A.TF_XREG])
    dx = np.abs(e.region(A.R_XS).cpu().numpy() - r["xs"]).max(axis=(0, 2))
    du = np.abs(e.region(A.R_US).cpu().numpy() - r["us"]).max(axis=(0, 2))
    print("k=%2d state-mismatch %s dx %s du %s step %s" % (k, bad.astype(int), dx, du, r["traj_f"][A.TF_STEP]))
