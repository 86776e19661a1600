"""Histogram of the accepted step-length index per lock-step iteration of the bench workload (C3): which of the
10 candidates of the line search the solver takes (10 = none accepted).  Usage: alpha_hist.py [B] [ITERS]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from aslr_to_amd import scenarios, _abi as A
from aslr_to_amd.engine import Engine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
N = int(sys.argv[2]) if len(sys.argv) > 2 else 55
sc = scenarios.two_dof_vsa_boxddp(B=B, T=100)
low = scenarios.lower(sc)
e = Engine(low)
e.set_candidate(None, None)
sp = scenarios.solver_params(sc, fixed_iterations=1)
tot = np.zeros(12, dtype=np.int64)
for i in range(N):
    e.iterate(sp, i == 0)
    torch.cuda.synchronize()
    acc = e.traj_i(A.TI_ACCEPTED).cpu().numpy()
    h = np.bincount(np.where(acc < 0, 10, acc), minlength=11)
    if i >= 5: tot[:11] += h
    if i % 5 == 0: print("it %2d accepted-index histogram %s" % (i, h))
print("iterations 5..%d: %s" % (N - 1, tot[:11]), "fractions", np.round(tot[:11] / tot[:11].sum(), 3))
