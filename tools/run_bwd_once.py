import sys, os, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aslr_to_amd import scenarios, _abi as A
from aslr_to_amd.engine import Engine
B = 4096
sc = scenarios.two_dof_vsa_boxddp(B=B, T=100)
low = scenarios.lower(sc)
e = Engine(low)
e.set_candidate(None, None)
sp_box = scenarios.solver_params(sc, fixed_iterations=1)
sp_ddp = scenarios.solver_params(sc, solver="SolverDDP", fixed_iterations=1)
for i in range(10): e.iterate(sp_box, i == 0)
torch.cuda.synchronize()
e.region(A.R_TRAJ_I)[A.TI_FEASIBLE].fill_(1)
mode = sys.argv[1] if len(sys.argv) > 1 else "ddp"
for _ in range(3):
    e.backward_pass(sp_ddp if mode == "ddp" else sp_box)
    e.forward_pass(sp_box)
    e.calc_diff()
torch.cuda.synchronize()
