import sys, os, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aslr_to_amd import scenarios, _abi as A
if os.environ.get("ASLR_LIB_OVERRIDE"):  # an experimental build of the library
    A.lib_path = lambda: os.path.abspath(os.environ["ASLR_LIB_OVERRIDE"])
from aslr_to_amd.engine import Engine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
sc = scenarios.two_dof_vsa_boxddp(B=B, T=100)
low = scenarios.lower(sc)
e = Engine(low)
e.set_candidate(None, None)
def ev(): return torch.cuda.Event(enable_timing=True)
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize(); a, b = ev(), ev(); a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n * 1e3
sp_box = scenarios.solver_params(sc, fixed_iterations=1)
sp_ddp = scenarios.solver_params(sc, solver="SolverDDP", fixed_iterations=1)
for i in range(20): e.iterate(sp_box, i == 0)
torch.cuda.synchronize()
e.region(A.R_TRAJ_I)[A.TI_FEASIBLE].fill_(1)
print("backward DDP: %.1f us  Box: %.1f us" % (timeit(lambda: e.backward_pass(sp_ddp)), timeit(lambda: e.backward_pass(sp_box))))
