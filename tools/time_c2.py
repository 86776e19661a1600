import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aslr_to_amd import scenarios, _abi as A
from aslr_to_amd.engine import Engine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
sc = scenarios.two_dof_sea(B=B, T=100)
low = scenarios.lower(sc)
e = Engine(low)
e.set_candidate(None, None)
def ev(): return torch.cuda.Event(enable_timing=True)
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize(); a, b = ev(), ev(); a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n * 1e3
sp = scenarios.solver_params(sc, solver="SolverDDP", fixed_iterations=1)
for i in range(10): e.iterate(sp, i == 0)
torch.cuda.synchronize()
print("B=%d SEA DDP: iterate %.1f us, backward %.1f us, forward %.1f us, calc_diff %.1f us" % (
    B, timeit(lambda: e.iterate(sp, False)), timeit(lambda: e.backward_pass(sp)), timeit(lambda: e.forward_pass(sp)), timeit(e.calc_diff)))
