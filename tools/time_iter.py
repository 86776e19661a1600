import time, sys, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aslr_to_amd import scenarios, _abi
if os.environ.get("ASLR_LIB_OVERRIDE"):  # an experimental build of the library
    _abi.lib_path = lambda: os.path.abspath(os.environ["ASLR_LIB_OVERRIDE"])
from aslr_to_amd.engine import Engine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
sc = scenarios.two_dof_vsa_boxddp(B=B, T=100)
low = scenarios.lower(sc)
sp = scenarios.solver_params(sc, fixed_iterations=1)
e = Engine(low)
e.set_candidate(None, None)
st = e._stream()
lib = e.lib
import ctypes as C
def ev(): return torch.cuda.Event(enable_timing=True)
# warmup
for i in range(5): e.iterate(sp, i == 0)
torch.cuda.synchronize()
# per-kernel timing via separate calls
from aslr_to_amd import _abi as A
def timeit(fn, n=20):
    torch.cuda.synchronize(); a, b = ev(), ev(); a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n * 1e3
t_it = timeit(lambda: e.iterate(sp, False), 30)
print("B=%d iterate: %.1f us  -> %.3e knot-steps/s" % (B, t_it, B * 100 / (t_it * 1e-6)))
print("calc_diff (standalone): %.1f us" % timeit(e.calc_diff))
print("calc (standalone): %.1f us" % timeit(e.calc))
print("backward (standalone, store_v): %.1f us" % timeit(lambda: e.backward_pass(sp)))
print("forward (standalone): %.1f us" % timeit(lambda: e.forward_pass(sp)))
print("active:", e.count_active(), "iters", e.traj_i(A.TI_ITER)[:8].tolist())
