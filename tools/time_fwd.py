"""Forward pass (rollout + trial costs + sum + select) timed stand-alone on a fixed state: cold start, one
calcDiff + backward sweep, then repeated forward passes.  ASLR_LIB_OVERRIDE selects an experimental build of the
library.  Usage: time_fwd.py [B]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if os.environ.get("ASLR_LIB_OVERRIDE"):
    from aslr_to_amd import _abi as _A
    _A.lib_path = lambda: os.path.abspath(os.environ["ASLR_LIB_OVERRIDE"])
import torch
from aslr_to_amd import scenarios
from aslr_to_amd.engine import Engine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
sc = scenarios.two_dof_vsa_boxddp(B=B, T=100)
e = Engine(scenarios.lower(sc))
sp = scenarios.solver_params(sc, fixed_iterations=1)
e.set_candidate(None, None)
e.calc_diff()
e.backward_pass(sp)
def ev(): return torch.cuda.Event(enable_timing=True)
for _ in range(3): e.forward_pass(sp)
torch.cuda.synchronize(); a, b = ev(), ev(); a.record()
for _ in range(20): e.forward_pass(sp)
b.record(); torch.cuda.synchronize()
print("B=%d forward pass on the cold-start state: %.1f us" % (B, a.elapsed_time(b) / 20 * 1e3))
