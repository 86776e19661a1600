"""A / B timing of the per-phase kernel times of the headline iteration (aslr_iterate_timed: HIP events around calc,
backward, forward) for two builds of the library in ONE process-pair run on the same box: the product library and
ASLR_LIB_OVERRIDE.  Usage: time_iter_ab.py [warmup] [n]   (defaults 5, 20: the driver's window)"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if os.environ.get("ASLR_LIB_OVERRIDE"):
    from aslr_to_amd import _abi as _A
    _A.lib_path = lambda: os.path.abspath(os.environ["ASLR_LIB_OVERRIDE"])
import torch
from aslr_to_amd import scenarios
from aslr_to_amd.engine import Engine
W = int(sys.argv[1]) if len(sys.argv) > 1 else 5
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20
sc = scenarios.two_dof_vsa_boxddp(B=4096, T=100)
e = Engine(scenarios.lower(sc))
sp = scenarios.solver_params(sc, fixed_iterations=1, maxiter=W + N)
e.set_candidate(None, None)
for i in range(W): e.iterate(sp, i == 0)
acc = [0.0, 0.0, 0.0]
for i in range(N):
    ms = e.iterate_timed(sp)
    acc = [a + m for a, m in zip(acc, ms)]
print("%s: iterations %d..%d: calc %.1f us, backward %.1f us, forward %.1f us, sum %.1f us; cost_sum %.9e" % (
    os.environ.get("ASLR_LIB_OVERRIDE", "product library"), W, W + N, acc[0] / N * 1e3, acc[1] / N * 1e3, acc[2] / N * 1e3,
    sum(acc) / N * 1e3, float(e.traj_f(0).sum().item())))
