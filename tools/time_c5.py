import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aslr_to_amd import scenarios, _abi as A
from aslr_to_amd.engine import Engine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
sc = scenarios.talos_arm_sea(B=B, T=150)
if len(sys.argv) > 2 and sys.argv[2] == "SolverBoxDDP":   # motor commands boxed: the QP runs at every knot
    lim = float(sys.argv[3]) if len(sys.argv) > 3 else 2.0
    sc["running"][0].u_lb = np.full(7, -lim)
    sc["running"][0].u_ub = np.full(7, lim)
low = scenarios.lower(sc)
e = Engine(low)
e.set_candidate(None, None)
def ev(): return torch.cuda.Event(enable_timing=True)
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize(); a, b = ev(), ev(); a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n * 1e3
sp = scenarios.solver_params(sc, solver=(sys.argv[2] if len(sys.argv) > 2 else "SolverDDP"), fixed_iterations=1)
for i in range(5): e.iterate(sp, i == 0)
torch.cuda.synchronize()
print("B=%d 7-DoF SEA DDP T=150: iterate %.1f us, backward %.1f us, forward %.1f us, calc_diff %.1f us, calc %.1f us" % (
    B, timeit(lambda: e.iterate(sp, False)), timeit(lambda: e.backward_pass(sp)), timeit(lambda: e.forward_pass(sp)), timeit(e.calc_diff), timeit(e.calc)))

# per-iteration phase times of a cold-started solve (HIP events) next to the trajectories whose backward pass had to be
# regularised and redone INSIDE the kernel (the sweep of a block is repeated until its Cholesky factorisations succeed, like
# SolverDDP.solve's catch / increaseRegularization / retry): the launch lasts as long as its slowest block
e.set_candidate(None, None)
spc = scenarios.solver_params(sc, solver=(sys.argv[2] if len(sys.argv) > 2 else "SolverDDP"), fixed_iterations=1)
prev = None
for i in range(12):
    ms = e.iterate_timed(spc, i == 0)
    xr = e.traj_f(A.TF_XREG).cpu().numpy()
    st = e.traj_i(A.TI_STATUS).cpu().numpy()
    nerr = int(((st & A.ST_BACKWARD_ERR) != 0).sum())
    print("iteration %2d: calcDiff %.3f ms, backward %.3f ms, forward %.3f ms; trajectories with a regularised-and-redone backward pass so far %d, max xreg %.0e"
          % (i, ms[0], ms[1], ms[2], nerr, xr.max()))
# the shard as sub-shards on streams (aslr_set_subshards)
import time
for k in (1, 2, 4):
    e2 = Engine(low)
    e2.set_subshards(k)
    e2.set_candidate(None, None)
    e2.iterate_n(spc, True, 8)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e2.iterate_n(spc, False, 20)
    torch.cuda.synchronize()
    print("%d sub-shard(s): %.1f us per iteration" % (k, (time.perf_counter() - t0) / 20 * 1e6))
    del e2
