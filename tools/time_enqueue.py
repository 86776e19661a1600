"""Host time to ENQUEUE n iterations (the aslr_iterate_n call returns when everything is queued) against the GPU time they
take, per sub-shard count: is the host the bottleneck of the sub-sharded schedule?  Usage: time_enqueue.py [k ...]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
from aslr_to_amd import scenarios
from aslr_to_amd.engine import Engine
ks = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 6, 8]
sc = scenarios.two_dof_vsa_boxddp(B=4096, T=100)
low = scenarios.lower(sc)
sp = scenarios.solver_params(sc, fixed_iterations=1)
for k in ks:
    e = Engine(low)
    e.set_subshards(k)
    e.set_candidate(None, None)
    e.iterate_n(sp, True, 5)
    torch.cuda.synchronize()
    n = 40
    t0 = time.perf_counter()
    e.iterate_n(sp, False, n)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%d sub-shard(s): enqueue %.1f us per iteration (host), complete %.1f us per iteration" % (k, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6), flush=True)
    del e
