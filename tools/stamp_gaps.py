"""In-kernel timestamps (no tracer): when the first wave of every kernel of a sub-shard's chain starts and its last one ends
(variant build `tools/build_variant.sh stamp "-DASLR_EXP_STAMP" "aslr_forward_nj2 aslr_calc_nj2 aslr_backward_nx8"`; the
kernels write s_memrealtime stamps into the unused head of VXX).  Prints per kernel its mean duration and the idle time to the
next kernel of the chain.  rocprofv3 --kernel-trace shows a 75-150 us gap between select and the next calc (tools/trace_gaps.py);
these stamps show a few microseconds: that gap is the tracer's.  Usage: stamp_gaps.py [k sub-shards = 4]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from aslr_to_amd import scenarios, _abi as A
A.lib_path = lambda: os.path.abspath("tools/ubench/libaslr_to_hip_stamp.so")
from aslr_to_amd.engine import Engine
k = int(sys.argv[1]) if len(sys.argv) > 1 else 4
sc = scenarios.two_dof_vsa_boxddp(B=4096, T=100)
e = Engine(scenarios.lower(sc)); e.set_subshards(k); e.set_candidate(None, None)
sp = scenarios.solver_params(sc, fixed_iterations=1)
e.iterate_n(sp, True, 5); torch.cuda.synchronize()
v = e.region(A.R_VXX).view(-1).view(torch.int64)
w0 = torch.zeros(4 * 1024, dtype=torch.int64)
v[:4096] = w0.to(v.device)
torch.cuda.synchronize()
N = 30
t0 = time.perf_counter(); e.iterate_n(sp, False, N); torch.cuda.synchronize()
print("%d sub-shard(s): %.1f us per iteration (wall clock)" % (k, (time.perf_counter() - t0) / N * 1e6))
names = ["calc", "backward", "rollout (first segment)", "rollout + trial costs", "trial costs (last segment)", "sums", "line search"]
w = v[:4096].cpu().numpy().reshape(4, 1024)
for sub in range(4 if k > 1 else 1):
    if k == 1 and sub > 0: break
    st = w[sub, :N * 16].reshape(N, 8, 2)[3:N - 1] / 100.0   # us; skip the first iterations
    nxt = w[sub, :N * 16].reshape(N, 8, 2)[4:N] / 100.0
    print("sub-shard %d: period %.1f us" % (sub, (st[-1, 0, 0] - st[0, 0, 0]) / (len(st) - 1)))
    for i, nm in enumerate(names):
        dur = st[:, i, 1] - st[:, i, 0]
        gap = (st[:, i + 1, 0] - st[:, i, 1]) if i < 6 else (nxt[:, 0, 0] - st[:, 6, 1])
        print("   %-28s duration %7.1f us   then idle %6.1f us (max %.1f)" % (nm, dur.mean(), gap.mean(), gap.max()))
