"""In-kernel timestamps (no tracer): how long after select_kernel has ended does the first calc wave of the sub-shard's next
iteration start?  Needs the variant build `tools/build_variant.sh stamp "-DASLR_EXP_STAMP" "aslr_forward_nj2 aslr_calc_nj2"`
(select and calc write s_memrealtime stamps into the unused head of VXX).  rocprofv3 --kernel-trace shows 75-150 us there
(tools/trace_gaps.py); these stamps show 3-6 us: the gap is the tracer's."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from aslr_to_amd import scenarios, _abi as A
A.lib_path = lambda: os.path.abspath("tools/ubench/libaslr_to_hip_stamp.so")
from aslr_to_amd.engine import Engine
sc = scenarios.two_dof_vsa_boxddp(B=4096, T=100)
e = Engine(scenarios.lower(sc)); e.set_subshards(4); e.set_candidate(None, None)
sp = scenarios.solver_params(sc, fixed_iterations=1)
e.iterate_n(sp, True, 5); torch.cuda.synchronize()
v = e.region(A.R_VXX).view(-1).view(torch.int64)
v[:2048] = 0
for sub in range(4):
    for k in range(100): v[sub * 512 + 2 * k + 1] = 0x7fffffffffffffff
torch.cuda.synchronize()
t0 = time.perf_counter(); e.iterate_n(sp, False, 30); torch.cuda.synchronize()
print("%.1f us per iteration" % ((time.perf_counter() - t0) / 30 * 1e6))
w = v[:2048].cpu().numpy().reshape(4, 512)
for sub in range(4):
    sel = w[sub, 0:58:2]; calc = w[sub, 1:59:2]
    g = (calc[:-1] - sel[:-1]) / 100.0  # calc start of iteration i+1 minus select end of iteration i
    print("sub-shard %d: select end -> first calc wave of the next iteration: mean %.1f us, min %.1f, max %.1f" % (sub, g.mean(), g.min(), g.max()), np.round(g[:10], 1))
