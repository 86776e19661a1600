"""How unevenly the waves of one backward sweep finish (a launch lasts as long as its slowest wave): needs the experimental build
`tools/build_variant.sh wt "-DASLR_EXP_WAVETIME" "aslr_backward_nx8"` (each wave writes its duration into the head of VX).
tools/wave_times.py [B] [WARM]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from aslr_to_amd import scenarios, _abi as A
A.lib_path = lambda: os.path.abspath(os.environ.get("ASLR_LIB_OVERRIDE", "tools/ubench/libaslr_to_hip_wt.so"))
from aslr_to_amd.engine import Engine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
W = int(sys.argv[2]) if len(sys.argv) > 2 else 10
sc = scenarios.two_dof_vsa_boxddp(B=B, T=100)
e = Engine(scenarios.lower(sc))
e.set_candidate(None, None)
sp = scenarios.solver_params(sc, fixed_iterations=1)
for i in range(W): e.iterate(sp, i == 0)
torch.cuda.synchronize()
nw = B // 4
for it in range(W, W + 6):
    e.iterate(sp, False)
    torch.cuda.synchronize()
    t = e.region(A.R_VX).flatten()[:nw].cpu().numpy() / 100.0  # us
    print("iteration %2d: waves %d  mean %.1f us  median %.1f  p90 %.1f  p99 %.1f  max %.1f  (max / mean %.3f)"
          % (it, nw, t.mean(), np.median(t), np.percentile(t, 90), np.percentile(t, 99), t.max(), t.max() / t.mean()))
