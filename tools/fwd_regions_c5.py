"""Where the cycles of the 7-DoF team rollout go (C5 shard: 512 trajectories x 150 knots): loads the
region-timing build of the library (`make -C aslr_to_amd/csrc prof`) and prints the average shader-clock cycles per
knot spent by wave 0 of a block in each phase.  Usage: fwd_regions_c5.py [B]"""
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aslr_to_amd import _abi as A
A.lib_path = lambda: os.path.join(ROOT, "tools", "ubench", "libaslr_to_hip_prof.so")
import numpy as np, torch
from aslr_to_amd import scenarios
from aslr_to_amd.engine import Engine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
sc = scenarios.talos_arm_sea(B=B, T=150)
e = Engine(scenarios.lower(sc))
e.set_candidate(None, None)
sp = scenarios.solver_params(sc, solver="SolverDDP", fixed_iterations=1)
lib = A.load_library()
out = (ctypes.c_ulonglong * 32)()
for i in range(5): e.iterate(sp, i == 0)
torch.cuda.synchronize()
lib.aslr_debug_fwd_prof7(out, 1)
for i in range(10): e.iterate(sp, False)
torch.cuda.synchronize()
lib.aslr_debug_fwd_prof7(out, 0)
v = np.array(list(out), dtype=np.float64)
knots = v[15]
# slot i = the cycles between the mark before it and mark i (marks sit at the START of the phase named in the kernel)
names = ["top: fence, stage prefetch to LDS, fence, next prefetch, candidate state store", "control law (row of K dx), store",
         "joint rotation to LDS, fence", "coupling / motor torques", "RNEA sweep per lane (columns of M, nonlinear effects), fence",
         "column of M^-1, fence", "(after the loop)", "accelerations, Euler step, loop"]
tot = v[:8].sum()
print("wave-0 knots %d, cycles per knot %.0f" % (knots, tot / knots))
for n, c in zip(names, v[:8]): print("  %-52s %8.0f cycles  %5.1f %%" % (n, c / knots, 100 * c / tot))
