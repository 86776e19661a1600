"""Where the cycles of the nx = 8 backward sweep go: loads the region-timing build of the library
(`make -C aslr_to_amd/csrc prof`), runs the bench workload for a few iterations and prints the average shader-clock
cycles per knot per wave spent in each region of the kernel.  Usage: bwd_regions.py [B] [solver]"""
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aslr_to_amd import _abi as A
A.lib_path = lambda: os.path.join(ROOT, "tools", "ubench", "libaslr_to_hip_prof.so")
import numpy as np, torch
from aslr_to_amd import scenarios
from aslr_to_amd.engine import Engine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
solver = sys.argv[2] if len(sys.argv) > 2 else "SolverBoxDDP"
sc = scenarios.two_dof_vsa_boxddp(B=B, T=100)
low = scenarios.lower(sc)
e = Engine(low)
e.set_candidate(None, None)
sp = scenarios.solver_params(sc, solver=solver, fixed_iterations=1)
lib = A.load_library()
out = (ctypes.c_ulonglong * 32)()
WARM = int(sys.argv[3]) if len(sys.argv) > 3 else 20
for i in range(WARM): e.iterate(sp, i == 0)
torch.cuda.synchronize()
lib.aslr_debug_bwd_prof(out, 1)
N = int(sys.argv[4]) if len(sys.argv) > 4 else 30
for i in range(N): e.iterate(sp, False)
torch.cuda.synchronize()
lib.aslr_debug_bwd_prof(out, 0)
v = np.array(list(out), dtype=np.float64)
knots = v[15]
# (nu = 4 runs the team-distributed gains of aslr_team_gains.hpp: regions 6-9 of the per-lane QP stay empty there)
names = ["top: wait for the record", "step 1", "step 2", "DMA issue", "gains inputs (Quu row, bounds) / first active set", "team gains: factor, Newton point, QP / plain gains",
         "  of which: first factor", "  first Newton point, decisions", "  QP line search + gradient", "  QP next active set / factor / Newton point, final factor",
         "tail (Quu k, Vx, Vxx, stores)", "step 4 + loop"]
names.append("gain column K / after the box branch")
v = np.concatenate([v[:12], v[16:17], v[12:16]])
tot = v[:13].sum()
v = np.concatenate([v[:13], np.zeros(3), v[13:]])
print("wave-knots %d, cycles per wave-knot %.0f" % (knots, tot / knots))
for n, c in zip(names, v[:13]): print("  %-32s %8.0f cycles  %5.1f %%" % (n, c / knots, 100 * c / tot))
print("  (old per-lane QP: QP calls per wave-knot, iterations per call, plain executions per wave-knot;\n   team gains: wave-knots that enter the QP loop, line searches per entry, final re-factorisations per wave-knot)\n  %.3f, %.3f, %.3f"
      % (v[17] / knots, v[16] / max(v[17], 1), v[18] / knots))
raw = np.array(list(out), dtype=np.float64)
print("  step lengths tried per line search: %.3f" % (raw[17] / max(raw[12], 1)))
