"""Where the cycles of a C3 calcDiff wave go (profile build, `make -C aslr_to_amd/csrc prof`): shader-clock cycles per wave
until its inputs have arrived, in the knot evaluation, and in the record assembly + stores.  Usage: calc_regions.py"""
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aslr_to_amd import _abi as A
A.lib_path = lambda: os.path.join(ROOT, "tools", "ubench", "libaslr_to_hip_prof.so")
import numpy as np, torch
from aslr_to_amd import scenarios
from aslr_to_amd.engine import Engine
sc = scenarios.two_dof_vsa_boxddp(B=4096, T=100)
e = Engine(scenarios.lower(sc))
e.set_candidate(None, None)
sp = scenarios.solver_params(sc, fixed_iterations=1)
lib = A.load_library()
out = (ctypes.c_ulonglong * 32)()
for i in range(20): e.iterate(sp, i == 0)
torch.cuda.synchronize()
lib.aslr_debug_calc_prof(out, 1)
for i in range(20): e.iterate(sp, False)
torch.cuda.synchronize()
lib.aslr_debug_calc_prof(out, 0)
v = np.array(list(out), dtype=np.float64)
n = v[15]
print("waves %d; cycles per wave: inputs %.0f, knot evaluation %.0f, record assembly and stores %.0f (total %.0f)"
      % (n, v[0] / n, v[1] / n, v[2] / n, v[:3].sum() / n))
