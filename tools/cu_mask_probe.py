"""Is the backward sweep slower at 4096 trajectories than at 1024 (one wave per SIMD against one per CU) because the four
waves of a CU share something, or because the whole chip is busy (clocks, memory system)?  Runs the 1024-trajectory
sweep on a stream restricted to 64 CUs (hipExtStreamCreateWithCUMask): 256 waves, four per CU, three quarters of the
chip idle.  tools/cu_mask_probe.py [ncu=64]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aslr_to_amd import scenarios, _abi as A
from aslr_to_amd.engine import Engine

hip = C.CDLL("libamdhip64.so")
ncu = int(sys.argv[1]) if len(sys.argv) > 1 else 64
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
torch.cuda.init(); torch.zeros(1, device="cuda")
def masked_stream(bits):
    words = (C.c_uint32 * 8)(*[(bits >> (32 * i)) & 0xffffffff for i in range(8)])
    st = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value)
def ev(): return torch.cuda.Event(enable_timing=True)
sc = scenarios.two_dof_vsa_boxddp(B=B, T=100)
low = scenarios.lower(sc)
e = Engine(low)
e.set_candidate(None, None)
sp_box = scenarios.solver_params(sc, fixed_iterations=1)
sp_ddp = scenarios.solver_params(sc, solver="SolverDDP", fixed_iterations=1)
for i in range(20): e.iterate(sp_box, i == 0)
torch.cuda.synchronize()
e.region(A.R_TRAJ_I)[A.TI_FEASIBLE].fill_(1)
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize(); a, b = ev(), ev(); a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n * 1e3
print("whole chip: DDP %.1f us  Box %.1f us" % (timeit(lambda: e.backward_pass(sp_ddp)), timeit(lambda: e.backward_pass(sp_box))))
for name, bits in (("first %d mask bits" % ncu, (1 << ncu) - 1),
                   ("every %dth bit" % (256 // ncu), sum(1 << i for i in range(0, 256, 256 // ncu)))):
    with torch.cuda.stream(masked_stream(bits)):
        print("%s: DDP %.1f us  Box %.1f us" % (name, timeit(lambda: e.backward_pass(sp_ddp)), timeit(lambda: e.backward_pass(sp_box))))
