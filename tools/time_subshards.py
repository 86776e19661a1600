"""Iteration time of the C3 shard with the library's sub-shard streams (aslr_set_subshards), on torch's default stream and
on a side stream.  Usage: time_subshards.py [k ...]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
if os.environ.get("ASLR_LIB_OVERRIDE"):  # an experimental build of the library (tools/ubench/*.so)
    from aslr_to_amd import _abi as _A
    _A.lib_path = lambda: os.path.abspath(os.environ["ASLR_LIB_OVERRIDE"])
from aslr_to_amd import scenarios
from aslr_to_amd.engine import Engine
ks = [int(a) for a in sys.argv[1:]] or [1, 2, 3, 4]
sc = scenarios.two_dof_vsa_boxddp(B=4096, T=100)
low = scenarios.lower(sc)
sp = scenarios.solver_params(sc, fixed_iterations=1)
for side in ((False, True) if not os.environ.get("ASLR_ONE_STREAM_KIND") else (False,)):
    st = torch.cuda.Stream() if side else torch.cuda.current_stream()
    with torch.cuda.stream(st):
        for k in ks:
            e = Engine(low)
            e.set_subshards(k)
            e.set_candidate(None, None)
            e.iterate_n(sp, True, 5)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            e.iterate_n(sp, False, 40)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 40
            print("caller on %s stream, %d sub-shard(s): %.1f us per iteration" % ("a side" if side else "the default", k, dt * 1e6), flush=True)
            del e
