"""Throughput of ONE GPU when its shard is split into k sub-shards iterated on k HIP streams (the latency-bound
sweeps of one sub-shard overlap the throughput-bound kernels of the others).  Usage: time_substreams.py [B] [k ...]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
if os.environ.get("ASLR_LIB_OVERRIDE"):  # an experimental build of the library (tools/ubench/*.so)
    from aslr_to_amd import _abi as _A
    _A.lib_path = lambda: os.path.abspath(os.environ["ASLR_LIB_OVERRIDE"])
from aslr_to_amd import scenarios
from aslr_to_amd.engine import Engine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ks = [int(a) for a in sys.argv[2:]] or [1, 2, 4]
sc = scenarios.two_dof_vsa_boxddp(B=B, T=100)
for k in ks:
    engines, streams = [], []
    for i in range(k):
        sub = dict(sc)
        lo, hi = i * B // k, (i + 1) * B // k
        sub["x0"], sub["frame_refs"] = sc["x0"][lo:hi], sc["frame_refs"][lo:hi]
        engines.append(Engine(scenarios.lower(sub)))
        streams.append(torch.cuda.Stream())
    sp = scenarios.solver_params(sc, fixed_iterations=1)
    def run(n, first):
        for it in range(n):
            for e, s in zip(engines, streams):
                with torch.cuda.stream(s):
                    if first and it == 0: e.set_candidate(None, None)
                    e.iterate(sp, first and it == 0)
    run(5, True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(40, False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 40
    print("B=%d in %d sub-shard(s) on %d stream(s): %.1f us per iteration of the whole shard -> %.3e knot-steps/s" % (B, k, k, dt * 1e6, B * 100 / dt))
    del engines
