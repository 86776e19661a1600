"""How often a trial-cost wave (64 consecutive trajectories of one (alpha, knot)) holds a candidate whose joint angles
exceed the 1e5 rad bound of sincos_fast (the library's full-range sincos is then executed by the whole wave), in the
steady state of the C3 bench solve.  Usage: diverged_fraction.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aslr_to_amd import scenarios, _abi as A
from aslr_to_amd.engine import Engine
sc = scenarios.two_dof_vsa_boxddp(B=4096, T=100)
e = Engine(scenarios.lower(sc))
sp = scenarios.solver_params(sc, fixed_iterations=1)
e.set_candidate(None, None)
for it in (5, 15, 30, 50):
    e.iterate_n(sp, it == 5, it if it == 5 else it - done)
    done = it
    torch.cuda.synchronize()
    X = e.region(A.R_XS_TRY)              # [alpha, T+1, B, nx] (de-interleaved copy)
    q = X[..., :2].abs().amax(dim=-1)     # [alpha, T+1, B]
    big = ~(q < 1e5)
    waves = big.reshape(A.NALPHA, 101, 64, 64).any(dim=-1)   # [alpha, t, wave]
    print("after %2d iterations: candidates with |q| >= 1e5: %.2f %%; trial-cost waves holding one: %.1f %%; per alpha index: %s"
          % (it, 100 * big.float().mean().item(), 100 * waves.float().mean().item(),
             " ".join("%.0f" % (100 * waves[a].float().mean().item()) for a in range(A.NALPHA))))
