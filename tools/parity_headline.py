"""The headline batch (C3: 2-DoF VSA BoxDDP, B = 4096, T = 100, full solves of up to 400 iterations) on the GPU against
the CPU oracle, trajectory by trajectory, with the first differing solver decision of every exception named from the
per-iteration logs of both sides (tests/_parity.py).  Usage: parity_headline.py [B] [out.txt] [SCENARIO] [T]
(SCENARIO defaults to two_dof_vsa_boxddp, T to 100; e.g. `parity_headline.py 64 out.txt two_dof_vsa_modified 200` is the
example examples/two_dof_vsa_modified.py solves)
ASLR_LIB_OVERRIDE selects another build of the library (e.g. one compiled with -ffp-contract=off)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from aslr_to_amd import _abi as A
if os.environ.get("ASLR_LIB_OVERRIDE"):
    A.lib_path = lambda: os.path.abspath(os.environ["ASLR_LIB_OVERRIDE"])
from aslr_to_amd import scenarios
from aslr_to_amd.engine import Engine
from oracle import pyoracle as po
import _parity
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
NAME = sys.argv[3] if len(sys.argv) > 3 else "two_dof_vsa_boxddp"
TT = int(sys.argv[4]) if len(sys.argv) > 4 else 100
sc = scenarios.SCENARIOS[NAME](B=B, T=TT, seed=0)
low = scenarios.lower(sc)
sp = scenarios.solver_params(sc)
nth = min(16, len(os.sched_getaffinity(0)))
t0 = time.time(); ref = po.solve(low, sp, nthreads=nth, log_cap=sp.maxiter); tc = time.time() - t0
e = Engine(low); e.set_candidate(None, None); e.enable_iteration_log(sp.maxiter)
torch.cuda.synchronize(); t0 = time.time(); e.solve(sp, poll_every=4); torch.cuda.synchronize(); tg = time.time() - t0
gpu = dict(xs=e.region(A.R_XS).cpu().numpy(), us=e.region(A.R_US).cpu().numpy(), traj_f=e.region(A.R_TRAJ_F).cpu().numpy(),
           traj_i=e.region(A.R_TRAJ_I).cpu().numpy(), log=e.iteration_log().cpu().numpy())
r = _parity.compare(gpu, ref, sp)
print("library: %s%s%s" % (A.lib_path(), "  ASLR_NO_PLANAR=1" if os.environ.get("ASLR_NO_PLANAR") else "",
                           "  " + os.environ.get("ASLR_NOTE", "")), file=out)
print("%s %s B=%d T=%d seed 0, th_stop %.0e, maxiter %d: oracle %.1f s (%d threads), gpu %.2f s"
      % (NAME, sc["solver"], B, TT, sp.th_stop, sp.maxiter, tc, nth, tg), file=out)
print("same iteration count %d / %d; same status word %d / %d; converged oracle %d, gpu %d, both %d; of those within "
      "1e-6 (xs, us) and 1e-4 (cost): %d  (max |dx| %.2e |du| %.2e |dcost| %.2e among them)"
      % (r["it_same"], B, r["st_same"], B, r["conv_oracle"], r["conv_gpu"], r["conv_both"], r["within"], r["max_dx"],
         r["max_du"], r["max_dc"]), file=out)
print("exceptions (different iteration count or status word, or converged on both sides beyond the tolerance): %d"
      % len(r["exceptions"]), file=out)
for row in r["exceptions"]:
    print("  " + _parity.describe(row, sp), file=out)
