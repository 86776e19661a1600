"""Why a status word differs by ASLR_ST_FORWARD_ERR only: for trajectory TRAJ of the headline batch, replay iteration
ITER on the GPU (solve ITER iterations, then calcDiff + backward + the 10-alpha forward pass) and let the oracle roll out
every step length from the GPU's own gains; print, per step length, which side flags the trial, the knot at which
|x|_inf first exceeds 1e15 / 1e30 on either side and the largest |x|.  Usage: forward_err_probe.py TRAJ ITER"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from aslr_to_amd import scenarios, _abi as A
from aslr_to_amd.engine import Engine
from oracle import pyoracle as po
traj, it = int(sys.argv[1]), int(sys.argv[2])
sc = scenarios.two_dof_vsa_boxddp(B=4096, T=100, seed=0)
sc["x0"], sc["frame_refs"] = sc["x0"][traj:traj + 1], sc["frame_refs"][traj:traj + 1]
low = scenarios.lower(sc)
e = Engine(low)
e.set_candidate(None, None)
sp = scenarios.solver_params(sc, maxiter=it)
e.solve(sp, poll_every=0)
torch.cuda.synchronize()
print("after %d iterations: status %d, xreg %.1e, feasible %d" % (it, e.traj_i(A.TI_STATUS)[0], e.traj_f(A.TF_XREG)[0], e.traj_i(A.TI_FEASIBLE)[0]))
ref = po.solve(low, sp)
print("oracle status %d; |dx| %.2e" % (ref["traj_i"][A.TI_STATUS][0], np.abs(e.region(A.R_XS).cpu().numpy() - ref["xs"]).max()))
e.calc_diff(); e.backward_pass(sp); e.forward_pass(sp)
torch.cuda.synchronize()
X, U = e.region(A.R_XS).cpu().numpy(), e.region(A.R_US).cpu().numpy()
K, k = e.region(A.R_KGAIN).cpu().numpy(), e.region(A.R_KFF).cpu().numpy()
XT = e.region(A.R_XS_TRY).cpu().numpy()
fails = e.region(A.R_TRAJ_I)[A.TI_TRYFAIL0:A.TI_TRYFAIL0 + A.NALPHA, 0].cpu().numpy()
ctry = e.region(A.R_TRAJ_F)[A.TF_COST_TRY0:A.TF_COST_TRY0 + A.NALPHA, 0].cpu().numpy()
def first_over(x, lim):
    m = np.abs(x).max(axis=1)
    w = np.nonzero(~(m < lim))[0]
    return int(w[0]) if w.size else -1
for a in range(A.NALPHA):
    xs_o, us_o, c_o, f_o = po.forward_pass(low, sp, 0.5 ** a, X, U, K, k, feasible=1)
    g, o = XT[a][:, 0, :], xs_o[:, 0, :]
    print("alpha 2^-%d: gpu fail %d (cost %s) oracle fail %d (cost %.6e); max|x| gpu %.3e oracle %.3e; first knot with |x|_inf >= 1e15: %d / %d, >= 1e30: %d / %d"
          % (a, fails[a], "%.6e" % ctry[a], f_o[0], c_o[0], np.nanmax(np.abs(g)), np.nanmax(np.abs(o)), first_over(g, 1e15), first_over(o, 1e15),
             first_over(g, 1e30), first_over(o, 1e30)))
