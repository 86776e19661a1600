import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from aslr_to_amd import scenarios, _abi as A
from aslr_to_amd.engine import Engine
from oracle import pyoracle as po
sc = scenarios.two_dof_sea(B=6, T=20)
sc["running"][0].differential.costs.costs["uReg"].weight = -5e-3
low = scenarios.lower(sc)
e = Engine(low)
np.set_printoptions(linewidth=200, precision=6)
for k in range(1, 21):
    sp = scenarios.solver_params(sc, maxiter=k)
    r = po.solve(low, sp)
    e.set_candidate(None, None)
    e.solve(sp, poll_every=0)
    torch.cuda.synchronize()
    g = lambda row: e.traj_f(row).cpu().numpy()
    gi = lambda row: e.traj_i(row).cpu().numpy()
    bad = (gi(A.TI_STATUS) != r["traj_i"][A.TI_STATUS]) | (gi(A.TI_ITER) != r["traj_i"][A.TI_ITER]) | (g(A.TF_XREG) != r["traj_f"][A.TF_XREG])
    dx = np.abs(e.region(A.R_XS).cpu().numpy() - r["xs"]).max(axis=(0, 2))
    print("k=%2d mismatch %s  dx %s" % (k, bad.astype(int), dx))
    if bad.any():
        for b in np.nonzero(bad)[0]:
            print("   b=%d gpu: st %d it %d xreg %g step %g cost %.12g dV %g dVexp %g d1 %g d2 %g feas %d" % (b, gi(A.TI_STATUS)[b], gi(A.TI_ITER)[b], g(A.TF_XREG)[b], g(A.TF_STEP)[b], g(A.TF_COST)[b], g(A.TF_DV)[b], g(A.TF_DVEXP)[b], g(A.TF_D1)[b], g(A.TF_D2)[b], gi(A.TI_FEASIBLE)[b]))
            f = r["traj_f"]; i = r["traj_i"]
            print("       cpu: st %d it %d xreg %g step %g cost %.12g dV %g dVexp %g d1 %g d2 %g feas %d" % (i[A.TI_STATUS][b], i[A.TI_ITER][b], f[A.TF_XREG][b], f[A.TF_STEP][b], f[A.TF_COST][b], f[A.TF_DV][b], f[A.TF_DVEXP][b], f[A.TF_D1][b], f[A.TF_D2][b], i[A.TI_FEASIBLE][b]))
        break
