"""Where a GPU solve and the oracle's part ways without a logged decision differing: for trajectory TRAJ of the headline
batch, bring both to the start of iteration ITER, then run ONE backward pass on both sides from the GPU's own records,
controls, warm start and regularisation and print, knot by knot (from the last), the relative difference of the
gains, the clamped set each side ends with and the conditioning of Quu.  Usage: backward_probe.py TRAJ ITER"""
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
for n in (it, it + 1):
    sp = scenarios.solver_params(sc, maxiter=n)
    e.set_candidate(None, None); e.solve(sp, poll_every=0); torch.cuda.synchronize()
    ref = po.solve(low, sp)
    print("after %d iterations: |dx| %.3e |du| %.3e, cost gpu %.12e oracle %.12e, xreg %.1e / %.1e"
          % (n, np.abs(e.region(A.R_XS).cpu().numpy() - ref["xs"]).max(), np.abs(e.region(A.R_US).cpu().numpy() - ref["us"]).max(),
             e.traj_f(A.TF_COST)[0], ref["traj_f"][A.TF_COST][0], e.traj_f(A.TF_XREG)[0], ref["traj_f"][A.TF_XREG][0]))
sp = scenarios.solver_params(sc, maxiter=it)
e.set_candidate(None, None); e.solve(sp, poll_every=0); torch.cuda.synchronize()
k_warm = e.region(A.R_KFF).cpu().numpy().copy()       # the warm start BoxQP takes at iteration `it`
xreg = float(e.traj_f(A.TF_XREG)[0])
e.calc_diff(); torch.cuda.synchronize()
deriv = e.region(A.R_DERIV).cpu().numpy().copy()
us = e.region(A.R_US).cpu().numpy().copy()
e.backward_pass(sp); torch.cuda.synchronize()
Kg, kg, Qug = e.region(A.R_KGAIN).cpu().numpy(), e.region(A.R_KFF).cpu().numpy(), e.region(A.R_QU).cpu().numpy()
o = po.backward_pass(low, sp, deriv, np.zeros((low.T + 1, 1, low.nx)), us, xreg, 1, kff0=k_warm)
lb, ub = np.array([-100., -100., 0., 0.]), np.array([100., 100., 100., 100.])
print("one backward pass from identical inputs (xreg %.1e): fail gpu-status %d oracle %d" % (xreg, e.traj_i(A.TI_STATUS)[0], o["fail"][0]))
worst = 0.0
for t in range(low.T - 1, -1, -1):
    dK = np.abs(Kg[t, 0] - o["K"][t, 0]).max() / max(1e-300, np.abs(o["K"][t, 0]).max())
    dk = np.abs(kg[t, 0] - o["k"][t, 0]).max() / max(1e-300, np.abs(o["k"][t, 0]).max())
    un_g, un_o = us[t, 0] - kg[t, 0], us[t, 0] - o["k"][t, 0]
    cg = "".join("L" if un_g[i] <= lb[i] else ("U" if un_g[i] >= ub[i] else ".") for i in range(4))
    co = "".join("L" if un_o[i] <= lb[i] else ("U" if un_o[i] >= ub[i] else ".") for i in range(4))
    flag = " <-- clamped sets differ" if cg != co else ""
    if max(dK, dk) > 10 * worst or flag:
        print("knot %3d: rel |dK| %.2e rel |dk| %.2e, u - k clamped gpu %s oracle %s, Qu zeroed gpu %s oracle %s%s"
              % (t, dK, dk, cg, co, (Qug[t, 0] == 0).astype(int), (o["Qu"][t, 0] == 0).astype(int), flag))
        worst = max(worst, dK, dk)
