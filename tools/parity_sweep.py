"""Large parity sweep (not part of the test suite): full solves of seeded batches on the GPU against the CPU oracle on
the box's host cores.  Usage: parity_sweep.py [B]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
if os.environ.get("ASLR_LIB_OVERRIDE"):  # an experimental / earlier build of the library
    from aslr_to_amd import _abi as _A
    _A.lib_path = lambda: os.path.abspath(os.environ["ASLR_LIB_OVERRIDE"])
from aslr_to_amd import scenarios, _abi as A
from aslr_to_amd.engine import Engine
from oracle import pyoracle as po
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ONLY = sys.argv[2] if len(sys.argv) > 2 else None  # restrict to one scenario name
nth = min(16, len(os.sched_getaffinity(0)))
for name, solver, T in (("two_dof_vsa_boxddp", "SolverBoxDDP", 100), ("two_dof_sea", "SolverDDP", 100),
                        ("two_dof_sea", "SolverFDDP", 100), ("talos_arm_sea", "SolverFDDP", 40)):
    if ONLY and name != ONLY: continue
    Bn = B if "talos" not in name else max(8, B // 16)
    sc = scenarios.SCENARIOS[name](B=Bn, T=T, seed=3)
    low = scenarios.lower(sc)
    sp = scenarios.solver_params(sc, solver=solver)
    t0 = time.time(); ref = po.solve(low, sp, nthreads=nth); tc = time.time() - t0
    e = Engine(low); e.set_candidate(None, None)
    torch.cuda.synchronize(); t0 = time.time(); e.solve(sp, poll_every=4); torch.cuda.synchronize(); tg = time.time() - t0
    it_g, it_r = e.traj_i(A.TI_ITER).cpu().numpy(), ref["traj_i"][A.TI_ITER]
    st_g, st_r = e.traj_i(A.TI_STATUS).cpu().numpy(), ref["traj_i"][A.TI_STATUS]
    conv = (st_r & A.ST_CONVERGED) != 0
    X, U = e.region(A.R_XS).cpu().numpy(), e.region(A.R_US).cpu().numpy()
    dx = np.abs(X - ref["xs"]).max(axis=(0, 2)); du = np.abs(U - ref["us"]).max(axis=(0, 2))
    dc = np.abs(e.traj_f(A.TF_COST).cpu().numpy() - ref["traj_f"][A.TF_COST])
    print("%-20s %-12s B=%4d T=%3d: converged %4d / %4d (gpu %4d); same iteration count %4d / %4d; same status %4d; "
          "converged: max|dx| %.2e max|du| %.2e max|dcost| %.2e; within 1e-6: %d / %d; oracle %.1f s (%d threads), gpu %.2f s"
          % (name, solver, Bn, T, conv.sum(), Bn, ((st_g & A.ST_CONVERGED) != 0).sum(), (it_g == it_r).sum(), Bn,
             (st_g == st_r).sum(), dx[conv].max() if conv.any() else 0, du[conv].max() if conv.any() else 0,
             dc[conv].max() if conv.any() else 0, ((dx < 1e-6) & (du < 1e-6))[conv].sum(), conv.sum(), tc, nth, tg))
