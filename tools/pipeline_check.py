"""The forward pass split in two launches (ASLR_PIPELINE=1: trial costs of the first half of the horizon under the rollout of the
second) must give the same bits as the plain sequence.  tools/pipeline_check.py [B] [ITERS]"""
import os, sys, subprocess, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import torch
    from aslr_to_amd import scenarios, _abi as A
    from aslr_to_amd.engine import Engine
    B, N, name, solver = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
    sc = scenarios.SCENARIOS[name](B=B, T=100, seed=0)
    e = Engine(scenarios.lower(sc))
    e.set_candidate(None, None)
    e.set_subshards(4 if B >= 256 else 1)
    sp = scenarios.solver_params(sc, solver=solver, fixed_iterations=1, maxiter=N)
    e.iterate_n(sp, True, N)
    torch.cuda.synchronize()
    h = hashlib.sha256()
    for r in (A.R_XS, A.R_US, A.R_TRAJ_F, A.R_TRAJ_I, A.R_KGAIN):
        h.update(e.region(r).cpu().numpy().tobytes())
    print(h.hexdigest())
    sys.exit(0)
B = sys.argv[1] if len(sys.argv) > 1 else "4096"
N = sys.argv[2] if len(sys.argv) > 2 else "25"
ENV = os.environ.get("CHECK_ENV", "ASLR_PIPELINE")  # the switch to compare 0 against 1
for name, solver in (("two_dof_vsa_boxddp", "SolverBoxDDP"), ("two_dof_sea", "SolverFDDP"), ("two_dof_sea", "SolverDDP")):
    out = []
    for pl in ("0", "1"):
        env = dict(os.environ, **{ENV: pl})
        out.append(subprocess.run([sys.executable, __file__, "--child", B, N, name, solver], env=env, capture_output=True, text=True).stdout.strip().split("\n")[-1])
    print(name, solver, "identical" if out[0] == out[1] and len(out[0]) == 64 else "DIFFERENT", out)
