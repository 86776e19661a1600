"""Worker of tests/test_dist_gloo.py: one rank of a world_size-2 gloo group on CPU."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from aslr_to_amd import _abi, crocoddyl, dist, scenarios
    from oracle import pyoracle

    rank, world, _ = dist.init_from_env(backend="gloo")
    sc = scenarios.two_dof_sea(B=7, T=20, seed=0)
    problem = crocoddyl.ShootingProblem(sc["x0"], sc["running"], sc["terminal"], frame_refs=sc["frame_refs"],
                                        rank=rank, world_size=world)
    sp = scenarios.solver_params(sc)
    # the compute of this CPU-only test is the oracle (tests may use it); the code under test is the
    # sharding + the single SUM all-reduce of the stats vector
    r = pyoracle.solve(problem.lowered, sp)
    st = r["traj_i"][_abi.TI_STATUS]
    v = torch.zeros(len(dist.STAT_FIELDS), dtype=torch.float64)
    v[0] = float(r["traj_f"][_abi.TF_COST].sum())
    v[1] = float(r["traj_f"][_abi.TF_STOP].sum())
    v[2] = problem.batch
    v[3] = int(((st & _abi.ST_CONVERGED) != 0).sum())
    v[4] = int(((st & _abi.ST_REG_MAX) != 0).sum())
    v[5] = int(r["traj_i"][_abi.TI_ITER].sum())
    dist.barrier()
    stats = dist.all_reduce_stats(v)
    tmax = dist.max_over_ranks(float(rank + 1))
    out = dict(rank=rank, world=world, rows=problem.rows, stats=stats, tmax=tmax,
               xs_first=r["xs"][:, 0, :].tolist())
    with open(sys.argv[1] + ".%d" % rank, "w") as f:
        json.dump(out, f)
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
