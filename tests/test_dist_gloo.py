"""N > 1 path on CPU: world_size 2 over gloo.  The batch is sharded in contiguous blocks with no
data-path collective; one SUM all-reduce of the 8-double stats vector gives the global figures
(SURVEY.md 8(e)).  Results per trajectory must not depend on the sharding.
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np

from aslr_to_amd import _abi, scenarios

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_ranks_over_gloo_reduce_to_the_single_process_answer(oracle, tmp_path):
    port = _free_port()
    out = str(tmp_path / "rank")
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_dist_worker.py"), out], env=env))
    for p in procs:
        assert p.wait(timeout=300) == 0
    res = [json.load(open(out + ".%d" % r)) for r in range(2)]
    assert [tuple(r["rows"]) for r in res] == [(0, 4), (4, 7)]
    assert res[0]["stats"] == res[1]["stats"]          # every rank holds the reduced vector
    assert res[0]["tmax"] == res[1]["tmax"] == 2.0     # MAX all-reduce (the benchmark's max-over-ranks timing)
    sc = scenarios.two_dof_sea(B=7, T=20, seed=0)
    low = scenarios.lower(sc)
    full = oracle.solve(low, scenarios.solver_params(sc))
    st = full["traj_i"][_abi.TI_STATUS]
    s = res[0]["stats"]
    assert s["n"] == 7
    assert s["iters_sum"] == int(full["traj_i"][_abi.TI_ITER].sum())
    assert s["converged"] == int(((st & _abi.ST_CONVERGED) != 0).sum())
    assert abs(s["cost_sum"] - full["traj_f"][_abi.TF_COST].sum()) < 1e-9
    # trajectory 0 of each shard equals the same trajectory of the unsharded solve, bit for bit
    np.testing.assert_array_equal(np.array(res[0]["xs_first"]), full["xs"][:, 0, :])
    np.testing.assert_array_equal(np.array(res[1]["xs_first"]), full["xs"][:, 4, :])
