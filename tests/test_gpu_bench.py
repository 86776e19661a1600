"""bench.py keeps its contract (one JSON line with the driver's keys, roofline and cpu_baseline objects)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_json_line_with_the_contract_keys():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "2",
                          "--batch-per-gpu", "256"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 2 and d["higher_is_better"] is True
    assert d["unit"] == "knot-steps/s" and d["dtype"] == "f64" and d["vs_baseline"] is None and d["scaling"] == "weak"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(r)
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    c = d["cpu_baseline"]
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(c) and c["kind"] == "port" and c["value"] > 0
    assert abs(d["value"] - 256 * 100 * 3 / (d["ms_per_step"] * 3e-3)) / d["value"] < 1e-6


def test_bench_gpus_2_launches_its_own_two_ranks():
    """`--gpus 2` without a torch.distributed.run wrapper: bench.py starts the two ranks itself.  On this one-GPU box
    the rehearsal switch puts both on cuda:0 with gloo for the two reductions (the throughput is then meaningless;
    the launch, the sharding, the barriers and the max-over-ranks timing are what is exercised)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env["ASLR_BENCH_REHEARSAL"] = "1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2",
                          "--batch-per-gpu", "256"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks"]["world_size"] == 2 and d["ranks"]["launcher"] == "bench.py"
    assert d["config"]["global_batch"] == 512 and d["solver_state"]["n"] == 512
    assert "cpu_baseline" not in d  # rank 0 at N = 1 only
    assert abs(d["value"] - 512 * 100 * 3 / (d["ms_per_step"] * 3e-3)) / d["value"] < 1e-6


@pytest.mark.parametrize("workload,batch,T,nx,nu,bytes_per_knot_step", [("c2", 128, 100, 8, 2, 3408), ("c5", 16, 150, 28, 7, 37088)])
def test_bench_other_baseline_configs_keep_the_contract(workload, batch, T, nx, nu, bytes_per_knot_step):
    """`--workload c2 / c5` (BASELINE.json configs[1] and configs[4] as one-GPU shards): the same one-line contract, the
    SURVEY 8(d) bytes of that configuration, a CPU baseline from the oracle on the same model."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", workload, "--steps", "3", "--warmup", "2",
                          "--batch-per-gpu", str(batch)], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["config"]["workload_id"] == workload and d["config"]["T"] == T and d["config"]["nx"] == nx and d["config"]["nu"] == nu
    assert d["unit"] == "knot-steps/s" and d["dtype"] == "f64" and d["n_gpus"] == 1 and d["vs_baseline"] is None
    assert d["roofline_iteration"]["algorithmic_bytes_per_knot_step"] == bytes_per_knot_step
    r = d["roofline"]
    assert r["bound"] == "hbm" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and "traffic" in r and r["traffic_note"]
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1
    assert abs(d["value"] - batch * T * 3 / (d["ms_per_step"] * 3e-3)) / d["value"] < 1e-6
    assert len(d["ranks"]["ms_per_step"]) == 1


def test_bench_rehearsal_with_four_self_launched_ranks_reports_every_rank():
    """The N > 1 path at a larger rank count than 2 (the GPU box allows at most 6 processes on its card, so 4 here; the
    8-rank launch itself is exercised on the CPU by tests/test_bench_launcher.py): four self-launched ranks, gloo for
    the reductions, one JSON line carrying every rank's own ms_per_step."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env["ASLR_BENCH_REHEARSAL"] = "1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "2", "--warmup", "1",
                          "--batch-per-gpu", "64"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 4 and d["ranks"]["world_size"] == 4 and d["config"]["global_batch"] == 256
    per = d["ranks"]["ms_per_step"]
    assert len(per) == 4 and all(v > 0 for v in per) and abs(max(per) - d["ms_per_step"]) < 1e-9
    assert d["solver_state"]["n"] == 256
