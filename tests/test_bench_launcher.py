"""bench.py --gpus N starts its own N ranks when it is not already one (no torch.distributed.run wrapper): the
environment plumbing of the launcher, and that a request it cannot honour fails loudly instead of measuring one GPU."""
import importlib.util
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench_module():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_rank_environment_is_what_torch_distributed_run_would_set():
    bench = _bench_module()
    base = {"PATH": "/usr/bin", "WORLD_SIZE": "7"}
    envs = [bench.rank_env(r, 4, 29123, base) for r in range(4)]
    for r, env in enumerate(envs):
        assert env["RANK"] == str(r) and env["LOCAL_RANK"] == str(r)
        assert env["WORLD_SIZE"] == "4" and env["LOCAL_WORLD_SIZE"] == "4"
        assert env["MASTER_ADDR"] == "127.0.0.1" and env["MASTER_PORT"] == "29123"
        assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"      # dmabuf IPC: RCCL needs it on this pool
        assert env["ASLR_BENCH_SELF_LAUNCHED"] == "1"
        assert env["PATH"] == "/usr/bin"
    assert base["WORLD_SIZE"] == "7"                         # the caller's mapping is not modified


def test_launcher_returns_the_first_failing_rank_code_and_prints_nothing(tmp_path):
    """Without GPUs every rank refuses to run: the launcher must exit non-zero and no JSON line may appear."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "ASLR_BENCH_REHEARSAL")}
    env["HIP_VISIBLE_DEVICES"] = ""  # also on a GPU box: no device for the children
    env["CUDA_VISIBLE_DEVICES"] = ""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                          "--batch-per-gpu", "8"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode != 0
    assert out.stdout.strip() == ""
    assert "--gpus 2 but only 0 GPU(s) visible" in out.stderr


def test_world_size_mismatch_is_rejected():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    for gpus in ("1", "4"):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", gpus, "--steps", "1"],
                             capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
        assert out.returncode != 0 and out.stdout.strip() == ""
        assert "--gpus %s but WORLD_SIZE=2" % gpus in out.stderr


def test_eight_rank_environments_and_a_missing_rank_times_out_instead_of_hanging():
    """The day-one 8-GPU launch: eight rank environments with one rendezvous; and a rank whose peers never arrive leaves
    with a non-zero exit code after the bounded rendezvous timeout (ASLR_BENCH_INIT_TIMEOUT) -- it does not wait for ever."""
    bench = _bench_module()
    envs = [bench.rank_env(r, 8, 29555, {"PATH": "/usr/bin"}) for r in range(8)]
    assert sorted(int(e["RANK"]) for e in envs) == list(range(8)) and {e["WORLD_SIZE"] for e in envs} == {"8"}
    assert len({e["MASTER_PORT"] for e in envs}) == 1
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, ASLR_BENCH_REHEARSAL="1", ASLR_BENCH_INIT_TIMEOUT="5")
    env.update(bench.rank_env(0, 2, port, {}))   # rank 0 of 2; rank 1 is never started
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                          "--batch-per-gpu", "8"], capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert out.returncode != 0 and out.stdout.strip() == ""
    assert "rendezvous failed" in out.stderr or "needs a ROCm GPU" in out.stderr
