"""Multi-GPU: one process per GPU, the batch sharded in contiguous blocks, NO data-path collective.

Shooting problems are independent (one `ShootingProblem` per solve in the reference,
examples/two_dof_vsa_boxddp.py:66), so the only exchange is ONE all-reduce (SUM) of an 8-double
vector per solve (or per k iterations) for global termination and reporting.  With backend "nccl"
this is RCCL over xGMI; the payload is 64 bytes, so it is pure latency and is kept off the
per-iteration path.  The same code runs on CPU tensors over gloo (tests/test_dist_gloo.py).
"""
import os

from . import _abi
from .lowering import shard_rows  # noqa: F401  (re-exported)

STAT_FIELDS = ("cost_sum", "stop_sum", "n", "converged", "failed", "iters_sum", "trials_sum", "active")


def init_from_env(backend=None, timeout_s=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* when WORLD_SIZE > 1 (`timeout_s`: bound on the
    rendezvous and on every collective, so that a missing rank raises instead of hanging the others).
    Returns (rank, world_size, local_rank)."""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        kw = {}
        if timeout_s is not None:
            import datetime
            kw["timeout"] = datetime.timedelta(seconds=float(timeout_s))
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, world, local


def local_stats(engine):
    """8-double vector of this shard's per-trajectory state, on the engine's device."""
    import torch
    st = engine.traj_i(_abi.TI_STATUS)
    v = torch.zeros(len(STAT_FIELDS), dtype=torch.float64, device=engine.device)
    v[0] = engine.traj_f(_abi.TF_COST).sum()
    v[1] = engine.traj_f(_abi.TF_STOP).sum()
    v[2] = float(engine.B)
    v[3] = ((st & _abi.ST_CONVERGED) != 0).sum()
    v[4] = ((st & _abi.ST_REG_MAX) != 0).sum()
    v[5] = engine.traj_i(_abi.TI_ITER).sum()
    v[6] = engine.traj_i(_abi.TI_NTRIALS).sum()
    v[7] = (engine.traj_i(_abi.TI_DONE) == 0).sum()
    return v


def all_reduce_stats(vec, group=None):
    """SUM-all-reduce the stats vector (no-op without an initialised process group) -> dict."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        if dist.get_backend(group) == "gloo":
            vec = vec.cpu()  # gloo reduces host tensors (CPU tests, single-GPU rehearsals of the N > 1 path)
        dist.all_reduce(vec, op=dist.ReduceOp.SUM, group=group)
    h = vec.detach().cpu().tolist()
    out = dict(zip(STAT_FIELDS, h))
    for k in ("n", "converged", "failed", "iters_sum", "trials_sum", "active"):
        out[k] = int(round(out[k]))
    return out


def max_over_ranks(value, device=None, group=None):
    """MAX-all-reduce of one float (the benchmark's max-over-ranks timing)."""
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        if dist.get_backend(group) == "gloo":
            t = t.cpu()
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())


def gather_floats(value, device=None, group=None):
    """Every rank's float, in rank order, on every rank (all_gather of one double)."""
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        if dist.get_backend(group) == "gloo":
            t = t.cpu()
        out = [torch.zeros_like(t) for _ in range(dist.get_world_size(group))]
        dist.all_gather(out, t, group=group)
        return [float(o.item()) for o in out]
    return [float(value)]


def barrier(group=None):
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.barrier(group=group)
