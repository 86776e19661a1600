#!/usr/bin/env python3
"""Benchmark of the batched DDP hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c3|c2|c5]

With N > 1 and no RANK / WORLD_SIZE in the environment (i.e. not under torch.distributed.run) the script starts its
own N ranks, one process per GPU (launch_ranks); under torch.distributed.run it is one of the ranks.

Workload (BASELINE.json metric / configs[2], SURVEY.md 8(d) "C3"; C4 for N = 8): batched BoxDDP on the
2-DoF VSA arm, T = 100, 4096 trajectories PER GPU (weak scaling; rank r owns rows
[4096 r, 4096 (r+1)) of the seed-0 batch of 4096 N), synthetic seeded inputs, cold start,
fixed-iteration mode (convergence exit disabled, as a throughput measurement needs).

A "step" is ONE lock-step DDP iteration over the shard: calc/calcDiff sweep + backward pass (Riccati +
BoxQP) + forward pass with the full 10-alpha line search.  W warm-up iterations are the first W
iterations of the solve; the next K are timed between barrier + torch.cuda.synchronize() pairs; the
maximum over ranks is taken.  value = knot-steps/s = (sum over ranks of B) * T * K / time.  Inputs are
resident in HBM before the timed region.

Extra objects on the JSON line:
  roofline      the dominant kernel of the iteration: ALGORITHMIC bytes per launch (SURVEY.md 8(d)'s
                per-knot-step figure for that phase x the B*T knot-steps one launch processes)
                / its mean launch duration measured here with HIP events on the launch stream.
  cpu_baseline  the CPU oracle (oracle/, a port -- Crocoddyl itself cannot be installed here) solving a
                bounded sample of the same batch on the host cores, rank 0 at N = 1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)

# --workload: BASELINE.json configs as one-GPU shards (weak scaling: the same shard on every GPU).  c3 is the headline
# (configs[2]; configs[3] = 8 x c3); c2 = configs[1]; c5 = configs[4] (4096 trajectories over 8 GPUs = 512 per GPU).
# `kernels`: the instantiations one iteration launches in steady state, per phase (calc, backward, forward), by their
# demangled names -- what the committed PMC summary is searched for (exact match, see pmc_traffic).
WORKLOADS = {
    "c3": dict(scenario="two_dof_vsa_boxddp", solver="SolverBoxDDP", T=100, batch=4096,
               metric="knot-steps/s (batched BoxDDP, 2-DoF VSA, T=100, 4096 trajectories per GPU)",
               what="two_dof_vsa_boxddp (examples/two_dof_vsa_boxddp.py, T=100): SolverBoxDDP, cold start, "
                    "fixed-iteration mode, full 10-alpha line search every iteration",
               kernels=["calc_kernel<2, 1, true, true, false, true>", "backward_kernel<8, 4, 2, 0, true, true>",
                        "rollout_kernel<2, 1, true, false>"],
               cpu=dict(per_thread=64, single=32, maxiter=None)),
    "c2": dict(scenario="two_dof_sea", solver="SolverDDP", T=100, batch=1024,
               metric="knot-steps/s (batched DDP, 2-DoF SEA, T=100, 1024 trajectories per GPU)",
               what="two_dof_sea (examples/two_dof_sea.py, T=100): SolverDDP (north_star; the script itself uses FDDP), "
                    "cold start, fixed-iteration mode, full 10-alpha line search every iteration",
               kernels=["calc_kernel<2, 0, true, true, false, true>", "backward_kernel<8, 2, 4, 0, false, true>",
                        "rollout_kernel<2, 0, true, false>"],
               cpu=dict(per_thread=256, single=128, maxiter=None)),
    "c5": dict(scenario="talos_arm_sea", solver="SolverDDP", T=150, batch=512,
               metric="knot-steps/s (batched DDP, 7-DoF arm + SEA, T=150, 512 trajectories per GPU = 4096 over 8 GPUs)",
               what="talos_arm_sea (7-joint chain with SEA actuation, cost stack of examples/two_dof_sea.py, T=150): SolverDDP, "
                    "cold start, fixed-iteration mode, full 10-alpha line search every iteration",
               kernels=["dyn_team_kernel<7, 1>", "backward_blk_kernel<28, 7, true, true, false>", "rollout_team_kernel<7, false>"],
               cpu=dict(per_thread=8, single=8, maxiter=30)),
}


def algorithmic_bytes(nx, nu):
    """bytes per knot-step of each phase (one accepted line-search trial)."""
    n, m = nx, nu
    blocks = 2 * n * n + 2 * n * m + m * m + n + m
    p1 = 8 * ((n + m) + (n + 1) + blocks)          # calc + calcDiff: reads x,u; writes xnext,cost,blocks
    p2 = 8 * (blocks + (m * n + 2 * m))            # backward: reads blocks; writes K,k,Qu
    p3 = 8 * ((n + 2 * m + m * n) + (n + m + 1))   # forward, per trial
    return p1, p2, p3


def pmc_traffic(kernel_name, workload):
    """HBM bytes per launch of a kernel from the committed rocprofv3 PMC summary (separate --pmc passes of this same
    command with --subshards 1: profiles/rNN/bench_pmc[_WORKLOAD].csv, newest round first): WRITE_SIZE + 2 x FETCH_SIZE,
    both in KiB -- the gfx950 correction of /opt/skills/guides/MI355X_MICROARCH.md (FETCH_SIZE counts wide coalesced
    reads at half).  The kernel must be in the summary under exactly this (demangled) instantiation name: a summary
    taken before the kernel changed its template signature does not describe it.
    -> (bytes or None, note): the note names the file and the commit the summary was taken at (its .meta.json)."""
    import csv
    suffix = "" if workload == "c3" else "_" + workload
    for rnd in ("r03", "r02", "r01"):
        path = os.path.join(ROOT, "profiles", rnd, "bench_pmc%s.csv" % suffix)
        if not os.path.exists(path):
            continue
        meta = {}
        try:
            meta = json.load(open(path[:-4] + ".meta.json"))
        except Exception:
            pass
        stamp = "profiles/%s/bench_pmc%s.csv taken at commit %s" % (rnd, suffix, meta.get("commit", "(unrecorded)"))
        fetch = write = None
        for r in csv.DictReader(open(path)):
            if r["kernel"] == "void aslr::" + kernel_name:
                if r["counter"] == "FETCH_SIZE":
                    fetch = float(r["median_per_dispatch"])
                if r["counter"] == "WRITE_SIZE":
                    write = float(r["median_per_dispatch"])
        if fetch is None or write is None:
            return None, "no FETCH_SIZE / WRITE_SIZE rows for `%s` in %s: the summary does not describe this kernel" % (kernel_name, stamp)
        return (2.0 * fetch + write) * 1024.0, "bytes per launch (not live): " + stamp
    return None, "no PMC summary committed for workload %s" % workload


def cpu_baseline(w, nthreads):
    """Oracle (CPU port) solving the first trajectories of the same batch in converge mode (a bounded sample)."""
    from aslr_to_amd import _abi, scenarios
    from oracle import pyoracle
    sc_fn, T, cfg = scenarios.SCENARIOS[w["scenario"]], w["T"], w["cpu"]
    kw = {} if cfg["maxiter"] is None else {"maxiter": cfg["maxiter"]}
    nsample = cfg["per_thread"] * nthreads   # c3: ~8 s of wall time on 16 host cores (~2 minutes of CPU work)
    sc = sc_fn(B=nsample, T=T, seed=0)
    low = scenarios.lower(sc)
    sp = scenarios.solver_params(sc, solver=w["solver"], **kw)
    t0 = time.perf_counter()
    r = pyoracle.solve(low, sp, nthreads=nthreads)
    dt = time.perf_counter() - t0
    iters = int(r["traj_i"][_abi.TI_ITER].sum())
    # mode (i) of SURVEY.md 8(d): one thread, the reference's forced nthreads = 1
    sc1 = sc_fn(B=cfg["single"], T=T, seed=0)
    low1 = scenarios.lower(sc1)
    t0 = time.perf_counter()
    r1 = pyoracle.solve(low1, scenarios.solver_params(sc1, solver=w["solver"], **kw), nthreads=1)
    dt1 = time.perf_counter() - t0
    single = int(r1["traj_i"][_abi.TI_ITER].sum()) * T / dt1
    return {"value": iters * T / dt, "unit": "knot-steps/s", "cores": nthreads, "kind": "port",
            "single_thread_value": single,
            "sample": "first %d trajectories of the seed-0 batch, full %s solves (th_stop 1e-7, maxiter %s), "
                      "%d DDP iterations in %.2f s, OpenMP over trajectories"
                      % (nsample, w["solver"], sp.maxiter, iters, dt),
            "single_thread_note": "the reference forces nthreads = 1 (examples/double_pendulum.py:54)"}


def rank_env(rank, world, port, base=None):
    """Environment of one child rank (the variables torch.distributed.run would set)."""
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0",
               ASLR_BENCH_SELF_LAUNCHED="1")
    return env


def launch_ranks(n, argv):
    """`python bench.py --gpus N` (N > 1) without a torch.distributed.run wrapper: start the N ranks from here, one
    process per GPU.  This parent never touches torch or the GPU (no exec from a process that initialised HIP); rank
    0 prints the JSON line on the inherited stdout, the other ranks' stdout goes to stderr.  Returns the exit code:
    non-zero as soon as any rank fails (the others are then terminated)."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv),
                                      env=rank_env(r, n, port), stdout=None if r == 0 else sys.stderr))
    rc = 0
    pending = list(procs)
    while pending:
        for p in list(pending):
            try:
                code = p.wait(timeout=0.2)
            except subprocess.TimeoutExpired:
                continue
            pending.remove(p)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                for q in pending:  # a rank died: the others would hang in the next collective
                    q.terminate()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="c3",
                    help="c3: 2-DoF VSA BoxDDP, 4096 per GPU (headline); c2: 2-DoF SEA DDP, 1024; c5: 7-DoF SEA DDP, T=150, 512 per GPU")
    ap.add_argument("--batch-per-gpu", type=int, default=0, help="trajectories per GPU (default: the workload's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--subshards", type=int, default=4,
                    help="iterate each GPU's shard as this many sub-shards on internal streams (1 = off)")
    ap.add_argument("--converge-mode", action="store_true",
                    help="also time the example's own solve (th_stop 1e-7, maxiter 400) after the measured region")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # read by the HIP runtime at initialisation (sub-shard streams)
    # stdout carries the ONE JSON line and nothing else: whatever the libraries below print there (gloo's connection
    # notes, ROCm warnings) goes to stderr instead; the line is written to the saved descriptor at the end
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    from aslr_to_amd import _abi, dist, scenarios
    from aslr_to_amd.crocoddyl import ShootingProblem

    # ASLR_BENCH_REHEARSAL=1: every rank on cuda:0 with gloo for the two reductions -- rehearses the N > 1 code
    # path on a one-GPU box (its throughput figure is meaningless: the ranks share the GPU)
    rehearsal = os.environ.get("ASLR_BENCH_REHEARSAL") == "1"
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if world_env != args.gpus:  # never report n_gpus different from --gpus
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world_env))
    if not rehearsal and torch.cuda.device_count() < args.gpus:
        raise SystemExit("bench.py: --gpus %d but only %d GPU(s) visible" % (args.gpus, torch.cuda.device_count()))
    try:
        rank, world, local = dist.init_from_env("gloo" if rehearsal else None, timeout_s=float(os.environ.get("ASLR_BENCH_INIT_TIMEOUT", "180")))
    except Exception as exc:  # a rank that cannot join must not leave the others waiting for ever
        raise SystemExit("bench.py: torch.distributed rendezvous failed (rank %s of %s): %s" % (os.environ.get("RANK"), world_env, exc))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU: the hot path has no CPU fallback")
    torch.cuda.set_device(local if world > 1 and not rehearsal else 0)
    ranks_seen = torch.distributed.get_world_size() if world > 1 else 1
    backend = torch.distributed.get_backend() if world > 1 else None
    dev = torch.device("cuda", torch.cuda.current_device())

    w = WORKLOADS[args.workload]
    T = w["T"]
    Bg = args.batch_per_gpu or w["batch"]
    sc = scenarios.SCENARIOS[w["scenario"]](B=Bg * world, T=T, seed=0)
    problem = ShootingProblem(sc["x0"], sc["running"], sc["terminal"], frame_refs=sc["frame_refs"],
                              rank=rank, world_size=world, device=dev)
    e = problem.engine
    sp = scenarios.solver_params(sc, solver=w["solver"], fixed_iterations=1, maxiter=args.warmup + args.steps)
    e.set_candidate(None, None)  # cold start, as examples/two_dof_vsa_boxddp.py:81
    # The shard is iterated as sub-shards on internal streams (same results bit for bit: trajectories are independent;
    # the one-wave-per-SIMD sweeps of a sub-shard run under the streaming kernels of the others).  The K steps are
    # enqueued by one aslr_iterate_n call, which forks from / joins the caller's stream around them.
    e.set_subshards(args.subshards)

    if args.warmup > 0:
        e.iterate_n(sp, True, args.warmup)
    dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    e.iterate_n(sp, args.warmup == 0, args.steps)
    torch.cuda.synchronize(dev)
    dist.barrier()
    mine = time.perf_counter() - t0
    elapsed = dist.max_over_ranks(mine, dev)
    per_rank_ms = [v / args.steps * 1e3 for v in dist.gather_floats(mine, dev)]  # (a slow GPU shows in the line)

    # per-kernel durations: HIP events on the launch stream around each phase, with the whole shard as ONE launch per
    # kernel (aslr_iterate_timed does not use the sub-shard streams), a few more iterations of the same solve
    nprobe = 10
    acc = [0.0, 0.0, 0.0]
    for _ in range(nprobe):
        ms = e.iterate_timed(sp)
        acc = [a + m for a, m in zip(acc, ms)]
    k_ms = [a / nprobe for a in acc]
    e.finalize()
    torch.cuda.synchronize(dev)
    stats = dist.all_reduce_stats(dist.local_stats(e))  # the ONE collective of a solve (RCCL over xGMI)

    # converge mode (--converge-mode): the example's own solve -- th_stop 1e-7, maxiter 400, exit when
    # every trajectory of the shard has stopped -- timed once, outside the figure of merit above (off by default so
    # that a rocprofv3 run of the default command averages the fixed-iteration launches only)
    converge = None
    if args.converge_mode and args.workload != "c3":
        raise SystemExit("bench.py: --converge-mode is defined for the headline workload (c3) only")
    if args.converge_mode:
        spc = scenarios.solver_params(sc)
        e.set_candidate(None, None)
        torch.cuda.synchronize(dev)
        tc0 = time.perf_counter()
        batch_iters = e.solve(spc, poll_every=4)
        torch.cuda.synchronize(dev)
        conv_wall = dist.max_over_ranks(time.perf_counter() - tc0, dev)
        cstats = dist.all_reduce_stats(dist.local_stats(e))
        converge = {"wall_s": conv_wall, "lock_step_iterations": int(batch_iters), "converged": cstats["converged"],
                    "trajectory_iterations": cstats["iters_sum"],
                    "useful_knot_steps_per_s": cstats["iters_sum"] * T / conv_wall,
                    "note": "th_stop 1e-7, maxiter 400, all trajectories iterate in lock-step until the last one "
                            "stops; useful = iterations each trajectory needed"}
        # the same solves as a POOL: 4x as many problems streamed through the same slots (aslr_solve_pool: a slot whose
        # problem has stopped is flushed and refilled on the device), so no slot waits for the batch's stragglers
        npool = 4 * Bg
        scp = scenarios.two_dof_vsa_boxddp(B=npool * world, T=T, seed=0)
        lo = rank * npool
        torch.cuda.synchronize(dev)
        tp0 = time.perf_counter()
        r = e.solve_pool(scp["x0"][lo:lo + npool], scp["frame_refs"][lo:lo + npool], spc, refill_every=4, poll_every=16)
        torch.cuda.synchronize(dev)
        pool_wall = dist.max_over_ranks(time.perf_counter() - tp0, dev)
        pool_iters = dist.max_over_ranks(float(r["iters"].sum().item()), dev) if world == 1 else float(r["iters"].sum().item())
        converge["pool"] = {"problems_per_gpu": npool, "slots_per_gpu": Bg, "wall_s": pool_wall,
                            "lock_step_iterations": int(r["batch_iters"]),
                            "converged": int(((r["status"] & _abi.ST_CONVERGED) != 0).sum().item()),
                            "trajectory_iterations": pool_iters,
                            "useful_knot_steps_per_s": pool_iters * T * world / pool_wall,
                            "note": "rank-local counts; H2D of the pool's x0 / targets (2.6 MB per GPU) is inside wall_s"}

    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    if rank != 0:
        return
    knot_steps = Bg * world * T * args.steps
    value = knot_steps / elapsed
    p1, p2, p3 = algorithmic_bytes(e.nx, e.nu)
    names = ["calc_kernel", "backward_kernel", "forward (rollout + trial_cost + sum_cost + select kernels)"]
    if args.workload == "c5":
        names = ["calcDiff (dyn_team_kernel x 2 + calc_kernel)", "backward_blk_kernel",
                 "forward (rollout_team + trial_cost + sum_cost + select kernels)"]
    # the sequential line search of the algorithm needs (accepted index + 1) trials; the kernel evaluates
    # all 10 step lengths at once, but only the required ones count as algorithmic traffic
    trials = stats["trials_sum"] / max(stats["iters_sum"], 1)
    phase_bytes = [p1, p2, p3 * trials]
    dom = max(range(3), key=lambda i: k_ms[i])
    launch_bytes = phase_bytes[dom] * Bg * T
    achieved = launch_bytes / (k_ms[dom] * 1e-3) / 1e9
    traffic, traffic_note = pmc_traffic(w["kernels"][dom], args.workload)
    out = {
        "metric": w["metric"] if Bg == w["batch"] else w["metric"].replace("%d trajectories per GPU" % w["batch"], "%d trajectories per GPU" % Bg),
        "value": value, "unit": "knot-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic" + (" (REHEARSAL: ranks share one GPU)" if rehearsal else ""),
        "config": {"workload": w["what"], "workload_id": args.workload,
                   "batch_per_gpu": Bg, "global_batch": Bg * world, "T": T, "nx": e.nx, "nu": e.nu,
                   "subshards_per_gpu": args.subshards,
                   "sharding": "contiguous batch blocks, no data-path collective"},
        "ranks": {"world_size": ranks_seen, "backend": backend, "ms_per_step": per_rank_ms,
                  "launcher": "bench.py" if os.environ.get("ASLR_BENCH_SELF_LAUNCHED") == "1" else
                              ("torch.distributed.run" if world > 1 else "single process"),
                  "note": "world size as torch.distributed reports it after init (nccl = RCCL over xGMI)"},
        "ddp_iterations_per_s": args.steps / elapsed,
        "trajectory_iterations_per_s": Bg * world * args.steps / elapsed,
        "line_search_trials_per_iteration": trials,
        "line_search_rollouts_per_s": 10 * Bg * world * args.steps / elapsed,   # every step length is rolled out
        "converge_mode": converge,   # --converge-mode
        "kernel_ms": dict(zip(names, k_ms)),
        "kernel_ms_note": "whole-shard launches on one stream (sub-shards off), HIP events; their sum is the "
                          "iteration without overlap, ms_per_step the iteration with the sub-shard streams",
        "roofline": {"bound": "hbm", "kernel": names[dom], "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_note": traffic_note,
                     "kernel_instantiation": w["kernels"][dom],
                     "algorithmic_bytes_per_launch": launch_bytes,
                     "algorithmic_bytes_per_knot_step": phase_bytes[dom],
                     "knot_steps_per_launch": Bg * T},
        "roofline_iteration": {"algorithmic_bytes_per_knot_step": p1 + p2 + p3,
                               "achieved": (p1 + p2 + p3) * value / world / 1e9, "peak": HBM_PEAK_GBS,
                               "unit": "GB/s per GPU", "frac": (p1 + p2 + p3) * value / world / 1e9 / HBM_PEAK_GBS,
                               "note": "SURVEY.md 8(d) whole-iteration figure (%d B per knot-step at nx=%d, nu=%d)" % (p1 + p2 + p3, e.nx, e.nu)},
        "solver_state": stats,
    }
    if world == 1 and not args.no_cpu_baseline:
        try:
            nthreads = len(os.sched_getaffinity(0))
        except AttributeError:
            nthreads = os.cpu_count() or 1
        nthreads = min(nthreads, 16)  # the CPU share of a one-GPU box
        out["cpu_baseline"] = cpu_baseline(w, nthreads)
    sys.stdout.flush()
    os.write(json_fd, (json.dumps(out) + "\n").encode())


if __name__ == "__main__":
    main()
