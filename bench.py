#!/usr/bin/env python
"""Throughput benchmark of the hot path (the counterpart of the reference's
``benchmarks/throughput_benchmark.py``): metric = env-steps/s = vector_steps x agents / seconds
around the ``train``-equivalent with validation off (:222-249), benchmark-default hyper-parameters
(gamma 0.99, lr Exp(0.1 -> 1e-5, 0.995), epsilon Exp(1.0 -> 0.01, 0.995), :53-59).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]

One "step" = one vector step of the fused loop (select -> env.step -> learn for every agent).
N > 1: launched by ``torch.distributed.run``, one rank per GPU; every GPU owns its own agents and a
table replica and exchanges Q-deltas every 100 steps over RCCL (weak scaling: per-GPU work fixed).
Rank 0 prints ONE JSON line.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

import numpy as np  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md

WORKLOADS = {
    # BASELINE.json `metric`: "env-steps/sec at 128 agents, 1e6 x 16 Q-table"
    "headline": {"agents": 128, "states": 1_000_000, "actions": 16, "masked": False},
    "c2": {"agents": 128, "states": 10_000, "actions": 8, "masked": False},
    "c3": {"agents": 4096, "states": 1_000_000, "actions": 16, "masked": False},
    "c4shard": {"agents": 8192, "states": 10_000_000, "actions": 32, "masked": False},
    "c5": {"agents": 1024, "states": 1_000_000, "actions": 64, "masked": True},
    "wide": {"agents": 262_144, "states": 10_000_000, "actions": 16, "masked": False},
    # the environment of every number the reference publishes (benchmark_results/*.json)
    "tictactoe": {"agents": 128, "states": 19_683, "actions": 9, "masked": True, "env": "tictactoe"},
}
PUBLISHED_TICTACTOE_SINGLE_THREAD_128 = 22_300.0  # benchmark_results/single_thread_128_agents.json
SYNC_EVERY = 100  # BASELINE.json configs[3]: all-reduce of Q-deltas every 100 steps


def algorithmic_bytes_per_env_step(actions: int, masked: bool, esize: int = 4) -> int:
    """two row reads + RMW of Q[s,a] + agent I/O (+ packed masks)."""
    b = 2 * actions * esize + 2 * esize + 17
    if masked:
        b += 2 * ((actions + 7) // 8)
    return b


def cpu_baseline(wl, budget_s: float = 12.0):
    """The oracle's interpreted NumPy restatement of the reference's single_thread loop, timed on
    one host core on a bounded sample of the same workload (float64 table like the reference,
    pre-faulted, warm-up discarded)."""
    for var in ("OMP_NUM_THREADS", "MKL_NUM_THREADS", "OPENBLAS_NUM_THREADS"):
        os.environ.setdefault(var, "1")  # throughput_benchmark.py:16-18
    from oracle.envs import HashTabularEnv, TicTacToeVecEnv
    from oracle.qlearn_oracle import OracleQLearning, OracleRuntime, OracleSchedule

    n = wl["agents"]
    if wl.get("env") == "tictactoe":
        env = TicTacToeVecEnv(n, seed=1)
    else:
        env = HashTabularEnv(n, wl["states"], wl["actions"], seed=1, masked=wl["masked"])
    algo = OracleQLearning(wl["states"], wl["actions"], 0.99, seed=0)
    algo.q_table.fill(0.0)  # pre-fault
    rt = OracleRuntime(algo, OracleSchedule("exponential", 0.1, 1e-5, 0.995),
                       OracleSchedule("exponential", 1.0, 0.01, 0.995))
    states, _ = env.reset()
    acc, hist = np.zeros(n, dtype=np.float32), []
    warm = max(2, min(100, 20000 // n))
    for _ in range(warm):
        states, _ = rt.run_single_step(env, states, acc, hist)
    rates, t_start = [], time.perf_counter()
    block = max(5, min(1000, 100_000 // n))
    while len(rates) < 5 and (time.perf_counter() - t_start < budget_s or not rates):
        t0 = time.perf_counter()
        for _ in range(block):
            states, _ = rt.run_single_step(env, states, acc, hist)
        rates.append(block * n / (time.perf_counter() - t0))
    out = {
        "value": float(np.median(rates)), "unit": "env-steps/s", "cores": 1, "kind": "port",
        "sample": f"{len(rates)} x {block} vector steps x {n} agents after {warm} warm-up steps, "
                  "oracle/qlearn_oracle.py (interpreted NumPy restatement of single_thread, fp64 table)",
    }
    try:
        if wl.get("env") == "tictactoe":
            raise ImportError  # the C restatement covers the hash environment only
        from oracle import c_oracle

        out["compiled_c_value"] = c_oracle.time_rollout(wl, seconds=3.0)
        out["compiled_c_note"] = "oracle/qlearn_oracle.c, same loop compiled with gcc -O2, 1 core"
    except Exception:  # the C restatement is optional test infrastructure
        pass
    return out


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20000)
    ap.add_argument("--warmup", type=int, default=2000)
    ap.add_argument("--workload", default="headline", choices=sorted(WORKLOADS))
    ap.add_argument("--mode", default="iter", choices=["iter", "vec"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    # stdout carries the ONE JSON line and nothing else: libraries that print banners to file
    # descriptor 1 (RCCL does at communicator creation) are sent to stderr until the line is printed
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    wl = WORKLOADS[args.workload]
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    use_dist = "RANK" in os.environ  # launched by torch.distributed.run (also with a single rank)
    if use_dist:
        import torch
        import torch.distributed as dist

        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    n_gpus = world if world > 1 else 1
    if args.gpus != n_gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but {n_gpus} rank(s) were started; launch N > 1 with\n"
                         f"  python -m torch.distributed.run --nnodes=1 --nproc-per-node {args.gpus} --master-addr 127.0.0.1 "
                         f"--master-port P bench.py --gpus {args.gpus} ...\n")
        raise SystemExit(2)

    from dist_classicrl_amd.algorithms.base_algorithms.q_learning_optimal import OptimalQLearningBase
    from dist_classicrl_amd.algorithms.runtime.gpu_rollout_runtime import GpuRolloutQLearning
    from dist_classicrl_amd.environments import HashTabularEnv, TicTacToeEnv
    from dist_classicrl_amd.schedules import ExponentialSchedule

    n = wl["agents"]
    algo = OptimalQLearningBase(wl["states"], wl["actions"], 0.99, seed=0, dtype=np.float32,
                                device=local_rank)
    if wl.get("env") == "tictactoe":
        env = TicTacToeEnv(n, seed=1, agent_offset=rank * n)
    else:
        env = HashTabularEnv(n, wl["states"], wl["actions"], seed=1, masked=wl["masked"],
                             agent_offset=rank * n)
    rt = GpuRolloutQLearning(algo, ExponentialSchedule(0.1, 1e-5, 0.995),
                             ExponentialSchedule(1.0, 0.01, 0.995), learn_mode=args.mode)
    if n >= 16384:
        # hundreds of thousands of episodes end per call: take the returns as one float32 array
        # instead of a Python list with one object per episode (the values are the same)
        rt.history_type = "array"
    if use_dist:
        from dist_classicrl_amd.distributed.delta_sync import attach_engine

        rt.sync_every = SYNC_EVERY
        rt.delta_sync = attach_engine(algo, SYNC_EVERY, n)

    def sync_all():
        if use_dist:
            import torch

            dist.barrier()
            torch.cuda.synchronize()
        algo._lib.qe_synchronize(algo.handle)

    def run_steps(steps, state):
        """`run_steps`; the reference divides by the number of finished episodes (single_thread_runtime.py:67),
        so a very short call in which no episode ends raises ZeroDivisionError AFTER all its work is done."""
        try:
            return rt.run_steps(steps, env, state)[3]
        except ZeroDivisionError:
            return env.state_dict()

    sd = run_steps(max(1, args.warmup), None)  # untimed warm-up (also resets the env)
    sync_all()
    t0 = time.perf_counter()
    sd = run_steps(args.steps, sd)  # EXACTLY K timed vector steps
    sync_all()
    elapsed = time.perf_counter() - t0
    stats = dict(rt.last_stats)
    if use_dist:
        import torch

        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank != 0:
        dist.destroy_process_group()
        return

    env_steps = args.steps * n * n_gpus
    bpe = algorithmic_bytes_per_env_step(wl["actions"], wl["masked"])
    launches = max(1, stats["dominant_launches"])
    avg_launch_s = stats["dominant_ms"] / launches / 1e3
    units_per_launch = stats["dominant_env_steps"] / launches
    achieved = bpe * units_per_launch / avg_launch_s / 1e9 if avg_launch_s > 0 else 0.0
    persistent = stats["launches"] < args.steps
    traffic = None  # HBM bytes per launch from the PMC passes kept under profiles/ (same kernel, same workload)
    try:
        prof = json.loads((ROOT / "profiles" / "r01h_traffic.json").read_text()).get(args.workload)
        if prof and persistent and "persistent" in prof["kernel"]:
            traffic = prof["traffic_bytes_per_env_step"] * units_per_launch
    except (OSError, ValueError, KeyError):
        pass
    line = {
        "metric": "env-steps/sec at 128 agents, 1e6x16 Q-table; 1/2/4/8 GPU + HBM GB/s %peak",
        "value": env_steps / elapsed,
        "unit": "env-steps/s",
        "n_gpus": n_gpus,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": f"{args.workload}: {n} agents/GPU, {wl['states']} states x {wl['actions']} actions, "
                        f"fp32 Q-table, {'TicTacToeEnv' if wl.get('env') == 'tictactoe' else 'HashTabularEnv'}"
                        f"{' (masked)' if wl['masked'] else ''}, "
                        f"learn={args.mode}, benchmark-default schedules"
                        f"{', episode returns as array' if rt.history_type == 'array' else ''}",
            "agents_per_gpu": n, "states": wl["states"], "actions": wl["actions"],
            "sync_every": SYNC_EVERY if use_dist else None,
            "parallelism": f"agents sharded x{n_gpus}, table replicas + RCCL delta all-gather" if n_gpus > 1 else "1 GPU",
        },
        "roofline": {
            "bound": "hbm",
            "kernel": "k_rollout_persistent" if persistent else "k_step_fast",
            "achieved": achieved,
            "peak": HBM_PEAK_GBPS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBPS,
            "traffic": traffic,
            "alg_bytes_per_env_step": bpe,
            "units_per_launch": units_per_launch,
            "avg_launch_us": avg_launch_s * 1e6,
            "launches_sampled": stats["dominant_launches"],
        },
        "device_region_ms": stats["kernel_ms"],
        "kernel_launches": stats["launches"],
        "episodes": int(stats["episodes"]),
        "contested_agent_steps": stats["involved"],
    }
    if not args.no_cpu_baseline and n_gpus == 1:
        line["cpu_baseline"] = cpu_baseline(wl)
        line["speedup_vs_cpu_baseline"] = line["value"] / line["cpu_baseline"]["value"]
    if args.workload == "tictactoe" and n_gpus == 1:
        line["vs_reference_published_single_thread_128_agents_i7_11700K"] = (
            line["value"] / PUBLISHED_TICTACTOE_SINGLE_THREAD_128)
    if use_dist:
        line["delta_sync"] = {"syncs": rt.delta_sync.syncs, "bytes_received_per_gpu": rt.delta_sync.bytes_exchanged}
    sys.stdout.flush()
    os.dup2(json_fd, 1)
    print(json.dumps(line), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
