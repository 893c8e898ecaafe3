#!/usr/bin/env python
"""Throughput benchmark of the hot path (the counterpart of the reference's
``benchmarks/throughput_benchmark.py``): metric = env-steps/s = vector_steps x agents / seconds
around the ``train``-equivalent with validation off (:222-249), benchmark-default hyper-parameters
(gamma 0.99, lr Exp(0.1 -> 1e-5, 0.995), epsilon Exp(1.0 -> 0.01, 0.995), :53-59).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME] [--agents A]

One "step" = one vector step of the fused loop (select -> env.step -> learn for every agent).
N > 1: one rank per GPU; every GPU owns its own agents and a table replica and exchanges Q-deltas every
100 steps over RCCL (weak scaling: per-GPU work fixed).  The ranks are either started by the caller
(``python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N``: RANK / WORLD_SIZE in the
environment) or, when bench.py is run plainly with ``--gpus N``, by bench.py itself as child processes
(like the reference's harness starts its own ranks, throughput_benchmark.py:326-366).  Rank 0 prints ONE
JSON line.
"""

from __future__ import annotations

import argparse
import json
import os
import platform
import socket
import subprocess
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
ROOFLINE_SAMPLES = 8  # further launches of the same K steps, bracketed by HIP events


def algorithmic_bytes_per_env_step(actions: int, masked: bool, esize: int = 4) -> int:
    """two row reads + RMW of Q[s,a] + agent I/O (+ packed masks)."""
    b = 2 * actions * esize + 2 * esize + 17
    if masked:
        b += 2 * ((actions + 7) // 8)
    return b


def cpu_description():
    model = ""
    try:
        for line in Path("/proc/cpuinfo").read_text().splitlines():
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"model": model or platform.processor(), "logical_cores": os.cpu_count()}


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
        "cpu": cpu_description(),
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


def launch_ranks(args) -> int:
    """`python bench.py --gpus N` without a launcher: start N ranks as child processes (the parent never
    touches a GPU) and pass their one JSON line through."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve()), *sys.argv[1:]]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env, check=False).returncode


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20000)
    ap.add_argument("--warmup", type=int, default=2000)
    ap.add_argument("--workload", default="headline", choices=sorted(WORKLOADS))
    ap.add_argument("--agents", type=int, default=None, help="agents per GPU (default: the workload's)")
    ap.add_argument("--states", type=int, default=None, help="table rows (default: the workload's; analysis runs only)")
    ap.add_argument("--mode", default="iter", choices=["iter", "vec"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--turn-forward", type=int, default=1, choices=[0, 1],
                    help="turnstile path (513 .. ~60 000 agents): 0 = no value forwarding in the progress words "
                         "(measurement switch, results are identical)")
    ap.add_argument("--turn-poll", type=int, default=0, choices=[0, 1],
                    help="turnstile path: 1 = progress words polled with sc1 loads, 0 = with returning atomics (measurement switch)")
    ap.add_argument("--stamp-hash-bits", type=int, default=0,
                    help="step-wise / wide paths: log2 of the hashed touch-counter slots (0 = automatic, 1 = one slot per "
                         "row; measurement switch, results are identical)")
    ap.add_argument("--lane-ordered-path", type=int, default=0, choices=[0, 1, 2, 3],
                    help="persistent path, up to 128 agents: 0 = automatic, 1 = the dataflow kernel, 2 = the build with the "
                         "general ordered path, 3 = the sparse build (measurement switch, results are identical)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend of the ranks; gloo + fewer GPUs than ranks (ranks then share GPUs) is a "
                         "rehearsal of the N > 1 path on a small box, not a measurement")
    args = ap.parse_args()

    launched = "RANK" in os.environ  # started by torch.distributed.run (also with a single rank)
    if args.gpus > 1 and not launched:
        raise SystemExit(launch_ranks(args))

    # stdout carries the ONE JSON line and nothing else: libraries that print banners to file
    # descriptor 1 (RCCL does at communicator creation) are sent to stderr until the line is printed
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    wl = dict(WORKLOADS[args.workload])
    if args.agents:
        wl["agents"] = args.agents
    if args.states:
        wl["states"] = args.states
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    use_dist = launched
    n_gpus = world if world > 1 else 1
    if args.gpus != n_gpus:  # an inconsistent environment: refuse rather than report a wrong n_gpus
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but WORLD_SIZE says {n_gpus} rank(s) were started\n")
        raise SystemExit(2)
    if use_dist:
        import torch
        import torch.distributed as dist

        if args.backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            local_rank %= max(1, torch.cuda.device_count())
            torch.cuda.set_device(local_rank)
            dist.init_process_group("gloo")

    from dist_classicrl_amd import _lib
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
    if not args.turn_forward:
        algo.set_engine_option(_lib.OPT_TURN_FORWARD, 0)
    if args.turn_poll:
        algo.set_engine_option(_lib.OPT_TURN_POLL, 1)
    if args.stamp_hash_bits:
        algo.set_engine_option(_lib.OPT_STAMP_HASH_BITS, args.stamp_hash_bits)
    if args.lane_ordered_path:
        algo.set_engine_option(_lib.OPT_LANE_ORDERED_PATH, args.lane_ordered_path)
    if n >= 16384:
        # hundreds of thousands of episodes end per call: take the returns as one float32 array
        # instead of a Python list with one object per episode (the values are the same)
        rt.history_type = "array"
    if use_dist:
        from dist_classicrl_amd.distributed.delta_sync import attach_engine

        rt.sync_every = SYNC_EVERY
        rt.delta_sync = attach_engine(algo, SYNC_EVERY, n)

    def device_sync():
        if use_dist:
            import torch

            torch.cuda.synchronize()
        algo._lib.qe_synchronize(algo.handle)

    def sync_all():
        device_sync()
        if use_dist:
            dist.barrier()
            device_sync()

    def run_steps(steps, state):
        """`run_steps`; the reference divides by the number of finished episodes (single_thread_runtime.py:67),
        so a very short call in which no episode ends raises ZeroDivisionError AFTER all its work is done."""
        try:
            return rt.run_steps(steps, env, state)[3]
        except ZeroDivisionError:
            return env.state_dict()

    # Untimed warm-up: W vector steps, taken in a few calls so that every per-call path of the timed call
    # (engine slots, page-locked result block, the Python around them) has run before the clock starts.
    # The timed launch carries no HIP events (the engine clocks it in-kernel): an event pair costs ~7 us
    # of a 20-step call.
    algo.set_engine_option(_lib.OPT_EVENT_TIMING, 0)
    w = max(1, args.warmup)
    pieces = [w // 4] * 3 + [w - 3 * (w // 4)] if w >= 8 else [1] * w
    sd = None
    for k in pieces:
        sd = run_steps(k, sd)
    sync_all()
    t0 = time.perf_counter()
    sd = run_steps(args.steps, sd)  # EXACTLY K timed vector steps
    t_call = time.perf_counter() - t0
    # the region closes as it opened, with the device drained and a barrier: every rank stops its clock when ITS
    # K steps are complete on the device, the slowest rank's time is the job's (MAX over ranks below) -- the
    # barrier's own latency (tens of microseconds, a large fraction of a 20-step call) is not step time
    device_sync()
    elapsed = time.perf_counter() - t0
    sync_all()
    stats = dict(rt.last_stats)
    if use_dist:
        import torch

        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # Roofline samples: further launches of the same K steps (the training simply continues), each
    # bracketed by HIP events on the engine's stream.
    algo.set_engine_option(_lib.OPT_EVENT_TIMING, 1)
    persistent = stats["launches"] < args.steps or args.steps == 1 and n <= 512 and wl["actions"] <= 64
    samples = []
    for _ in range(ROOFLINE_SAMPLES if persistent else 1):
        sd = run_steps(args.steps, sd)
        samples.append(dict(rt.last_stats))
    rt.close_training()  # (replicas: the records logged since the last regular exchange, and what is in flight)
    sync_all()
    if rank != 0:
        dist.destroy_process_group()
        return

    env_steps = args.steps * n * n_gpus
    bpe = algorithmic_bytes_per_env_step(wl["actions"], wl["masked"])
    if persistent:
        # one launch per (chunk of a) call: the launch IS the step loop
        kernel = "k_rollout_df" if "k_rollout_df" in _lib.variant_symbol(int(samples[0].get("kernel_variant", 0))) else "k_rollout_lane"
        launches = sum(s["dominant_launches"] for s in samples)
        launch_s = sum(s["dominant_ms"] for s in samples) / max(1, launches) / 1e3
        units_per_launch = sum(s["dominant_env_steps"] for s in samples) / max(1, launches)
        if launches == 0:  # (replica-exchange calls time every eighth launch only: fall back to the in-kernel clock)
            launches = sum(s["launches"] for s in samples)
            launch_s = sum(s["device_clock_ms"] for s in samples) / max(1, launches) / 1e3
            units_per_launch = args.steps * n * len(samples) / max(1, launches)
        achieved = bpe * units_per_launch / launch_s / 1e9 if launch_s > 0 else 0.0
        timing = (f"HIP events on the engine's stream around {launches} further launches of the same {args.steps} steps right "
                  "after the timed region; the timed launch itself carries no events (in-kernel clock: device_region_ms)")
        kernel_note = "one launch runs all K vector steps on one CU: a latency-bound dependent chain per step, not a bandwidth-bound kernel"
    elif stats["launches"] < 1.5 * args.steps:
        # turnstile path (513 .. ~60 000 agents, learn_iter): ONE launch per vector step -- the kernel is the step
        s0 = samples[0]
        kernel = "k_step_turn"
        launches = s0["dominant_launches"]
        launch_s = s0["dominant_ms"] / max(1, launches) / 1e3
        units_per_launch = n
        achieved = bpe * n / launch_s / 1e9 if launch_s > 0 else 0.0
        timing = (f"HIP events on the engine's stream around each of the first {launches} launches of one further call of the "
                  "same K steps (the rest of that call is replayed from a HIP graph)")
        kernel_note = (f"one launch per vector step; whole region of that call: {s0['kernel_ms'] / args.steps * 1e3:.2f} us per step; "
                       "agents that share a row hand it on inside the launch (latency chain), the others stream")
    else:
        # several kernels per vector step: the roofline figure is the WHOLE step (algorithmic bytes of one
        # vector step / device time of one vector step, HIP events around the region); the engine
        # additionally samples the first kernel of the step (k_step_fast), reported in the note
        s0 = samples[0]
        kernel = "vector step (k_step_fast + token rounds + k_step_slow + k_advance)"
        launch_s = s0["kernel_ms"] / args.steps / 1e3
        launches = args.steps
        units_per_launch = n
        achieved = bpe * n / launch_s / 1e9 if launch_s > 0 else 0.0
        timing = "HIP events around the whole stream region of one further call of the same K steps"
        kernel_note = (f"k_step_fast alone: {s0['dominant_ms'] / max(1, s0['dominant_launches']) * 1e3:.2f} us per launch over "
                       f"{s0['dominant_launches']} sampled launches; per-kernel split: profiles/")
    # The kernel instantiation that ran (qe_rollout_stats.kernel_variant), as rocprofv3 names it.
    env_name = "TttEnv" if wl.get("env") == "tictactoe" else "HashEnv"
    variant = int(samples[0].get("kernel_variant", 0) or stats.get("kernel_variant", 0))
    kernel_symbol = _lib.variant_symbol(variant, "float", env_name, int(algo.lanes_per_row), args.mode == "vec")
    # HBM bytes from the PMC passes kept under profiles/ -- NOT measured in this run: quoted only if the profile was
    # taken on the same kernel instantiation (its symbol is recorded in the file, with the commit it was collected at).
    traffic_profile = None
    try:
        prof = json.loads((ROOT / "profiles" / "r03_traffic.json").read_text()).get(args.workload)
        if prof and prof.get("kernel_symbol") == kernel_symbol:
            traffic_profile = {"file": "profiles/r03_traffic.json", **prof}
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
            "parallelism": (f"agents sharded x{n_gpus}, table replicas + {'RCCL' if args.backend == 'nccl' else 'gloo (rehearsal)'} "
                            "delta all-gather") if n_gpus > 1 else "1 GPU",
        },
        "roofline": {
            "bound": "hbm",
            "kernel": kernel,
            "kernel_symbol": kernel_symbol,
            "achieved": achieved,
            "peak": HBM_PEAK_GBPS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBPS,
            "step_frac": achieved / HBM_PEAK_GBPS,
            "traffic": None,
            "traffic_from_profile": traffic_profile,
            "alg_bytes_per_env_step": bpe,
            "units_per_launch": units_per_launch,
            "avg_launch_us": launch_s * 1e6,
            "launches_sampled": launches,
            "timing": timing,
            "note": kernel_note,
        },
        "device_region_ms": stats.get("device_clock_ms") or stats["kernel_ms"],
        "timed_call_us": t_call * 1e6,  # run_steps itself; the rest of the timed region is the closing synchronisation
        "host_enqueue_us": stats.get("host_begin_us"),
        "host_wait_us": stats.get("host_end_us"),
        "kernel_launches": stats["launches"],
        "episodes": int(stats["episodes"]),
        "contested_agent_steps": stats["involved"],
        # the fields of the reference's result files (throughput_benchmark.py:251-259, 310-317)
        "reference_fields": {
            "runtime": "gpu_rollout" if n_gpus == 1 else "gpu_rollout_replicas",
            "total_steps": args.steps, "effective_steps": args.steps * n * n_gpus, "elapsed_time": elapsed,
            "throughput": env_steps / elapsed, "step_multiplier": n * n_gpus, "num_agents": n,
            "num_processes": n_gpus, "timestamp": time.strftime("%Y-%m-%dT%H:%M:%S"),
        },
    }
    if not args.no_cpu_baseline and n_gpus == 1:
        line["cpu_baseline"] = cpu_baseline(wl)
        line["speedup_vs_cpu_baseline"] = line["value"] / line["cpu_baseline"]["value"]
        if "compiled_c_value" in line["cpu_baseline"]:
            line["speedup_vs_compiled_c_one_core"] = line["value"] / line["cpu_baseline"]["compiled_c_value"]
    if args.workload == "tictactoe" and n_gpus == 1:
        line["vs_reference_published_single_thread_128_agents_i7_11700K"] = (
            line["value"] / PUBLISHED_TICTACTOE_SINGLE_THREAD_128)
    if use_dist:
        line["delta_sync"] = {"syncs": rt.delta_sync.syncs, "bytes_received_per_gpu": rt.delta_sync.bytes_exchanged,
                              "sync_every": SYNC_EVERY, "collective": "all_gather_into_tensor of (cell, delta) logs",
                              "note": "whole process (warm-up, timed call, roofline samples, final flush); the cadence runs across "
                                      "calls: a timed call shorter than sync_every steps contains an exchange only if the "
                                      "100th logged step falls into it"}
    sys.stdout.flush()
    os.dup2(json_fd, 1)
    print(json.dumps(line), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
