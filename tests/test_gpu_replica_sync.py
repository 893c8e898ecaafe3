"""GPU, one rank over RCCL: the code path the multi-GPU bench takes (`bench.py` under
torch.distributed.run), exercised end to end on the one GPU a test box has.

With world_size 1 there are no remote deltas, so a run with the replica exchange attached (delta log
written by the kernels, asynchronous all-gather every `sync_every` steps on the shared non-default
stream, log double-buffering, chunking at the sync cadence) must equal the plain run bit for bit."""

import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize(("n", "S", "A", "steps", "sync_every"), [(128, 5000, 16, 450, 100), (3000, 20000, 8, 130, 50)])
def test_one_rank_exchange_equals_plain_run(n, S, A, steps, sync_every):
    torch = pytest.importorskip("torch")
    import torch.distributed as dist

    torch.cuda.set_device(0)  # torch's HIP runtime must see the GPU before the engine's does (two runtimes, one process)

    from dist_classicrl_amd.algorithms.base_algorithms.q_learning_optimal import OptimalQLearningBase
    from dist_classicrl_amd.algorithms.runtime.gpu_rollout_runtime import GpuRolloutQLearning
    from dist_classicrl_amd.distributed.delta_sync import attach_engine
    from dist_classicrl_amd.environments import HashTabularEnv
    from dist_classicrl_amd.schedules import ExponentialSchedule

    def run(with_sync):
        algo = OptimalQLearningBase(S, A, 0.99, seed=3)
        rt = GpuRolloutQLearning(algo, ExponentialSchedule(0.1, 1e-5, 0.9995), ExponentialSchedule(1.0, 0.05, 0.9995))
        if with_sync:
            rt.sync_every = sync_every
            rt.delta_sync = attach_engine(algo, sync_every, n)
        _, history, _, sd = rt.run_steps(steps, HashTabularEnv(n, S, A, seed=5), None)
        if with_sync:
            # the cadence runs across calls: a regular exchange every sync_every steps; the records logged
            # after the last one go out when the training is closed
            assert rt.delta_sync.syncs == steps // sync_every
            rt.close_training()
        syncs = rt.delta_sync.syncs if with_sync else 0
        return np.asarray(algo.q_table), np.array(history), sd["states"], syncs

    plain = run(False)
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1,
                                device_id=torch.device("cuda", 0))
    try:
        synced = run(True)
    finally:
        if created:
            dist.destroy_process_group()
    assert synced[3] == -(-steps // sync_every)
    assert np.array_equal(plain[0], synced[0]) and np.count_nonzero(plain[0]) > 100
    assert np.array_equal(plain[1], synced[1]) and np.array_equal(plain[2], synced[2])


def test_short_calls_below_the_exchange_cadence_equal_the_plain_run():
    """A driver that steps 20 vector steps per `run_steps` call with the replica exchange attached: the calls
    that stay below the cadence take the one-call fused path (they only append to the delta log), the call in
    which the 50th logged step falls exchanges; tables, returns and states equal the plain run bit for bit."""
    torch = pytest.importorskip("torch")
    import torch.distributed as dist

    torch.cuda.set_device(0)  # torch's HIP runtime must see the GPU before the engine's does (two runtimes, one process)

    from dist_classicrl_amd.algorithms.base_algorithms.q_learning_optimal import OptimalQLearningBase
    from dist_classicrl_amd.algorithms.runtime.gpu_rollout_runtime import GpuRolloutQLearning
    from dist_classicrl_amd.distributed.delta_sync import attach_engine
    from dist_classicrl_amd.environments import HashTabularEnv
    from dist_classicrl_amd.schedules import ExponentialSchedule

    n, S, A, sync_every, calls, per_call = 128, 5000, 16, 50, 8, 20

    def run(with_sync):
        algo = OptimalQLearningBase(S, A, 0.99, seed=3)
        rt = GpuRolloutQLearning(algo, ExponentialSchedule(0.1, 1e-5, 0.9995), ExponentialSchedule(1.0, 0.05, 0.9995))
        if with_sync:
            rt.sync_every = sync_every
            rt.delta_sync = attach_engine(algo, sync_every, n)
        env, sd, history = HashTabularEnv(n, S, A, seed=5), None, []
        for _ in range(calls):
            try:
                _, h, _, sd = rt.run_steps(per_call, env, sd)
            except ZeroDivisionError:  # (no episode ended in the call: the reference's quirk)
                h, sd = [], env.state_dict()
            history += h
        syncs = 0
        if with_sync:
            assert rt.delta_sync.syncs == calls * per_call // sync_every  # 160 steps: 3 regular exchanges
            rt.close_training()
            syncs = rt.delta_sync.syncs
        return np.asarray(algo.q_table), np.array(history), sd["states"], syncs

    plain = run(False)
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1,
                                device_id=torch.device("cuda", 0))
    try:
        synced = run(True)
    finally:
        if created:
            dist.destroy_process_group()
    assert synced[3] == 4  # + the 10 steps logged after the last regular exchange
    assert np.array_equal(plain[0], synced[0]) and np.count_nonzero(plain[0]) > 100
    assert np.array_equal(plain[1], synced[1]) and np.array_equal(plain[2], synced[2])


def test_bench_under_the_distributed_launcher_env():
    """`bench.py` exactly as the driver's `torch.distributed.run` starts a rank (RANK/WORLD_SIZE/MASTER_* in the
    environment), one rank: prints one JSON line with the contract's fields."""
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(_free_port()))
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "1", "--steps", "1500", "--warmup", "200",
                          "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600, check=True)
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith("{")  # nothing but the JSON line on stdout (RCCL banner -> stderr)
    line = json.loads(lines[0])
    assert line["n_gpus"] == 1 and line["steps"] == 1500 and line["scaling"] == "weak"
    assert line["config"]["sync_every"] == 100 and line["value"] > 1e6
    assert set(line["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
