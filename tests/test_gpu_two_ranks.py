"""GPU, two or three ranks sharing the one GPU of a test box (gloo carries the collective, the engines are real):
the replica exchange end to end with actual remote deltas -- delta log written by the kernels, double-buffered
asynchronous exchange, one-launch scatter-add of the other rank's records -- against a single-process
simulation of the same protocol on the C oracle.

The remote records are applied deterministically (sorted by cell, (rank, slot) order within a cell:
``qe_delta_apply_sorted_dev``), so every replica must equal the simulation BIT FOR BIT; replicas differ
from each other in the last bits only (each adds its own increments first), never in the policy."""

import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu



def _worker(rank, world, port, out_dir, N_PER_RANK, S, A, SYNC, CHUNKS):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dist_classicrl_amd.algorithms.base_algorithms.q_learning_optimal import OptimalQLearningBase
    from dist_classicrl_amd.algorithms.runtime.gpu_rollout_runtime import GpuRolloutQLearning
    from dist_classicrl_amd.distributed.delta_sync import attach_engine
    from dist_classicrl_amd.environments import HashTabularEnv
    from dist_classicrl_amd.schedules import ConstantSchedule

    algo = OptimalQLearningBase(S, A, 0.99, seed=0)
    env = HashTabularEnv(N_PER_RANK, S, A, seed=1, agent_offset=rank * N_PER_RANK)
    rt = GpuRolloutQLearning(algo, ConstantSchedule(0.1), ConstantSchedule(0.2))
    rt.sync_every = SYNC
    rt.delta_sync = attach_engine(algo, SYNC, N_PER_RANK)
    rt.trace_actions = None
    _, history, _, sd = rt.run_steps(SYNC * CHUNKS, env, None)
    np.save(os.path.join(out_dir, f"q{rank}.npy"), np.asarray(algo.q_table))
    np.save(os.path.join(out_dir, f"obs{rank}.npy"), sd["states"])
    np.save(os.path.join(out_dir, f"meta{rank}.npy"), np.array([rt.delta_sync.syncs, rt.delta_sync.bytes_exchanged]))
    dist.destroy_process_group()


@pytest.mark.parametrize(("world", "N_PER_RANK", "S", "A", "SYNC", "CHUNKS"), [
    (2, 96, 4000, 8, 20, 4),        # persistent kernel per rank
    (3, 96, 4000, 8, 20, 3),        # three ranks: the middle one skips its own segment of the gathered log
    (2, 2100, 200000, 8, 10, 3),    # wide step-wise path per rank
])
def test_ranks_sharing_one_gpu_match_the_simulated_protocol(tmp_path, world, N_PER_RANK, S, A, SYNC, CHUNKS):
    torch = pytest.importorskip("torch")
    import torch.multiprocessing as mp

    from oracle import c_oracle

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(world, port, str(tmp_path), N_PER_RANK, S, A, SYNC, CHUNKS), nprocs=world, join=True)
    ranks = range(world)
    got = [np.load(tmp_path / f"q{r}.npy") for r in ranks]
    obs = [np.load(tmp_path / f"obs{r}.npy") for r in ranks]
    for r in ranks:
        syncs, nbytes = np.load(tmp_path / f"meta{r}.npy")
        assert syncs == CHUNKS and nbytes == CHUNKS * SYNC * N_PER_RANK * 8 * (world - 1)

    # simulation: each rank's records are its per-agent increments in (step, agent) order -- the content of
    # the engine's delta log; the other ranks add them one chunk late (overlap) and at the final flush,
    # rank by rank, each in slot order (= per cell: (rank, slot) order, the engine's sorted apply)
    runs = [c_oracle.CHashRollout(N_PER_RANK, S, A, agent_offset=r * N_PER_RANK, dtype=np.float32) for r in ranks]
    eps, lr = np.full(SYNC, 0.2), np.full(SYNC, 0.1)

    def apply_others(recs):
        for me in ranks:
            for other in ranks:
                if other != me:
                    np.add.at(runs[me].q.reshape(-1), *recs[other])

    def chunk_records(run):
        out = run.run(eps, lr, log_episodes=False, delta_log=True)
        return out["cells"].astype(np.int64), out["deltas"]

    late = None
    for _ in range(CHUNKS):
        recs = [chunk_records(run) for run in runs]
        if late is not None:
            apply_others(late)
        late = recs
    apply_others(late)

    for r in ranks:
        assert np.array_equal(obs[r], runs[r].obs), "agents took a different path than in the simulation"
        bad = got[r] != runs[r].q
        assert not bad.any(), (f"rank {r}: {int(bad.sum())} cells differ from the simulation, largest difference "
                               f"{np.abs(got[r] - runs[r].q).max():.3g}, first at {np.argwhere(bad)[:5].tolist()}: "
                               f"{got[r][bad][:5]} vs {runs[r].q[bad][:5]}")
        assert np.allclose(got[0], got[r], rtol=1e-5, atol=1e-6)  # same sums, another fp32 summation order
    assert np.count_nonzero(got[0]) > 500 and not np.array_equal(obs[0], obs[1])


# ------------------------------------------------------------------------------------------------
# with an exchange after EVERY step (and no overlap) the replica protocol is
# single-GPU `learn_vec` over all agents, up to the order in which increments are added.  Here with the
# real engines on both sides.
K1_N, K1_S, K1_A, K1_STEPS = 64, 500, 8, 8


def _worker_every_step(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dist_classicrl_amd.algorithms.base_algorithms.q_learning_optimal import OptimalQLearningBase
    from dist_classicrl_amd.algorithms.runtime.gpu_rollout_runtime import GpuRolloutQLearning
    from dist_classicrl_amd.distributed.delta_sync import attach_engine
    from dist_classicrl_amd.environments import HashTabularEnv
    from dist_classicrl_amd.schedules import ConstantSchedule

    algo = OptimalQLearningBase(K1_S, K1_A, 0.99, seed=0)
    env = HashTabularEnv(K1_N, K1_S, K1_A, seed=1, agent_offset=rank * K1_N)
    rt = GpuRolloutQLearning(algo, ConstantSchedule(0.1), ConstantSchedule(0.3), learn_mode="vec")
    rt.sync_every = 1
    rt.delta_sync = attach_engine(algo, 1, K1_N, overlap=False)
    rt.trace_actions = True
    try:
        rt.run_steps(K1_STEPS, env, None)
    except ZeroDivisionError:
        pass
    np.save(os.path.join(out_dir, f"k1_q{rank}.npy"), np.asarray(algo.q_table))
    np.save(os.path.join(out_dir, f"k1_a{rank}.npy"), rt.last_trace)
    assert rt.delta_sync.syncs == K1_STEPS
    dist.destroy_process_group()


def test_exchange_every_step_equals_learn_vec_over_all_agents_on_the_engines(tmp_path):
    pytest.importorskip("torch")
    import torch.multiprocessing as mp

    from dist_classicrl_amd.algorithms.base_algorithms.q_learning_optimal import OptimalQLearningBase
    from dist_classicrl_amd.algorithms.runtime.gpu_rollout_runtime import GpuRolloutQLearning
    from dist_classicrl_amd.environments import HashTabularEnv
    from dist_classicrl_amd.schedules import ConstantSchedule

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker_every_step, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    # one engine, all agents, learn_vec
    algo = OptimalQLearningBase(K1_S, K1_A, 0.99, seed=0)
    rt = GpuRolloutQLearning(algo, ConstantSchedule(0.1), ConstantSchedule(0.3), learn_mode="vec")
    rt.trace_actions = True
    try:
        rt.run_steps(K1_STEPS, HashTabularEnv(2 * K1_N, K1_S, K1_A, seed=1), None)
    except ZeroDivisionError:
        pass
    whole_q = np.asarray(algo.q_table)
    got_actions = np.concatenate([np.load(tmp_path / "k1_a0.npy"), np.load(tmp_path / "k1_a1.npy")], axis=1)
    assert np.array_equal(got_actions, rt.last_trace)  # same agents, same draws, same policy: bit-exact actions
    for r in (0, 1):
        q = np.load(tmp_path / f"k1_q{r}.npy")
        # float32 increments added in another order than np.add.at's float64-add-then-round: 1e-6 relative each
        assert np.allclose(q, whole_q, rtol=2e-6, atol=1e-7)
    assert np.count_nonzero(whole_q) > 100
