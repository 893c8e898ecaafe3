"""GPU, two or three ranks sharing the one GPU of a test box (gloo carries the collective, the engines are real):
the replica exchange end to end with actual remote deltas -- delta log written by the kernels, double-buffered
asynchronous exchange, one-launch scatter-add of the other rank's records -- against a single-process
simulation of the same protocol on the C oracle.

float32 atomics add the remote records in no fixed order, so tables are compared with a tolerance
(1e-5 relative); actions are integers and must match exactly."""

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

    # simulation: each rank's records are its per-step table increments; the other ranks add them one
    # chunk late (overlap) and at the final flush
    runs = [c_oracle.CHashRollout(N_PER_RANK, S, A, agent_offset=r * N_PER_RANK, dtype=np.float32) for r in ranks]
    eps, lr = np.full(1, 0.2), np.full(1, 0.1)

    def apply_others(recs):
        for me in ranks:
            for other in ranks:
                if other != me:
                    np.add.at(runs[me].q.reshape(-1), *recs[other])

    def chunk_records(run):
        cells, deltas = [], []
        for _ in range(SYNC):
            before = run.q.copy()
            run.run(eps, lr, log_episodes=False)
            idx = np.flatnonzero((run.q != before).ravel())
            cells.append(idx)
            deltas.append((run.q.ravel()[idx] - before.ravel()[idx]).astype(np.float32))
        return np.concatenate(cells), np.concatenate(deltas)

    late = None
    for _ in range(CHUNKS):
        recs = [chunk_records(run) for run in runs]
        if late is not None:
            apply_others(late)
        late = recs
    apply_others(late)

    for r in ranks:
        assert np.array_equal(obs[r], runs[r].obs), "agents took a different path than in the simulation"
        bad = ~np.isclose(got[r], runs[r].q, rtol=1e-5, atol=1e-6)
        assert not bad.any(), (f"rank {r}: {int(bad.sum())} cells differ from the simulation, largest difference "
                               f"{np.abs(got[r] - runs[r].q).max():.3g}, first at {np.argwhere(bad)[:5].tolist()}: "
                               f"{got[r][bad][:5]} vs {runs[r].q[bad][:5]}")
        assert np.allclose(got[0], got[r], rtol=1e-5, atol=1e-6)
    assert np.count_nonzero(got[0]) > 500 and not np.array_equal(obs[0], obs[1])
