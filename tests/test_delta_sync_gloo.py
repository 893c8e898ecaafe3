"""CPU, world_size 2, gloo: the multi-GPU replica protocol (agents sharded per rank, full table
replica per rank, all-gather of (cell, delta) logs every `sync_every` steps, remote deltas added
locally) exercised with the C oracle standing in for the engine on each rank.

Checks: (1) DeltaSync moves exactly the other rank's records (rank 0's replica equals a
single-process simulation of the same protocol bit for bit); (2) the two replicas agree up to
float32 summation order; (3) the agents really are sharded (global agent ids, disjoint draws)."""

import os
import socket

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402

N_PER_RANK, S, A, CHUNK, CHUNKS = 64, 400, 8, 10, 6


def _chunk_schedules(k):
    from oracle import c_oracle

    eps, _ = c_oracle.exp_schedule(1.0, 0.01, 0.995, N_PER_RANK, CHUNK * CHUNKS)
    lr, _ = c_oracle.exp_schedule(0.1, 1e-5, 0.995, N_PER_RANK, CHUNK * CHUNKS)
    return eps[k * CHUNK:(k + 1) * CHUNK], lr[k * CHUNK:(k + 1) * CHUNK]


def _run_chunk(run, k):
    """Advance one rank by CHUNK steps; return its (cell, delta) records for the exchange."""
    before = run.q.copy()
    run.run(*_chunk_schedules(k), log_episodes=False)
    cells = np.flatnonzero((run.q != before).ravel()).astype(np.int32)
    deltas = (run.q.ravel()[cells] - before.ravel()[cells]).astype(np.float32)
    return cells, deltas


def _apply(q, cells, deltas):
    np.add.at(q.reshape(-1), cells, deltas)


def _worker(rank, world, port, out_dir, overlap, one_launch=False, gathered=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dist_classicrl_amd.distributed.delta_sync import DeltaSync
    from oracle import c_oracle

    run = c_oracle.CHashRollout(N_PER_RANK, S, A, agent_offset=rank * N_PER_RANK, dtype=np.float32)

    def apply_fn(entries, count):
        e = entries[:count].numpy()
        _apply(run.q, e[:, 0], e[:, 1].view(np.float32))

    def apply_skip_fn(entries, total, skip_begin, skip_end):  # the engine's one-launch form
        e = entries[:total].numpy()
        keep = np.r_[0:skip_begin, skip_end:total]
        _apply(run.q, e[keep, 0], e[keep, 1].view(np.float32))

    # one_launch: fixed-size logs like the engine's (slot = step * n + agent), so count == capacity
    cap = CHUNK * N_PER_RANK if one_launch else S * A
    def apply_gathered_fn(g, capacity, count, world_, rank_):  # the engine's whole apply step (qe_delta_apply_gathered_dev)
        assert capacity == cap and world_ == world and rank_ == rank
        others = np.concatenate([g[r, :count].numpy() for r in range(world) if r != rank])
        order = np.argsort(others[:, 0], kind="stable")  # by cell; within a cell (rank, slot) order
        _apply(run.q, others[order, 0], others[order, 1].view(np.float32))

    sync = DeltaSync(cap, "cpu", apply_fn, overlap=overlap, apply_skip_fn=apply_skip_fn if one_launch else None,
                     apply_gathered_fn=apply_gathered_fn if gathered else None)
    for k in range(CHUNKS):
        cells, deltas = _run_chunk(run, k)
        # every rank must exchange the same record count: pad with (cell 0, +0.0) no-ops
        cnt = torch.tensor([cells.size])
        dist.all_reduce(cnt, op=dist.ReduceOp.MAX)
        count = cap if one_launch else int(cnt.item())
        log = sync.log  # the buffer the "engine" writes this chunk's records into
        log.zero_()
        log[:cells.size, 0] = torch.from_numpy(cells)
        log[:cells.size, 1] = torch.from_numpy(deltas.view(np.int32))
        sync.exchange(count)
    sync.flush()
    np.save(os.path.join(out_dir, f"q{rank}.npy"), run.q)
    np.save(os.path.join(out_dir, f"obs{rank}.npy"), run.obs)
    assert sync.syncs == CHUNKS
    dist.destroy_process_group()


@pytest.mark.parametrize(("world", "overlap", "one_launch", "gathered"),
                         [(2, False, False, False), (2, True, False, False), (2, True, True, False), (3, True, True, False),
                          (3, False, False, False), (2, True, True, True), (3, True, False, True)])
def test_replica_sync_matches_single_process_simulation(tmp_path, world, overlap, one_launch, gathered):
    from oracle import c_oracle

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(world, port, str(tmp_path), overlap, one_launch, gathered), nprocs=world, join=True)
    got = [np.load(tmp_path / f"q{r}.npy") for r in range(world)]

    # single-process simulation of the same protocol: without overlap the other ranks' records are
    # applied at the sync point (in rank order), with overlap one chunk later (and at the final flush)
    runs = [c_oracle.CHashRollout(N_PER_RANK, S, A, agent_offset=r * N_PER_RANK, dtype=np.float32)
            for r in range(world)]

    def apply_others(recs):
        for me in range(world):
            for other in range(world):
                if other != me:
                    _apply(runs[me].q, *recs[other])

    late = None
    for k in range(CHUNKS):
        recs = [_run_chunk(run, k) for run in runs]
        ready = late if overlap else recs
        if ready is not None:
            apply_others(ready)
        late = recs
    if overlap:
        apply_others(late)
    for r in range(world):
        assert np.array_equal(got[r], runs[r].q), r
    assert np.count_nonzero(got[0]) > 100
    for r in range(1, world):
        assert np.allclose(got[0], got[r], rtol=1e-5, atol=1e-6)  # same sums, different fp32 summation order
    assert not np.array_equal(np.load(tmp_path / "obs0.npy"), np.load(tmp_path / "obs1.npy"))


# ------------------------------------------------------------------------------------------------
# with an exchange after EVERY step (and no overlap) the replica protocol is
# single-process `learn_vec` over all agents, up to the order in which increments are added.
K1_STEPS = 6


def _worker_every_step(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dist_classicrl_amd.distributed.delta_sync import DeltaSync
    from oracle import c_oracle

    run = c_oracle.CHashRollout(N_PER_RANK, S, A, agent_offset=rank * N_PER_RANK, dtype=np.float32, mode="vec")

    def apply_fn(entries, count):
        e = entries[:count].numpy()
        _apply(run.q, e[:, 0], e[:, 1].view(np.float32))

    sync = DeltaSync(N_PER_RANK, "cpu", apply_fn, overlap=False)
    eps, _ = c_oracle.exp_schedule(0.3, 0.3, 1.0, 2 * N_PER_RANK, K1_STEPS)
    lr, _ = c_oracle.exp_schedule(0.1, 0.1, 1.0, 2 * N_PER_RANK, K1_STEPS)
    actions = []
    for t in range(K1_STEPS):
        before = run.q.copy()
        out = run.run(eps[t:t + 1], lr[t:t + 1], trace=True, log_episodes=False)
        actions.append(out["actions"][0])
        cells = np.flatnonzero((run.q != before).ravel()).astype(np.int32)
        deltas = (run.q.ravel()[cells] - before.ravel()[cells]).astype(np.float32)
        log = sync.log
        log.zero_()
        log[:cells.size, 0] = torch.from_numpy(cells)
        log[:cells.size, 1] = torch.from_numpy(deltas.view(np.int32))
        sync.exchange(N_PER_RANK)  # padded with (cell 0, +0.0) no-ops
    np.save(os.path.join(out_dir, f"k1_q{rank}.npy"), run.q)
    np.save(os.path.join(out_dir, f"k1_a{rank}.npy"), np.stack(actions))
    dist.destroy_process_group()


def test_exchange_every_step_equals_learn_vec_over_all_agents(tmp_path):
    from oracle import c_oracle

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker_every_step, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    whole = c_oracle.CHashRollout(2 * N_PER_RANK, S, A, dtype=np.float32, mode="vec")
    eps, _ = c_oracle.exp_schedule(0.3, 0.3, 1.0, 2 * N_PER_RANK, K1_STEPS)
    lr, _ = c_oracle.exp_schedule(0.1, 0.1, 1.0, 2 * N_PER_RANK, K1_STEPS)
    ref = whole.run(eps, lr, trace=True, log_episodes=False)
    got_actions = np.concatenate([np.load(tmp_path / "k1_a0.npy"), np.load(tmp_path / "k1_a1.npy")], axis=1)
    assert np.array_equal(got_actions, ref["actions"])  # same agents, same draws, same policy
    for r in (0, 1):
        q = np.load(tmp_path / f"k1_q{r}.npy")
        # float32 deltas added in a different order than np.add.at: 1e-6 relative per increment
        assert np.allclose(q, whole.q, rtol=2e-6, atol=1e-7)
    assert np.count_nonzero(whole.q) > 100
