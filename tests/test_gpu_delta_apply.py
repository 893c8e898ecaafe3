"""GPU: the apply step of the replica exchange (replaces the parameter server's apply loop,
``q_learning_async_dist.py:359-447``) and the turnstile path's residency assumption.

* ``qe_delta_apply_gathered_dev`` -- the engine's own stable radix sort of the other ranks' records + one sequential
  run per cell -- against a CPU simulation (NumPy stable sort, float32 additions in (rank, slot) order per cell) and
  against round 2's path (host-sorted records through ``qe_delta_apply_sorted_dev``): bit for bit, including eight
  ranks' worth of records (5.7 M) at the shape of one BASELINE config-4 shard.
* ``k_step_turn`` (one launch per vector step, workgroups that wait for each other) while another kernel holds a
  quarter of the CUs: finishes, bit-exact.
"""

import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _product():
    from dist_classicrl_amd import _lib, environments, schedules
    from dist_classicrl_amd.algorithms.base_algorithms.q_learning_optimal import OptimalQLearningBase
    from dist_classicrl_amd.algorithms.runtime.gpu_rollout_runtime import GpuRolloutQLearning

    return _lib, OptimalQLearningBase, GpuRolloutQLearning, environments, schedules


class _DevBuf:
    """Plain device memory from the HIP runtime the engine is linked against (no torch in this process)."""

    def __init__(self, host: np.ndarray):
        self.hip = C.CDLL("libamdhip64.so")
        self.ptr, self.nbytes = C.c_void_p(), host.nbytes
        assert self.hip.hipMalloc(C.byref(self.ptr), C.c_size_t(max(self.nbytes, 8))) == 0
        assert self.hip.hipMemcpy(self.ptr, host.ctypes.data_as(C.c_void_p), C.c_size_t(self.nbytes), 1) == 0  # H2D

    def free(self):
        assert self.hip.hipFree(self.ptr) == 0


def _records(rng, world, capacity, count, cells, hot=0.0):
    """(world, capacity) records {uint32 cell, float32 delta}; a fraction `hot` of them on a handful of cells."""
    cell = rng.integers(0, cells, size=(world, capacity), dtype=np.uint32)
    if hot:
        few = rng.integers(0, cells, size=7, dtype=np.uint32)
        pick = rng.random((world, capacity)) < hot
        cell[pick] = few[rng.integers(0, 7, size=int(pick.sum()))]
    delta = (rng.standard_normal((world, capacity)) * 0.1).astype(np.float32)
    rec = np.empty((world, capacity, 2), dtype=np.uint32)
    rec[..., 0] = cell
    rec[..., 1] = delta.view(np.uint32)
    return rec


def _expected(q0_cells, rec, count, rank):
    """CPU simulation: the other ranks' first `count` records, stably sorted by cell, float32 adds in that order."""
    others = np.concatenate([rec[r, :count] for r in range(rec.shape[0]) if r != rank])
    order = np.argsort(others[:, 0], kind="stable")
    cells, deltas = others[order, 0], others[order, 1].view(np.float32)
    uniq, inv = np.unique(cells, return_inverse=True)
    acc = q0_cells(uniq).astype(np.float32)
    np.add.at(acc, inv, deltas)  # sequential float32 additions in array order
    return uniq, acc, others[order]


@pytest.mark.parametrize(("S", "A", "world", "capacity", "count", "rank", "hot"), [
    (1000, 8, 2, 500, 500, 0, 0.0),
    (1000, 8, 3, 4000, 3000, 1, 0.3),        # partly filled segments, the middle rank skips its own
    (50, 4, 8, 20000, 20000, 7, 0.5),        # far more records than cells: long runs per cell
    (1_000_000, 16, 4, 12800, 12800, 2, 0.01),   # headline shape, 100 steps x 128 agents per rank
    (1_000_000, 16, 8, 409_600, 409_600, 3, 0.02),  # C3 sharded over 8: 100 steps x 4096 agents per rank
    (300, 300, 2, 70000, 65536, 1, 0.0),     # A > 256 (row stride not a power of two), ragged tile boundary
    (7, 3, 2, 10, 1, 1, 0.0),                # one record
    (1000, 8, 5, 100, 63, 4, 0.0),           # less than a wavefront's batch per rank, the last rank's view
    (100_000, 8, 3, 5000, 4097, 0, 0.1),     # one record past a tile, two tiles and two records in all
])
def test_gathered_apply_matches_cpu_simulation_and_the_sorted_path(S, A, world, capacity, count, rank, hot):
    _lib, Algo, _, _, _ = _product()
    lib = _lib.load()
    rng = np.random.default_rng(S + world)
    algo = Algo(S, A, 0.99, seed=0)
    ld = int(lib.qe_table_row_stride(algo.handle))
    q0 = rng.standard_normal((S, A)).astype(np.float32)
    algo.q_table = q0
    # records address padded cells (row * ld + column) like the engine's own log
    rec = _records(rng, world, capacity, count, S * A, hot)
    flat = rec[..., 0].astype(np.int64)
    rec[..., 0] = (flat // A * ld + flat % A).astype(np.uint32)
    uniq, acc, sorted_others = _expected(lambda cells: q0[cells // ld, cells % ld], rec, count, rank)
    buf = _DevBuf(rec)
    _lib.check(lib.qe_delta_apply_gathered_dev(algo.handle, buf.ptr, capacity, count, world, rank))
    _lib.check(lib.qe_synchronize(algo.handle))
    got = np.asarray(algo.q_table)
    want = q0.copy()
    want[uniq // ld, uniq % ld] = acc
    assert np.array_equal(got, want)
    # round 2's path on a second engine: host-sorted records -> qe_delta_apply_sorted_dev
    algo2 = Algo(S, A, 0.99, seed=0)
    algo2.q_table = q0
    buf2 = _DevBuf(np.ascontiguousarray(sorted_others))
    _lib.check(lib.qe_delta_apply_sorted_dev(algo2.handle, buf2.ptr, sorted_others.shape[0]))
    _lib.check(lib.qe_synchronize(algo2.handle))
    assert np.array_equal(np.asarray(algo2.q_table), got)
    buf.free()
    buf2.free()


def test_eight_ranks_worth_of_records_at_the_c4_shard_shape():
    """BASELINE config 4: 65 536 agents over 8 GPUs, exchange every 100 steps -> each replica applies 7 x 819 200 remote
    records to its 1e7 x 32 table.  Touched cells are read back (the table itself is 1.28 GB)."""
    _lib, Algo, _, _, _ = _product()
    lib = _lib.load()
    S, A, world, per_rank, rank = 10_000_000, 32, 8, 819_200, 5
    rng = np.random.default_rng(4)
    algo = Algo(S, A, 0.99, seed=0)
    assert int(lib.qe_table_row_stride(algo.handle)) == A
    rec = _records(rng, world, per_rank, per_rank, S * A, hot=0.001)
    uniq, acc, _ = _expected(lambda cells: np.zeros(cells.size, dtype=np.float32), rec, per_rank, rank)
    buf = _DevBuf(rec)
    _lib.check(lib.qe_delta_apply_gathered_dev(algo.handle, buf.ptr, per_rank, per_rank, world, rank))
    _lib.check(lib.qe_synchronize(algo.handle))
    vals = np.empty(uniq.size, dtype=np.float64)
    states, actions = (uniq // A).astype(np.int32), (uniq % A).astype(np.int32)
    _lib.check(lib.qe_table_cells(algo.handle, _lib.ptr(states, C.c_int32), _lib.ptr(actions, C.c_int32), uniq.size,
                                  _lib.ptr(vals, C.c_double), 0))
    assert np.array_equal(vals.astype(np.float32), acc)
    # this rank's own records were not applied: cells only IT touched are still zero
    own = np.setdiff1d(rec[rank, :, 0], uniq)[:100000]
    if own.size:
        v2 = np.empty(own.size, dtype=np.float64)
        s2, a2 = (own // A).astype(np.int32), (own % A).astype(np.int32)
        _lib.check(lib.qe_table_cells(algo.handle, _lib.ptr(s2, C.c_int32), _lib.ptr(a2, C.c_int32), own.size,
                                      _lib.ptr(v2, C.c_double), 0))
        assert not v2.any()
    buf.free()


@pytest.mark.parametrize(("n", "S", "A", "steps"), [(8192, 10_000_000, 32, 120), (4096, 1_000_000, 16, 150)])
def test_turnstile_path_with_a_quarter_of_the_chip_taken(n, S, A, steps):
    """The turnstile kernel's workgroups wait for each other inside one launch, so all of them must be resident; the
    engine sizes the path from the runtime's occupancy answer minus a quarter of the CUs.  Here a filler kernel holds a
    quarter of the CUs (one workgroup per CU, most of its LDS) on another stream while the rollout runs: it must finish
    (no ERR_TURN_TIMEOUT) and equal the C oracle bit for bit."""
    from oracle import c_oracle

    _lib, Algo, Runtime, envs, sch = _product()
    lib = _lib.load()
    algo = Algo(S, A, 0.99, seed=0)
    rt = Runtime(algo, sch.ExponentialSchedule(0.1, 1e-5, 0.995), sch.ExponentialSchedule(1.0, 0.01, 0.995))
    env = envs.HashTabularEnv(n, S, A, seed=1)
    sd, history = None, []
    for k in (steps // 3, steps // 3, steps - 2 * (steps // 3)):
        _lib.check(lib.qe_debug_occupy_cus(algo.handle, 64, 150_000))  # 64 of 256 CUs for up to 150 ms
        _avg, h, env, sd = rt.run_steps(k, env, sd)
        history += h
        assert _lib.decode_variant(rt.last_stats["kernel_variant"])["path"] == "turnstile"
    ref = c_oracle.CHashRollout(n, S, A, dtype=np.float32)
    eps, _ = c_oracle.exp_schedule(1.0, 0.01, 0.995, n, steps)
    lr, _ = c_oracle.exp_schedule(0.1, 1e-5, 0.995, n, steps)
    want = ref.run(eps, lr)
    assert np.array_equal(np.asarray(algo.q_table), ref.q)
    assert np.array_equal(np.array(history, dtype=np.float32), want["history"])
    assert np.array_equal(sd["states"], ref.obs)
