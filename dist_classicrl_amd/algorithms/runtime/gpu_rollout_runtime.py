"""``GpuRolloutQLearning``: the reference's single_thread / parallel / distributed runtimes
collapsed into one device-resident rollout loop.

``run_steps`` has the contract of ``SingleThreadQLearning.run_steps``
(``dist_classicrl/algorithms/runtime/single_thread_runtime.py:28-76``).  With a
:class:`~dist_classicrl_amd.environments.DeviceVecEnv` the whole hot loop -- ``steps`` x
``run_single_step`` (``base_runtime.py:184-222``) -- is ONE ``qe_rollout`` call: select, env.step,
learn, schedule reads and episode bookkeeping all stay on the GPU.  With any other environment
object it falls back to the per-step host loop of :class:`BaseRuntime` (still GPU select/learn).
"""

from __future__ import annotations

import ctypes as C

import numpy as np

from dist_classicrl_amd import _lib
from dist_classicrl_amd.environments.device_envs import DeviceVecEnv

from .base_runtime import BaseRuntime, _count_agents

_SHORT_CALL = 64  # up to here the schedule values are produced by a plain Python loop (no NumPy set-up cost)
_ShortSchedule = C.c_double * _SHORT_CALL  # (one array type for every short call: ctypes builds a type per length, slowly)


def _schedule_values(schedule, n_updates, count):
    """The ``count`` values ``get_value(); update(n_updates); ...`` reads, as a ctypes double array
    (short calls) or a float64 ndarray; the schedule is left advanced (base_runtime.py:248,262-263)."""
    if count <= _SHORT_CALL:
        out = _ShortSchedule()
        into = getattr(schedule, "advance_into", None)
        if into is not None:
            into(n_updates, count, out)
        else:  # duck-typed schedule (e.g. the reference's classes)
            get, update = schedule.get_value, schedule.update
            for t in range(count):
                out[t] = get()
                update(n_updates)
        return out
    fast = getattr(schedule, "advance_values", None)
    if fast is not None:
        return np.ascontiguousarray(fast(n_updates, count), dtype=np.float64)
    out = np.empty(count, dtype=np.float64)  # duck-typed schedule (e.g. the reference's classes)
    for t in range(count):
        out[t] = schedule.get_value()
        schedule.update(n_updates)
    return out


def _f64_ptr(values):
    if isinstance(values, np.ndarray):
        return _lib.ptr(values, C.c_double)
    return C.cast(values, C.POINTER(C.c_double))


def _sequential_sum(values: np.ndarray):
    if len(values) == 0:
        return 0
    return np.cumsum(values, dtype=values.dtype)[-1]


class GpuRolloutQLearning(BaseRuntime):
    """Drop-in for ``SingleThreadQLearning`` (and, semantically, for the parallel/MPI runtimes)."""

    def __init__(self, algorithm, lr_schedule, exploration_rate_schedule, learn_mode="iter") -> None:
        super().__init__(algorithm, lr_schedule, exploration_rate_schedule)
        if learn_mode not in ("iter", "vec"):
            msg = "learn_mode must be 'iter' (reference `learn`, sequential) or 'vec' (`learn_vec`)"
            raise ValueError(msg)
        self.learn_mode = learn_mode
        self.last_stats = None  # accumulated qe_rollout_stats of the latest run_steps call
        self._as_list = None    # returns of the call in progress as Python floats (see _rollout)
        self._run_sum = np.zeros(1, dtype=np.float32)
        self.delta_sync = None  # dist_classicrl_amd.distributed.DeltaSync (multi-GPU replicas)
        self.sync_every = 100   # vector steps between two replica exchanges (BASELINE config 4)
        self._since_sync = 0    # vector steps logged since the last exchange (the cadence runs across calls)
        # set to True to collect every action of the following run_steps calls (tests); the (steps, n)
        # array of the latest call is left in `last_trace` (and, for older callers, here)
        self.trace_actions = None
        self.last_trace = None
        # element type of the reward history `run_steps` returns: "float" (list of Python floats, the
        # float32 returns widened exactly), "float32" (list of numpy.float32 scalars, the reference's
        # element type, ~2x slower to build) or "array" (the float32 array itself: no per-episode
        # Python object; for runs that finish millions of episodes per call)
        self.history_type = "float"
        # scratch of the one-call form of a short run_steps (qe_rollout_fused)
        self._fused_cap = 4096
        self._fused_step = np.empty(self._fused_cap, dtype=np.int32)
        self._fused_ret = np.empty(self._fused_cap, dtype=np.float32)
        self._fused_step_p = _lib.ptr(self._fused_step, C.c_int32)
        self._fused_ret_p = _lib.ptr(self._fused_ret, C.c_float)
        self._fused_sum = C.c_float()
        self._fused_stats = _lib.RolloutStats()

    def _history(self, rets: np.ndarray):
        if self.history_type == "array":
            return rets
        return list(rets) if self.history_type == "float32" else rets.tolist()

    def init_training(self) -> None:
        return None

    def close_training(self) -> None:
        self.flush_replica_exchange()

    def flush_replica_exchange(self) -> None:
        """Exchange the records logged since the last regular exchange (a training that does not end on a
        multiple of ``sync_every`` steps) and apply everything still in flight."""
        sync = self.delta_sync
        if sync is None:
            return
        if self._since_sync:
            sync.exchange(self._since_sync * sync.capacity // self.sync_every)
            _lib.check(_lib.load().qe_delta_log_reset(self.algorithm.handle))
            self._since_sync = 0
        sync.flush()

    def _prepare_env(self, env):
        if isinstance(env, DeviceVecEnv):
            env.bind(self.algorithm)
        return env

    # ------------------------------------------------------------------ fused rollout
    _PIPELINE_CHUNK = 2000  # vector steps per launch when nothing else (log capacity, sync cadence) binds

    def _collect(self, lib, algo, st, done, total, history, ep_steps):
        if st.episodes_dropped:
            # cannot happen with chunks sized by qe_rollout_chunk_limit; never lose returns silently
            msg = f"episode log overflow: {st.episodes_dropped} episode returns were dropped"
            raise _lib.EngineError(msg)
        cnt = int(st.episodes)
        if cnt:
            step_idx = np.empty(cnt, dtype=np.int32)
            ret = np.empty(cnt, dtype=np.float32)
            lib.qe_episode_log(algo.handle, cnt, _lib.ptr(step_idx, C.c_int32), None, _lib.ptr(ret, C.c_float))
            history.append(ret)
            if self._as_list is not None:
                # Python floats and the running float32 sum chunk by chunk: under the launches that follow instead of
                # behind the last one (130 000 returns of a 20 000-step call: 2 ms)
                self._as_list.extend(ret.tolist())
                self._run_sum = np.cumsum(np.concatenate((self._run_sum, ret)), dtype=np.float32)[-1:]
            if done:
                step_idx += done
            ep_steps.append(step_idx)
        for f in total:
            total[f] += getattr(st, f)
        self._variants.add(int(st.kernel_variant))

    def _rollout(self, env, steps, learn, as_list=False):
        """Returns ``(returns, step of each return)``; with ``as_list`` the returns are also left as a list of Python floats
        in ``self._as_list`` and their sequential float32 sum in ``self._run_sum[0]`` (single_thread_runtime.py:67)."""
        lib = _lib.load()
        self._as_list = [] if as_list else None
        self._run_sum = np.zeros(1, dtype=np.float32)
        algo = self.algorithm
        n = env.num_agents
        env._resident = None  # the device state moves on
        mode = _lib.LEARN_ITER if self.learn_mode == "iter" else _lib.LEARN_VEC
        total = {"kernel_ms": 0.0, "launches": 0, "episodes": 0, "involved": 0, "episodes_dropped": 0,
                 "dominant_ms": 0.0, "dominant_launches": 0, "dominant_env_steps": 0, "device_clock_ms": 0.0,
                 "host_begin_us": 0.0, "host_end_us": 0.0, "complex_steps": 0}
        history, ep_steps, traces = [], [], []
        self._variants = set()  # kernel builds the launches of this call ran (qe_rollout_stats.kernel_variant)
        chunk_max = max(1, int(lib.qe_rollout_chunk_limit(algo.handle, env.handle, 1 if learn else 0)))
        sync = self.delta_sync if learn else None
        collect_trace = self.trace_actions is not None and self.trace_actions is not False
        if learn and not collect_trace:
            chunk_max = min(chunk_max, self._PIPELINE_CHUNK)
        # Launches of the call: cut where the episode log could overflow and -- with replicas -- where
        # `sync_every` steps have been logged since the last exchange (the log keeps filling across calls:
        # a call shorter than the cadence exchanges nothing, the records wait for the calls that follow).
        sizes, exchanges, left, since = [], [], steps, self._since_sync if sync is not None else 0
        while left > 0:
            k = min(left, chunk_max, self.sync_every - since if sync is not None else left)
            since += k
            sizes.append(k)
            exchanges.append(sync is not None and since == self.sync_every)
            if exchanges[-1]:
                since = 0
            left -= k
        starts = np.cumsum([0] + sizes[:-1])
        logged = [self._since_sync if sync is not None else 0]  # steps in the current log, by chunk

        def exchange(k):
            logged[0] += sizes[k]
            if exchanges[k]:  # all-gather of the (cell, delta) records since the last exchange, stream-ordered
                sync.exchange(logged[0] * n)
                _lib.check(lib.qe_delta_log_reset(algo.handle))
                logged[0] = 0

        try:
            if learn and not collect_trace:
                # Pipelined: chunk k+1 is enqueued before the results of chunk k are read back, so the GPU
                # never waits for the host (schedule arithmetic, episode-log handling, replica exchange).
                eps = _schedule_values(self.exploration_rate_schedule, n, steps)
                lr = _schedule_values(self.lr_schedule, n, steps)
                one = len(sizes) == 1
                if not one:
                    # the schedule values of the whole call go to the device once (qe_schedule_plan); the
                    # chunks consume them in order.  (One launch: they travel with it, inside the kernel
                    # arguments when short.)
                    _lib.check(lib.qe_schedule_plan(algo.handle, _f64_ptr(eps), _f64_ptr(lr), steps))
                begun = []  # chunks enqueued and not yet collected

                def begin(k):
                    _lib.check(lib.qe_rollout_begin(algo.handle, env.handle, sizes[k], _f64_ptr(eps) if one else None,
                                                    _f64_ptr(lr) if one else None, mode, k & 1))
                    begun.append(k)

                def end(k):
                    st = _lib.RolloutStats()
                    begun.remove(k)
                    _lib.check(lib.qe_rollout_end(algo.handle, k & 1, C.byref(st)))
                    self._collect(lib, algo, st, int(starts[k]), total, history, ep_steps)

                try:
                    begin(0)
                    for k in range(1, len(sizes)):
                        exchange(k - 1)
                        begin(k)
                        end(k - 1)
                    exchange(len(sizes) - 1)
                    end(len(sizes) - 1)
                finally:
                    # an exception (e.g. the reference's IndexError for an agent without a selectable
                    # action, or a failed collective) must not leave a slot marked busy: drain what was
                    # begun, ignoring its status
                    for k in list(begun):
                        lib.qe_rollout_end(algo.handle, k & 1, None)
                    begun.clear()
            else:
                for k, size in enumerate(sizes):
                    st = _lib.RolloutStats()
                    if learn:
                        eps = _schedule_values(self.exploration_rate_schedule, n, size)
                        lr = _schedule_values(self.lr_schedule, n, size)
                        trace = np.empty((size, n), dtype=np.int32)
                        _lib.check(lib.qe_rollout(algo.handle, env.handle, size, _f64_ptr(eps), _f64_ptr(lr), mode,
                                                  _lib.ptr(trace, C.c_int32), C.byref(st)))
                        traces.append(trace)
                        exchange(k)
                    else:
                        _lib.check(lib.qe_evaluate(algo.handle, env.handle, size, C.byref(st)))
                    self._collect(lib, algo, st, int(starts[k]), total, history, ep_steps)
        finally:
            if sync is not None:
                self._since_sync = logged[0]
                if any(exchanges):
                    sync.flush()  # the exchange still in flight is completed (remote deltas applied) before returning
        total["kernel_variant"] = max(self._variants) if self._variants else 0
        total["kernel_variants"] = sorted(self._variants)
        self.last_stats = total
        if traces:
            self.last_trace = np.concatenate(traces)
            self.trace_actions = self.last_trace
        rets = (history[0] if len(history) == 1 else np.concatenate(history)) if history else np.empty(0, dtype=np.float32)
        at = (ep_steps[0] if len(ep_steps) == 1 else np.concatenate(ep_steps)) if ep_steps else np.empty(0, dtype=np.int32)
        return rets, at

    def _run_steps_fused(self, steps, env):
        """``run_steps`` body as ONE engine call (``qe_rollout_fused``): schedule values in, episode returns,
        their float32 sum and the resume state out.  For calls that fit one launch on an unmasked
        environment without replica exchange or action trace -- in particular the short calls of a
        step-by-step driver, whose cost is all per-call overhead."""
        lib, algo, n = _lib.load(), self.algorithm, env.num_agents
        env._resident = None  # the device state moves on
        mode = _lib.LEARN_ITER if self.learn_mode == "iter" else _lib.LEARN_VEC
        eps = _schedule_values(self.exploration_rate_schedule, n, steps)
        lr = _schedule_values(self.lr_schedule, n, steps)
        state = np.empty(3 * n, dtype=np.uint32)  # observations | env-internal state | running returns
        base = state.ctypes.data
        st = self._fused_stats
        cnt = lib.qe_rollout_fused(algo.handle, env.handle, steps, _f64_ptr(eps), _f64_ptr(lr), mode, C.byref(st),
                                   self._fused_cap, self._fused_step_p, self._fused_ret_p, C.byref(self._fused_sum),
                                   C.cast(base, C.POINTER(C.c_int32)), C.cast(base + 4 * n, C.POINTER(C.c_uint32)),
                                   C.cast(base + 8 * n, C.POINTER(C.c_float)))
        if cnt < 0:
            _lib.check(cnt)
        if st.episodes_dropped:
            msg = f"episode log overflow: {st.episodes_dropped} episode returns were dropped"
            raise _lib.EngineError(msg)
        if cnt <= self._fused_cap:
            rets = self._fused_ret[:cnt].copy()
        else:
            rets = np.empty(cnt, dtype=np.float32)
            lib.qe_episode_log(algo.handle, cnt, None, None, _lib.ptr(rets, C.c_float))
        self.last_stats = {f: getattr(st, f) for f, _ in st._fields_}
        self.last_stats["kernel_variants"] = [int(st.kernel_variant)]
        obs, aux, rewards = state[:n].view(np.int32), state[n:2 * n], state[2 * n:].view(np.float32)
        total = np.float32(self._fused_sum.value) if cnt else 0  # (0 / 0 -> ZeroDivisionError, like the reference)
        return rets, total, env.adopt_state(obs, rewards, aux)

    def run_steps(self, steps, env, curr_state_dict=None):
        if not isinstance(env, DeviceVecEnv):
            return self._run_steps_host(steps, env, curr_state_dict)
        env.bind(self.algorithm)
        if curr_state_dict is None:
            env.reset_device()
        elif not env.is_resident(curr_state_dict):
            # a state dict from elsewhere (another process, an edited copy): hand it to the device.  The
            # dict this runtime returned last is recognised and costs nothing -- the environments never
            # left the GPU.
            env.restore(curr_state_dict["states"], curr_state_dict["rewards"], curr_state_dict.get("aux"))
        collect_trace = self.trace_actions is not None and self.trace_actions is not False
        # (replicas: a call that does not cross the exchange cadence only appends to the delta log -- the kernels
        # write it either way -- and takes the same one-call path as a single GPU; if it ends exactly on the
        # cadence the exchange follows it)
        sync = self.delta_sync
        if (not collect_trace and not env.masked and 0 < steps <= env.chunk_limit(True) and self.history_type == "float"
                and (sync is None or self._since_sync + steps <= self.sync_every)):
            rets, total, state_dict = self._run_steps_fused(steps, env)
            if sync is not None:
                self._since_sync += steps
                if self._since_sync == self.sync_every:  # the call ended on the cadence: exchange now, as _rollout would
                    sync.exchange(self._since_sync * env.num_agents)
                    _lib.check(_lib.load().qe_delta_log_reset(self.algorithm.handle))
                    self._since_sync = 0
                    sync.flush()
            reward_history = rets.tolist()
        else:
            as_list = self.history_type == "float"
            rets, _ = self._rollout(env, steps, learn=True, as_list=as_list)
            if as_list:
                reward_history, total = self._as_list, (self._run_sum[0] if len(self._as_list) else 0)
                self._as_list = None
            else:
                reward_history = self._history(rets)
                total = _sequential_sum(rets)
            state_dict = env.state_dict()
        state_dict["episode_rewards"] = reward_history
        # not part of the reference's dict (single_thread_runtime.py:70-75): what an exact resume in a
        # fresh process needs besides the table -- draw counter and schedule values (restore_training_state)
        state_dict["rng_step"] = self.algorithm.step_counter
        state_dict["lr"] = self.lr_schedule.get_value()
        state_dict["exploration_rate"] = self.exploration_rate_schedule.get_value()
        return (
            # sum(reward_history) / len(reward_history) of the reference (:67): a sequential float32
            # accumulation (np.cumsum / the engine's running sum); ZeroDivisionError if no episode ended
            total / len(reward_history),
            reward_history,
            env,
            state_dict,
        )

    def restore_training_state(self, state_dict) -> None:
        """Continue a run in a fresh process exactly where ``state_dict`` (returned by ``run_steps``,
        e.g. un-pickled) left it: draw counter and both schedule values.  The table is restored with
        ``algorithm.load(filename)``, the environments by passing the dict to ``run_steps``."""
        self.algorithm.step_counter = int(state_dict["rng_step"])
        self.lr_schedule.set_value(state_dict["lr"])
        self.exploration_rate_schedule.set_value(state_dict["exploration_rate"])

    def _run_steps_host(self, steps, env, curr_state_dict):
        reward_history = []
        if curr_state_dict is None:
            states, infos = env.reset()
            agent_rewards = np.zeros(_count_agents(states), dtype=np.float32)
        else:
            states, infos = curr_state_dict["states"], curr_state_dict["infos"]
            agent_rewards = np.array(curr_state_dict["rewards"], dtype=np.float32)  # accumulated in place below
        for _ in range(steps):
            states, infos = self.run_single_step(env, states, agent_rewards, reward_history)
        return (
            sum(reward_history) / len(reward_history),
            reward_history,
            env,
            {"states": states, "infos": infos, "rewards": agent_rewards, "episode_rewards": reward_history},
        )

    # ------------------------------------------------------------------ evaluation on device
    def evaluate_steps(self, env, steps):
        if not isinstance(env, DeviceVecEnv):
            return super().evaluate_steps(env, steps)
        env.bind(self.algorithm)
        env.reset_device(seed=42)
        vector_steps = len(range(0, steps, env.num_agents))
        rets, _ = self._rollout(env, vector_steps, learn=False)
        history = list(rets)
        return sum(history), history

    def evaluate_episodes(self, env, episodes):
        if not isinstance(env, DeviceVecEnv):
            return super().evaluate_episodes(env, episodes)
        env.bind(self.algorithm)
        env.reset_device(seed=42)
        start = self.algorithm.step_counter
        history, used, chunk = [], 0, 64
        while len(history) < episodes:
            rets, at = self._rollout(env, chunk, learn=False)
            need = episodes - len(history)
            if rets.size >= need:
                # stop at the END of the vector step in which the target is reached: episodes ending
                # in that same step are still recorded (base_runtime.py:361-383)
                last_step = at[need - 1]
                keep = int(np.searchsorted(at, last_step, side="right"))
                history.extend(rets[:keep])
                used += int(last_step) + 1
                break
            history.extend(rets)
            used += chunk
            chunk = min(chunk * 2, 1 << 16)
        self.algorithm.step_counter = start + used
        return sum(history), history


# the name the reference's callers import
SingleThreadQLearning = GpuRolloutQLearning
