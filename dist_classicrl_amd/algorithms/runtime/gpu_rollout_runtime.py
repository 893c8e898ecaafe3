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

_EP_LOG_CAPACITY = 1 << 22  # entries held by the engine's episode log (csrc/qe_engine.hip)


def _schedule_values(schedule, n_updates, count):
    fast = getattr(schedule, "advance_values", None)
    if fast is not None:
        return np.ascontiguousarray(fast(n_updates, count), dtype=np.float64)
    out = np.empty(count, dtype=np.float64)  # duck-typed schedule (e.g. the reference's classes)
    for t in range(count):
        out[t] = schedule.get_value()
        schedule.update(n_updates)
    return out


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
        self.delta_sync = None  # dist_classicrl_amd.distributed.DeltaSync (multi-GPU replicas)
        self.sync_every = 100
        self.trace_actions = None  # set to True to collect every action (tests)
        # element type of the reward history `run_steps` returns: "float" (list of Python floats, the
        # float32 returns widened exactly), "float32" (list of numpy.float32 scalars, the reference's
        # element type, ~2x slower to build) or "array" (the float32 array itself: no per-episode
        # Python object; for runs that finish millions of episodes per call)
        self.history_type = "float"

    def _history(self, rets: np.ndarray):
        if self.history_type == "array":
            return rets
        return list(rets) if self.history_type == "float32" else rets.tolist()

    def init_training(self) -> None:
        return None

    def close_training(self) -> None:
        return None

    def _prepare_env(self, env):
        if isinstance(env, DeviceVecEnv):
            env.bind(self.algorithm)
        return env

    # ------------------------------------------------------------------ fused rollout
    _PIPELINE_CHUNK = 2000  # vector steps per launch when nothing else (log capacity, sync cadence) binds

    def _collect(self, lib, algo, st, done, total, history, ep_steps):
        cnt = int(lib.qe_episode_log(algo.handle, 0, None, None, None))
        if cnt:
            step_idx = np.empty(cnt, dtype=np.int32)
            ret = np.empty(cnt, dtype=np.float32)
            lib.qe_episode_log(algo.handle, cnt, _lib.ptr(step_idx, C.c_int32), None, _lib.ptr(ret, C.c_float))
            history.append(ret)
            ep_steps.append(step_idx + done)
        for f in total:
            total[f] += getattr(st, f)

    def _rollout(self, env, steps, learn):
        lib = _lib.load()
        algo = self.algorithm
        n = env.num_agents
        mode = _lib.LEARN_ITER if self.learn_mode == "iter" else _lib.LEARN_VEC
        total = {"kernel_ms": 0.0, "launches": 0, "episodes": 0, "involved": 0, "episodes_dropped": 0,
                 "dominant_ms": 0.0, "dominant_launches": 0, "dominant_env_steps": 0}
        history, ep_steps, traces = [], [], []
        chunk_max = max(1, _EP_LOG_CAPACITY // n)
        sync = self.delta_sync if learn else None
        if sync is not None:
            chunk_max = min(chunk_max, self.sync_every)
        if learn and not self.trace_actions:
            # Pipelined: chunk k+1 is enqueued before the results of chunk k are read back, so the GPU
            # never waits for the host (schedule arithmetic, episode-log handling, replica exchange).
            chunk_max = min(chunk_max, self._PIPELINE_CHUNK)
            sizes = [min(chunk_max, steps - d) for d in range(0, steps, chunk_max)]
            starts = np.cumsum([0] + sizes[:-1])

            # the schedule values of the whole call go to the device once (qe_schedule_plan); the chunks
            # consume them in order
            eps = _schedule_values(self.exploration_rate_schedule, n, steps)
            lr = _schedule_values(self.lr_schedule, n, steps)
            _lib.check(lib.qe_schedule_plan(algo.handle, _lib.ptr(eps, C.c_double), _lib.ptr(lr, C.c_double), steps))

            def begin(k):
                _lib.check(lib.qe_rollout_begin(algo.handle, env.handle, sizes[k], None, None, mode, k & 1))

            def end(k):
                st = _lib.RolloutStats()
                _lib.check(lib.qe_rollout_end(algo.handle, k & 1, C.byref(st)))
                self._collect(lib, algo, st, int(starts[k]), total, history, ep_steps)

            def exchange(k):
                if sync is not None:  # all-gather of chunk k's (cell, delta) records, stream-ordered
                    sync.exchange(sizes[k] * n)
                    _lib.check(lib.qe_delta_log_reset(algo.handle))

            begin(0)
            for k in range(1, len(sizes)):
                exchange(k - 1)
                begin(k)
                end(k - 1)
            exchange(len(sizes) - 1)
            end(len(sizes) - 1)
            if sync is not None:
                sync.flush()  # remote deltas still in flight are applied before returning
        else:
            done = 0
            while done < steps:
                k = min(chunk_max, steps - done)
                st = _lib.RolloutStats()
                if learn:
                    eps = _schedule_values(self.exploration_rate_schedule, n, k)
                    lr = _schedule_values(self.lr_schedule, n, k)
                    trace = np.empty((k, n), dtype=np.int32)
                    _lib.check(lib.qe_rollout(algo.handle, env.handle, k, _lib.ptr(eps, C.c_double),
                                              _lib.ptr(lr, C.c_double), mode, _lib.ptr(trace, C.c_int32),
                                              C.byref(st)))
                    traces.append(trace)
                    if sync is not None:
                        sync.exchange(k * n)
                        _lib.check(lib.qe_delta_log_reset(algo.handle))
                else:
                    _lib.check(lib.qe_evaluate(algo.handle, env.handle, k, C.byref(st)))
                self._collect(lib, algo, st, done, total, history, ep_steps)
                done += k
            if sync is not None:
                sync.flush()
        self.last_stats = total
        if traces:
            self.trace_actions = np.concatenate(traces)
        rets = np.concatenate(history) if history else np.empty(0, dtype=np.float32)
        at = np.concatenate(ep_steps) if ep_steps else np.empty(0, dtype=np.int32)
        return rets, at

    def run_steps(self, steps, env, curr_state_dict=None):
        if not isinstance(env, DeviceVecEnv):
            return self._run_steps_host(steps, env, curr_state_dict)
        env.bind(self.algorithm)
        if curr_state_dict is None:
            env.reset()
        else:
            env.restore(curr_state_dict["states"], curr_state_dict["rewards"])
        rets, _ = self._rollout(env, steps, learn=True)
        reward_history = self._history(rets)
        states, agent_rewards = env.observe()
        return (
            # sum(reward_history) / len(reward_history) of the reference (:67): a sequential float32
            # accumulation, which is what cumsum computes; ZeroDivisionError if no episode ended
            _sequential_sum(rets) / len(reward_history),
            reward_history,
            env,
            {"states": states, "infos": [{}] * env.num_agents, "rewards": agent_rewards,
             "episode_rewards": reward_history},
        )

    def _run_steps_host(self, steps, env, curr_state_dict):
        reward_history = []
        if curr_state_dict is None:
            states, infos = env.reset()
            agent_rewards = np.zeros(_count_agents(states), dtype=np.float32)
        else:
            states, infos = curr_state_dict["states"], curr_state_dict["infos"]
            agent_rewards = curr_state_dict["rewards"]
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
        env.reset(seed=42)
        vector_steps = len(range(0, steps, env.num_agents))
        rets, _ = self._rollout(env, vector_steps, learn=False)
        history = list(rets)
        return sum(history), history

    def evaluate_episodes(self, env, episodes):
        if not isinstance(env, DeviceVecEnv):
            return super().evaluate_episodes(env, episodes)
        env.bind(self.algorithm)
        env.reset(seed=42)
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
