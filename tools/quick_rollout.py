"""Diagnostic: time a rollout for a given shape / epsilon and print device time per step."""
import os, sys, time
sys.path.insert(0, ".")
import numpy as np
from dist_classicrl_amd.algorithms.base_algorithms.q_learning_optimal import OptimalQLearningBase
from dist_classicrl_amd.algorithms.runtime.gpu_rollout_runtime import GpuRolloutQLearning
from dist_classicrl_amd.environments import HashTabularEnv
from dist_classicrl_amd.schedules import ConstantSchedule, ExponentialSchedule

def run(n, S, A, eps, steps, path="auto", masked=False, mode="iter"):
    algo = OptimalQLearningBase(S, A, 0.99, seed=0)
    algo.set_rollout_path(path)
    from dist_classicrl_amd import _lib
    if os.environ.get("QE_NO_GRAPH"):
        _lib.check(_lib.load().qe_set_option(algo.handle, 1, 0))
    if os.environ.get("QE_LISTED_MIN"):
        _lib.check(_lib.load().qe_set_option(algo.handle, 3, int(os.environ["QE_LISTED_MIN"])))
    if os.environ.get("QE_ROUNDS"):
        _lib.check(_lib.load().qe_set_option(algo.handle, 2, int(os.environ["QE_ROUNDS"])))
    if os.environ.get("QE_TTT"):
        from dist_classicrl_amd.environments import TicTacToeEnv
        algo = OptimalQLearningBase(19683, 9, 0.99, seed=0)
        env = TicTacToeEnv(n, seed=1)
    else:
        env = HashTabularEnv(n, S, A, seed=1, masked=masked)
    e = ConstantSchedule(eps) if eps is not None else ExponentialSchedule(1.0, 0.01, 0.995)
    lr = ExponentialSchedule(0.1, 1e-5, 0.995) if os.environ.get("QE_BENCH_LR") else ConstantSchedule(0.1)
    rt = GpuRolloutQLearning(algo, lr, e, learn_mode=mode)
    _, _, _, sd = rt.run_steps(steps // 4 + 1, env, None)
    t0 = time.perf_counter()
    _, h, _, sd = rt.run_steps(steps, env, sd)
    dt = time.perf_counter() - t0
    st = rt.last_stats
    print(f"n={n} S={S} A={A} eps={eps} path={path} mode={mode}: device {st['kernel_ms']/steps*1e3:.2f} us/step, "
          f"wall {dt/steps*1e6:.2f} us/step, {n*steps/dt/1e6:.1f} M env-steps/s, involved/step {st['involved']/steps:.3f}, launches {st['launches']}")

if __name__ == "__main__":
    for args in [a.split(",") for a in sys.argv[1:]]:
        n, S, A = int(args[0]), int(args[1]), int(args[2])
        eps = None if args[3] == "sched" else float(args[3])
        steps = int(args[4])
        path = args[5] if len(args) > 5 else "auto"
        mode = args[6] if len(args) > 6 else "iter"
        run(n, S, A, eps, steps, path, mode=mode)
