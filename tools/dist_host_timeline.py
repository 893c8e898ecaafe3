"""Diagnostic: host time spent in exchange / begin / end per chunk on the one-rank replica-sync path."""
import os, sys, time
sys.path.insert(0, ".")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
import numpy as np
PLAIN = bool(os.environ.get("PLAIN"))
if not PLAIN:
    import torch, torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    from dist_classicrl_amd.distributed.delta_sync import attach_engine, DeltaSync
from dist_classicrl_amd.algorithms.base_algorithms.q_learning_optimal import OptimalQLearningBase
from dist_classicrl_amd.algorithms.runtime import gpu_rollout_runtime as grr
from dist_classicrl_amd.environments import HashTabularEnv
from dist_classicrl_amd.schedules import ExponentialSchedule
from dist_classicrl_amd import _lib

n = 128
algo = OptimalQLearningBase(1_000_000, 16, 0.99, seed=0)
env = HashTabularEnv(n, 1_000_000, 16, seed=1)
rt = grr.GpuRolloutQLearning(algo, ExponentialSchedule(0.1, 1e-5, 0.995), ExponentialSchedule(1.0, 0.01, 0.995))
if not PLAIN:
    rt.sync_every = 100
    rt.delta_sync = attach_engine(algo, 100, n)
T0 = {}
T = {"exchange": 0.0, "begin": 0.0, "end": 0.0}
lib = _lib.load()
if not PLAIN:
    orig_ex = DeltaSync.exchange
    def timed_ex(self, count):
        t = time.perf_counter(); orig_ex(self, count); T["exchange"] += time.perf_counter() - t
    DeltaSync.exchange = timed_ex
ob, oe = lib.qe_rollout_begin, lib.qe_rollout_end
class W:
    def __init__(s, f, k): s.f, s.k = f, k
    def __call__(s, *a):
        t = time.perf_counter(); r = s.f(*a); T[s.k] += time.perf_counter() - t; return r
lib.qe_rollout_begin = W(ob, "begin"); lib.qe_rollout_end = W(oe, "end")
_, _, _, sd = rt.run_steps(2000, env, None)
for k in T: T[k] = 0.0
op = lib.qe_schedule_plan
lib.qe_schedule_plan = W(op, "plan"); T["plan"] = 0.0
t0 = time.perf_counter()
_, _, _, sd = rt.run_steps(20000, env, sd)
el = time.perf_counter() - t0
chunks = 200 if not PLAIN else 10
print(f"total {el*1e3:.1f} ms = {el/chunks*1e6:.1f} us/chunk; " + ", ".join(f"{k} {v/chunks*1e6:.1f} us/chunk" for k, v in T.items()),
      f"kernel {rt.last_stats['kernel_ms']/chunks*1e3:.1f} us/chunk")
if not PLAIN: dist.destroy_process_group()
