"""Diagnostic: cost of applying one exchange's remote records at a BASELINE config-4 shard
(8192 agents x 100 steps per rank): the engine's own radix sort + apply, round 2's torch stable sort +
qe_delta_apply_sorted_dev, and the atomic form.

    python tools/time_delta_apply.py [ranks=8] [agents=8192] [steps=100] [states=10000000] [actions=32]
"""
import ctypes as C
import sys
import time

sys.path.insert(0, ".")
import torch

from dist_classicrl_amd import _lib
from dist_classicrl_amd.algorithms.base_algorithms.q_learning_optimal import OptimalQLearningBase

ranks, agents, steps, S, A = (int(x) for x in (sys.argv[1:6] + ["8", "8192", "100", "10000000", "32"][len(sys.argv) - 1:]))
algo = OptimalQLearningBase(S, A, 0.99, seed=0)
lib = _lib.load()
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(device=dev)
_lib.check(lib.qe_set_stream(algo.handle, C.c_void_p(stream.cuda_stream)))
count = (ranks - 1) * agents * steps
g = torch.Generator(device=dev).manual_seed(1)
ld = int(algo.q_table.shape[1]) if hasattr(algo, "q_table") else A
cells = torch.randint(0, S * A, (count,), device=dev, generator=g, dtype=torch.int64).to(torch.int32)
rec = torch.stack([cells, torch.zeros(count, dtype=torch.float32, device=dev).view(torch.int32)], dim=1).contiguous()
gathered = torch.zeros((ranks, agents * steps, 2), dtype=torch.int32, device=dev)
gathered[1:] = rec.view(ranks - 1, agents * steps, 2)  # (this rank = 0: its own segment is skipped)
with torch.cuda.stream(stream):
    for rep in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _lib.check(lib.qe_delta_apply_gathered_dev(algo.handle, C.c_void_p(gathered.data_ptr()), agents * steps, agents * steps, ranks, 0))
        torch.cuda.synchronize()
        t2 = time.perf_counter()
    print(f"in-engine radix sort + apply (qe_delta_apply_gathered_dev): {count} records from {ranks - 1} ranks: {1e3 * (t2 - t0):.2f} ms")
    for name in ("sorted", "atomic"):
        for rep in range(4):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            if name == "sorted":
                order = torch.sort(rec[:, 0], stable=True).indices
                srt = rec[order].contiguous()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                _lib.check(lib.qe_delta_apply_sorted_dev(algo.handle, C.c_void_p(srt.data_ptr()), count))
            else:
                t1 = t0
                _lib.check(lib.qe_delta_apply_dev(algo.handle, C.c_void_p(rec.data_ptr()), count))
            torch.cuda.synchronize()
            t2 = time.perf_counter()
        print(f"{name}: {count} records from {ranks - 1} ranks: sort+gather {1e3 * (t1 - t0):.2f} ms, apply {1e3 * (t2 - t1):.2f} ms")
