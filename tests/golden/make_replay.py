"""Golden fixture for the experience-replay ring, generated from the REAL reference class
(`algorithms/buffers/experience_replay.py`; run in the build container only):

    PYTHONPATH=/root/reference/src PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_replay.py

A scripted sequence of pushes (with wrap-around) interleaved with `sample(1)` calls; the fixture holds the
pushed experiences and, after every operation, position / full / len and the sampled tuple."""
import sys
from pathlib import Path

import numpy as np

from dist_classicrl.algorithms.buffers.experience_replay import ExperienceReplay

CAPACITY, SEED = 37, 11
rng = np.random.default_rng(123)
rb = ExperienceReplay(CAPACITY, SEED)
pushed, log = [], []
for op in range(160):
    if op < 5 or rng.random() < 0.7:
        exp = (int(rng.integers(1000)), int(rng.integers(9)), float(rng.random()), int(rng.integers(1000)),
               bool(rng.random() < 0.2))
        rb.push(exp)
        pushed.append(exp)
        log.append((0, rb.position, int(rb.full), len(rb), 0, 0, 0.0, 0, 0))
    else:
        s, a, r, n, d = rb.sample(1)
        log.append((1, rb.position, int(rb.full), len(rb), s, a, r, n, int(d)))
valid = CAPACITY if rb.full else rb.position
np.savez_compressed(
    Path(__file__).resolve().parent / "replay.npz",
    meta=np.array([CAPACITY, SEED]),
    pushed=np.array([(s, a, r, n, int(d)) for s, a, r, n, d in pushed], dtype=np.float64),
    log=np.array(log, dtype=np.float64),
    final_state=rb.state_buffer[:valid], final_action=rb.action_buffer[:valid], final_reward=rb.reward_buffer[:valid],
    final_next=rb.next_state_buffer[:valid], final_done=rb.done_buffer[:valid],
)
try:
    rb.sample(2)
    raised = "none"
except TypeError:
    raised = "TypeError"
print(len(pushed), "pushes,", sum(1 for e in log if e[0] == 1), "samples; sample(2) ->", raised)
