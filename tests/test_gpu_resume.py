"""GPU: checkpoint / resume across processes.

``algo.save`` writes the reference's ``.npy`` table (q_learning_optimal.py:252-261), ``run_steps`` returns
the reference's resume dict (single_thread_runtime.py:58-75) plus what an exact continuation needs (env
``aux``, draw counter, schedule values).  A run that is saved, re-loaded in a FRESH process and continued
must equal the run that simply continues, bit for bit: table, episode returns, final observations."""

import pickle
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]

COMMON = """
import pickle, sys
import numpy as np
sys.path.insert(0, {root!r})
from dist_classicrl_amd.algorithms.base_algorithms.q_learning_optimal import OptimalQLearningBase
from dist_classicrl_amd.algorithms.runtime.gpu_rollout_runtime import GpuRolloutQLearning
from dist_classicrl_amd.environments import HashTabularEnv, TicTacToeEnv, RiggedTwoArmedBanditVecEnv
from dist_classicrl_amd.schedules import ExponentialSchedule, LinearSchedule

def make(kind, dtype):
    if kind == "hash":
        env = HashTabularEnv(96, 3000, 16, seed=1)
    elif kind == "hash_many":  # above the persistent kernel's 512 agents: one launch per step (turnstile path)
        env = HashTabularEnv(700, 900, 16, seed=1)
    elif kind == "hash_masked":
        env = HashTabularEnv(64, 700, 12, seed=3, masked=True)
    elif kind == "ttt":
        env = TicTacToeEnv(64, seed=1)
    else:
        env = RiggedTwoArmedBanditVecEnv(8, episode_len=7)
    algo = OptimalQLearningBase(env.state_size, env.action_size, 0.95, seed=11, dtype=np.dtype(dtype))
    rt = GpuRolloutQLearning(algo, ExponentialSchedule(0.5, 1e-3, 0.9995), LinearSchedule(0.9, -2e-5))
    return algo, env, rt
"""

FIRST = COMMON + """
algo, env, rt = make({kind!r}, {dtype!r})
_, hist, _, sd = rt.run_steps({k1}, env, None)
algo.save({table!r})
sd = {{k: (v if not isinstance(v, np.ndarray) else np.array(v)) for k, v in sd.items()}}
pickle.dump((sd, hist), open({state!r}, "wb"))
"""

SECOND = COMMON + """
algo, env, rt = make({kind!r}, {dtype!r})
sd, _ = pickle.load(open({state!r}, "rb"))
algo.load({table!r})
rt.restore_training_state(sd)
_, hist, _, sd2 = rt.run_steps({k2}, env, sd)
obs = sd2["states"]["observation"] if isinstance(sd2["states"], dict) else sd2["states"]
pickle.dump((np.asarray(algo.q_table), hist, np.array(obs), np.array(sd2["rewards"]), sd2["rng_step"], sd2["lr"], sd2["exploration_rate"]),
            open({out!r}, "wb"))
"""

STRAIGHT = COMMON + """
algo, env, rt = make({kind!r}, {dtype!r})
_, hist1, _, sd = rt.run_steps({k1}, env, None)
_, hist, _, sd2 = rt.run_steps({k2}, env, sd)
obs = sd2["states"]["observation"] if isinstance(sd2["states"], dict) else sd2["states"]
pickle.dump((np.asarray(algo.q_table), hist, np.array(obs), np.array(sd2["rewards"]), sd2["rng_step"], sd2["lr"], sd2["exploration_rate"]),
            open({out!r}, "wb"))
"""


def _run(code):
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]


@pytest.mark.parametrize(("kind", "dtype", "k1", "k2"), [
    ("hash", "float32", 70, 90),          # persistent path, state dict crosses the process boundary
    ("hash_masked", "float64", 40, 40),   # masked observations (dict states), float64 table
    ("ttt", "float32", 60, 50),           # env-internal state = the boards
    ("bandit", "float64", 9, 12),         # env-internal state = position inside the episode
    ("hash_many", "float32", 30, 25),     # 700 agents on 900 states: every step has chains of row sharers
])
def test_save_load_resume_in_a_fresh_process_equals_continuing(tmp_path, kind, dtype, k1, k2):
    table, state = str(tmp_path / "table.npy"), str(tmp_path / "state.pkl")
    resumed, straight = str(tmp_path / "resumed.pkl"), str(tmp_path / "straight.pkl")
    fmt = dict(root=str(ROOT), kind=kind, dtype=dtype, k1=k1, k2=k2, table=table, state=state)
    _run(FIRST.format(**fmt))
    saved = np.load(table)  # the reference's format: a plain (S, A) .npy
    assert saved.dtype == np.dtype(dtype) and saved.ndim == 2
    _run(SECOND.format(out=resumed, **fmt))
    _run(STRAIGHT.format(out=straight, **fmt))
    got, want = pickle.load(open(resumed, "rb")), pickle.load(open(straight, "rb"))
    assert np.array_equal(got[0], want[0]), "table after the resumed run differs"
    assert got[1] == want[1], "episode returns of the resumed run differ"
    assert np.array_equal(got[2], want[2]) and np.array_equal(got[3], want[3])
    assert got[4:] == want[4:]  # draw counter and schedule values
    assert len(want[1]) > 0 and np.count_nonzero(want[0]) > 0
