"""Golden fixture for agents WITHOUT a valid action (mask all zeros), generated from the REAL reference
(run in the build container only):

    PYTHONPATH=/root/reference/src PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_empty_mask.py

For every selection variant the reference's result on a small batch whose agent 0 has an all-zero mask:
the action array, or the exception it raised.  Draws are injected with oracle/draws.py.  Deterministic
calls of the Python-list variants are left out: there the reference returns -1 without drawing, which
the injection shim cannot observe on the unmodified reference (see InjectedDraws.skip_choice)."""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from dist_classicrl.algorithms.base_algorithms.q_learning_optimal import OptimalQLearningBase  # noqa: E402

from oracle.draws import InjectedDraws  # noqa: E402

S, A, SEED, STEP = 6, 5, 7, 3
Q = np.arange(S * A, dtype=np.float64).reshape(S, A) % 7
LIST_VARIANTS = {"choose_actions_iter"}
out = {"q": Q, "meta": np.array([S, A, SEED, STEP])}
cases = []
for n in (1, 3, 40, 300):
    for det in (0, 1):
        for eps in (0.0, 0.5, 1.0):
            for method in ("choose_actions", "choose_actions_iter", "choose_actions_vec_iter", "choose_masked_actions_vec"):
                algo = OptimalQLearningBase(S, A, 0.9)
                algo.q_table = Q.copy()
                # which variant `choose_actions` dispatches to (q_learning_optimal.py:644-726)
                lists = method in LIST_VARIANTS or (method == "choose_actions" and A <= 10)
                if det and lists:
                    continue
                algo._rng = algo._np_rng = shim = InjectedDraws(SEED)
                shim.begin(STEP, n, eps, deterministic=bool(det))
                states = (np.arange(n) % S).astype(np.int32)
                masks = np.ones((n, A), dtype=np.int32)
                masks[0] = 0
                try:
                    if method == "choose_masked_actions_vec":
                        res = getattr(algo, method)(states, masks, eps, deterministic=bool(det))
                    else:
                        res = getattr(algo, method)(states, eps, deterministic=bool(det), action_masks=masks)
                    res, err = np.asarray(res, dtype=np.int32), 0
                except IndexError:
                    res, err = np.zeros(0, dtype=np.int32), 1
                k = len(cases)
                cases.append((n, det, eps, method))
                out[f"c{k}_cfg"] = np.array([n, det, int(eps * 100)])
                out[f"c{k}_method"] = np.array(method)
                out[f"c{k}_actions"] = res
                out[f"c{k}_raised"] = np.array(err)
out["count"] = np.array(len(cases))
np.savez_compressed(Path(__file__).resolve().parent / "empty_mask.npz", **out)
print(len(cases), "cases")
