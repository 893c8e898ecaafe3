"""Randomised parity sweep of `choose_actions` and its variants against the NumPy oracle (draws injected
with the counter-based protocol) on the GPU box: table shapes, ties, masks, epsilon, dtypes, batch sizes
on both sides of the reference dispatcher's thresholds."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
np.seterr(all="ignore")
from dist_classicrl_amd.algorithms.base_algorithms.q_learning_optimal import OptimalQLearningBase as Algo
from oracle.draws import InjectedDraws
from oracle.qlearn_oracle import OracleQLearning

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t_end, ok, bad = time.time() + budget, 0, 0
METHODS = ["choose_actions", "choose_actions", "choose_actions_iter", "choose_actions_vec_iter", "choose_actions_vec",
           "choose_masked_actions_vec"]
while time.time() < t_end:
    S = int(rng.choice([1, 3, 50, 2000]))
    A = int(rng.choice([1, 2, 3, 4, 5, 9, 10, 11, 16, 31, 64, 100, 257, 300]))
    n = int(rng.choice([1, 2, 9, 10, 11, 64, 99, 100, 101, 500, 3000]))
    dt = str(rng.choice(["f4", "f8"]))
    method = str(rng.choice(METHODS))
    masked = method == "choose_masked_actions_vec" or (method not in ("choose_actions_vec",) and rng.random() < 0.4)
    det = bool(rng.random() < 0.3)
    eps = float(rng.choice([0.0, 0.05, 0.3, 1.0]))
    seed, step = int(rng.integers(1 << 30)), int(rng.integers(1 << 20))
    # few distinct values -> many ties
    q0 = rng.integers(0, int(rng.choice([2, 3, 50])), size=(S, A)).astype(dt)
    states = rng.integers(S, size=n).astype(np.int32)
    masks = None
    if masked:
        masks = (rng.random((n, A)) < 0.5).astype(np.int32)
        if len(sys.argv) > 3 and sys.argv[3] == "empty":  # some agents without any valid action
            masks[rng.random(n) < 0.9, rng.integers(A)] = 1
            masks[rng.random(n) < 0.1] = 0
        else:
            masks[np.arange(n), rng.integers(A, size=n)] = 1
    algo, ref = Algo(S, A, 0.9, seed=seed, dtype=np.dtype(dt)), OracleQLearning(S, A, 0.9, dtype=np.dtype(dt))
    algo.q_table = q0
    algo.step_counter = step
    ref.q_table = q0.copy()
    ref._rng = ref._np_rng = shim = InjectedDraws(seed)
    shim.begin(step, n, eps, deterministic=det)
    def call(obj):
        f = getattr(obj, method)
        if method == "choose_actions_vec":
            return f(states, eps, deterministic=det)
        if method == "choose_masked_actions_vec":
            return f(states, masks, eps, deterministic=det)
        return f(states, eps, deterministic=det, action_masks=masks)
    try:
        want = call(ref)
        werr = None
    except Exception as ex:  # noqa: BLE001
        want, werr = None, type(ex)
    try:
        got = call(algo)
        gerr = None
    except Exception as ex:  # noqa: BLE001
        got, gerr = None, type(ex)
    if werr is gerr and (werr is not None or np.array_equal(got, want)):
        ok += 1
    else:
        bad += 1
        print("MISMATCH", (S, A, n, dt, method, masked, det, eps), werr, gerr, flush=True)
print(f"fuzz_select: {ok} ok, {bad} bad")
