"""Randomised parity sweep on the GPU box (not part of the test suite): random shapes, dtypes, learn
modes, schedules and rollout paths, product vs the NumPy oracle, everything compared bit for bit.
Usage: python tools/fuzz_parity.py <seconds> [seed] [path].  Prints every failing configuration."""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
np.seterr(all="ignore")
import pytest
from helpers import run_oracle_trace
import test_gpu_parity as tp

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
only_path = sys.argv[3] if len(sys.argv) > 3 else None  # e.g. "turnstile" / "turnstile_reread": every case through that path
t_end = time.time() + budget
n_ok = n_bad = n_skip = n_div = 0
t_note = time.time()
while time.time() < t_end:
    if time.time() - t_note > 30:
        t_note = time.time()
        print(f"... {n_ok} ok, {n_bad} bad, {n_skip} skipped", flush=True)
    kind = rng.choice(["hash", "hash", "hash", "hash", "grid", "bandit", "ttt"])
    if kind == "hash":
        A = int(rng.choice([1, 2, 3, 4, 5, 8, 9, 12, 16, 17, 32, 33, 64, 70]))
        n = int(rng.choice([1, 2, 7, 64, 128, 130, 300, 512, 513, 1024, 2048, 2500, 4096, 6000]))
        S = int(rng.choice([1, 2, 5, 40, 300, 2000, 20000, 100000]))
        spec = ("hash", n, S, A, bool(rng.random() < 0.4))
    elif kind == "grid":
        spec = ("grid", int(rng.choice([1, 16, 200, 700])), int(rng.choice([4, 6, 10])))
    elif kind == "bandit":
        spec = ("bandit", int(rng.choice([1, 2, 300, 600])), int(rng.choice([1, 4, 7])))
    else:
        spec = ("ttt", int(rng.choice([1, 64, 128, 600])))
    if only_path == "persistent_light":  # the shapes that build exists for
        if kind == "hash":
            spec = ("hash", int(rng.choice([64, 128])), int(rng.choice([5, 40, 300, 2000, 20000, 100000])), int(rng.choice([5, 8, 9, 12, 16])), False)
        elif kind == "ttt":
            spec = ("ttt", int(rng.choice([64, 128])))
    n_agents = spec[1]
    steps = int(min(rng.integers(3, 70), max(3, 150_000 // max(1, n_agents))))
    dt = str(rng.choice(["f4", "f8"]))
    mode = str(rng.choice(["iter", "iter", "vec"]))
    if only_path == "persistent_light":
        dt, mode = "f4", "iter"
    sched = str(rng.choice(["const", "bench", "linear"]))
    path = str(rng.choice(["auto", "stepwise", "persistent", "wide", "wide_listed", "turnstile", "turnstile"]))
    if only_path:
        path = only_path
    cfg = (spec, steps, dt, mode, sched, path)
    try:
        try:
            want = run_oracle_trace(spec, steps, dt, sched, mode)
        except IndexError:  # diverged table (NaN maximum): the reference crashes here; so must the product
            try:
                tp._run_product_trace(spec, steps, dt, sched, mode, path=path)
            except IndexError:
                n_ok += 1
            except pytest.skip.Exception:
                n_skip += 1
            else:
                n_bad += 1
                print("NO-ERROR", cfg, flush=True)
            continue
        got = tp._run_product_trace(spec, steps, dt, sched, mode, path=path)
    except pytest.skip.Exception:
        n_skip += 1
        continue
    except Exception as ex:  # noqa: BLE001
        if "persistent rollout needs" in str(ex):
            n_skip += 1
            continue
        n_bad += 1
        print("ERROR", cfg, type(ex).__name__, ex, flush=True)
        continue
    ok = all(np.array_equal(got[k], want[k], equal_nan=(k == "q")) for k in ("actions", "q", "history", "final_obs", "agent_rewards", "final_sched"))
    if ok:
        n_ok += 1
    elif not np.isfinite(want["q"]).all() and all(np.array_equal(got[k], want[k]) for k in ("actions", "history", "final_obs")):
        # the known divergence: the oracle's table has overflowed to inf / NaN (lr = 1, learn_vec,
        # hundreds of colliding increments); NumPy's max propagates NaN, the kernels' does not
        n_div += 1
    else:
        n_bad += 1
        which = [k for k in ("actions", "q", "history", "final_obs", "agent_rewards", "final_sched") if not np.array_equal(got[k], want[k], equal_nan=(k == "q"))]
        print("MISMATCH", cfg, which, flush=True)
print(f"fuzz: {n_ok} ok, {n_bad} bad, {n_skip} skipped, {n_div} in the known NaN regime (tables differ, actions equal)")
