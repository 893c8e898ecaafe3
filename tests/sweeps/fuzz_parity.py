"""Randomised parity sweep on the GPU box (not part of the test suite): random shapes, dtypes, learn
modes, schedules and rollout paths, product vs the NumPy oracle, everything compared bit for bit.
Usage: python tests/sweeps/fuzz_parity.py <seconds> [seed] [path].  Prints every failing configuration.
`path` = a rollout path of tests/test_gpu_parity.py (traced runs), or persistent_df / persistent_full / persistent_sparse: plain
training rollouts of up to 128 agents WITHOUT an action trace through the dataflow kernel / the full build / the sparse build
(QE_OPT_LANE_ORDERED_PATH 1 / 2 / 3), cut into random run_steps calls -- the instantiations a user gets."""
import os, sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
np.seterr(all="ignore")
import pytest
from helpers import run_oracle_trace
import test_gpu_parity as tp

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
only_path = sys.argv[3] if len(sys.argv) > 3 else None  # e.g. "turnstile" / "turnstile_reread": every case through that path
UNTRACED = {"persistent_df": 1, "persistent_full": 2, "persistent_sparse": 3}


def run_untraced(spec, steps, dt, sched, mode, choice, rng):
    """The product WITHOUT an action trace, in random run_steps calls; same keys as _run_product_trace minus the actions."""
    from dist_classicrl_amd import _lib
    Algo, Runtime, _, _ = tp._product()
    env = tp.make_device_env(spec)
    algo = Algo(env.state_size, env.action_size, 0.99, seed=0, dtype=np.dtype(dt))
    algo.set_engine_option(_lib.OPT_LANE_ORDERED_PATH, choice)
    from helpers import schedule_params
    lr_p, eps_p = schedule_params(sched)
    rt = Runtime(algo, tp.make_schedule(lr_p), tp.make_schedule(eps_p), learn_mode=mode)
    sd, history, left = None, [], steps
    while left > 0:
        k = int(min(left, rng.integers(1, 40)))
        try:
            _avg, h, env, sd = rt.run_steps(k, env, sd)
        except ZeroDivisionError:
            h, sd = [], env.state_dict()
        d = _lib.decode_variant(rt.last_stats["kernel_variant"])
        assert d["path"] == "persistent" and d["lean"] == 1, d
        history += h
        left -= k
    obs, acc = env.observe()
    return {"q": np.asarray(algo.q_table), "history": np.array(history, dtype=np.float32),
            "final_obs": obs["observation"] if isinstance(obs, dict) else obs, "agent_rewards": acc,
            "final_sched": np.array([rt.lr_schedule.get_value(), rt.exploration_rate_schedule.get_value()])}


t_end = time.time() + budget
n_ok = n_bad = n_skip = 0
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
    if only_path in UNTRACED:  # the shapes those builds exist for
        kind = str(rng.choice(["hash", "hash", "hash", "ttt"]))
        agents = [64, 128] if only_path == "persistent_sparse" else [1, 37, 64, 90, 128]
        if kind == "hash":
            spec = ("hash", int(rng.choice(agents)), int(rng.choice([5, 40, 300, 2000, 20000, 100000, 3000000])), int(rng.choice([5, 8, 9, 12, 16])), False)
        else:
            spec = ("ttt", int(rng.choice(agents)))
    n_agents = spec[1]
    steps = int(min(rng.integers(3, 70), max(3, 150_000 // max(1, n_agents))))
    dt = str(rng.choice(["f4", "f8"]))
    mode = str(rng.choice(["iter", "iter", "vec"]))
    if only_path in UNTRACED:
        dt, mode = "f4", "iter"
    sched = str(rng.choice(["const", "bench", "linear"]))
    path = str(rng.choice(["auto", "stepwise", "persistent", "wide", "wide_listed", "turnstile", "turnstile"]))
    if only_path:
        path = only_path
    if path in ("stepwise", "wide", "wide_listed"):  # hashed touch counters: 2^bits slots (1 = one per row)
        os.environ["QE_TEST_STAMP_BITS"] = str(int(rng.choice([1, 2, 5, 9, 13])))
    else:
        os.environ.pop("QE_TEST_STAMP_BITS", None)
    cfg = (spec, steps, dt, mode, sched, path, os.environ.get("QE_TEST_STAMP_BITS"))
    try:
        try:
            want = run_oracle_trace(spec, steps, dt, sched, mode)
        except IndexError:  # diverged table (NaN maximum): the reference crashes here; so must the product
            try:
                if only_path in UNTRACED:
                    run_untraced(spec, steps, dt, sched, mode, UNTRACED[only_path], rng)
                else:
                    tp._run_product_trace(spec, steps, dt, sched, mode, path=path)
            except IndexError:
                n_ok += 1
            except pytest.skip.Exception:
                n_skip += 1
            else:
                n_bad += 1
                print("NO-ERROR", cfg, flush=True)
            continue
        if only_path in UNTRACED:
            got = run_untraced(spec, steps, dt, sched, mode, UNTRACED[only_path], rng)
            got["actions"] = want["actions"]  # (no trace: everything else must equal)
        else:
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
    else:
        n_bad += 1
        which = [k for k in ("actions", "q", "history", "final_obs", "agent_rewards", "final_sched") if not np.array_equal(got[k], want[k], equal_nan=(k == "q"))]
        print("MISMATCH", cfg, which, flush=True)
print(f"fuzz: {n_ok} ok, {n_bad} bad, {n_skip} skipped (tables compared with their NaNs; round 2's NaN-regime allowance is gone)")
