import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
np.seterr(all="ignore")
from helpers import run_oracle_trace
import test_gpu_parity as tp
cfgs = [(('bandit', 600, 4), 43, 'f4', 'vec', 'linear', 'auto'), (('hash', 128, 1, 12, True), 51, 'f4', 'vec', 'linear', 'persistent')]
for spec, steps, dt, mode, sched, path in cfgs:
    prev_want = None
    for st in range(1, steps + 1):
        want = run_oracle_trace(spec, st, dt, sched, mode)
        got = tp._run_product_trace(spec, st, dt, sched, mode, path=path)
        if not np.array_equal(got["q"], want["q"], equal_nan=True):
            bad = np.argwhere(~((got["q"] == want["q"]) | (np.isnan(got["q"]) & np.isnan(want["q"]))))
            print(spec, "first differing step", st, "cells", len(bad))
            for r, c in bad[:6]:
                print("   cell", r, c, "got", repr(got["q"][r, c]), "want", repr(want["q"][r, c]), "before", repr(prev_want["q"][r, c]) if prev_want else None)
            print("   table before:", prev_want["q"][:2] if prev_want else None)
            break
        prev_want = want
    else:
        print(spec, "no mismatch with equal_nan")
