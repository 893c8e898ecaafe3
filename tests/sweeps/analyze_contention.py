"""Diagnostic (CPU, uses the C oracle): shape of the per-step dependency graph of learn_iter at large agent counts.

For one vector step after `warm` steps of the given workload it reports how many agents touch a row
that another agent writes (the engine's "involved" set), how those agents group into connected
components (agents linked through a shared row), and the length of the longest dependency chain
under the true-dependency rules the ordered path uses.
"""
import sys
from collections import defaultdict
sys.path.insert(0, ".")
import numpy as np
from oracle.c_oracle import CHashRollout, exp_schedule

n, S, A, warm = (int(v) for v in sys.argv[1:5])
run = CHashRollout(n, S, A, dtype=np.float32)
eps, _ = exp_schedule(1.0, 0.01, 0.995, n, warm)
lr, _ = exp_schedule(0.1, 1e-5, 0.995, n, warm)
run.run(eps, lr, log_episodes=False)
for rep in range(3):
    s = run.obs.copy()
    ep0 = run.episode.copy()
    out = run.run(eps[-1:], lr[-1:], trace=True, log_episodes=False)
    a = out["actions"][0]
    nxt = run.obs.copy()
    term = run.episode != ep0
    # touches: W(row s) always, R(row nxt) unless terminated.  (The engine also reads nxt for select(t+1).)
    writers, readers = defaultdict(list), defaultdict(list)
    for i in range(n):
        writers[int(s[i])].append(i)
        readers[int(nxt[i])].append(i)
    shared_written = {r for r, w in writers.items() if len(w) + len(readers.get(r, ())) > 1}
    involved = np.array([int(s[i]) in shared_written or int(nxt[i]) in shared_written for i in range(n)])
    idx = np.flatnonzero(involved)
    # union-find over rows
    parent = {}
    def find(x):
        while parent.setdefault(x, x) != x:
            parent[x] = parent[parent[x]]
            x = parent[x]
        return x
    for i in idx:
        ra, rb = find(("r", int(s[i]))), find(("r", int(nxt[i])))
        if ra != rb:
            parent[ra] = rb
    comp = defaultdict(int)
    for i in idx:
        comp[find(("r", int(s[i])))] += 1
    sizes = np.array(sorted(comp.values(), reverse=True))
    # longest chain: depth[i] = 1 + max depth of the agents i must follow
    depth = {}
    last_w_depth = defaultdict(int)      # per row: max depth among writers so far
    last_r_depth = defaultdict(int)      # per row: max depth among readers so far
    last_cell_depth = defaultdict(int)
    same_cell_run = 0
    for i in idx:
        rs, rn, cell = int(s[i]), int(nxt[i]), (int(s[i]), int(a[i]))
        d = max(last_r_depth[rs], last_cell_depth[cell], 0 if term[i] else last_w_depth[rn]) + 1
        depth[i] = d
        last_w_depth[rs] = max(last_w_depth[rs], d)
        last_cell_depth[cell] = d
        if not term[i]:
            last_r_depth[rn] = max(last_r_depth[rn], d)
    dmax = max(depth.values()) if depth else 0
    cells = defaultdict(int)
    for i in idx:
        cells[(int(s[i]), int(a[i]))] += 1
    cs = np.array(sorted(cells.values(), reverse=True))
    print(f"step {run.step}: involved {len(idx)} of {n}; rows {len({int(s[i]) for i in idx} | {int(nxt[i]) for i in idx})}; "
          f"components {len(sizes)} (largest {sizes[:5].tolist()}); longest chain {dmax}; "
          f"distinct cells {len(cs)} (largest {cs[:5].tolist()}); agents in cells with >1 writer {int(cs[cs > 1].sum())}")
    # ---- how many agents are still pending after r rounds under three scheduling rules
    si, ni, ti = s[idx].astype(np.int64), nxt[idx].astype(np.int64), term[idx]
    INF = np.iinfo(np.int64).max
    def simulate(rule, rounds=40):
        pend = np.ones(len(idx), dtype=bool)
        left = []
        for _ in range(rounds):
            p = np.flatnonzero(pend)
            if len(p) == 0:
                break
            wmin, rmin = defaultdict(lambda: INF), defaultdict(lambda: INF)
            for k in p:
                i = int(idx[k])
                wmin[si[k]] = min(wmin[si[k]], i)
                if not ti[k]:
                    rmin[ni[k]] = min(rmin[ni[k]], i)
            for k in p:
                i = int(idx[k])
                if rule == "one":   # one token per row: lowest pending toucher of every row
                    ok = min(wmin[si[k]], rmin[si[k]]) == i and (ti[k] or min(wmin[ni[k]], rmin[ni[k]]) == i)
                else:               # reader / writer tokens
                    ok = wmin[si[k]] == i and rmin[si[k]] >= i and (ti[k] or wmin[ni[k]] >= i)
                if ok:
                    pend[k] = False
            left.append(int(pend.sum()))
        return left
    print("  one token per row :", simulate("one")[:32])
    print("  reader/writer toks:", simulate("rw")[:32])
    hist = np.bincount(np.array(list(depth.values())))
    print("  true-dependency depth: agents left after round r:", (len(idx) - np.cumsum(hist)[1:]).tolist()[:32])
