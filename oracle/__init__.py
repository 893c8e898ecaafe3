"""TEST INFRASTRUCTURE ONLY.

CPU restatement (the *oracle*) of the dist_classicrl hot path: epsilon-greedy (masked) argmax
selection, TD update (sequential ``learn_iter`` and batch ``learn_vec`` semantics), the
``run_single_step`` loop body, schedules and the synthetic tabular environments.

Nothing under ``oracle/`` is product code.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it, and only as the checker / the reported CPU
baseline.  The product path (``dist_classicrl_amd``) never imports this package and fails loudly
when its HIP library is missing.

Parity pin: the oracle is checked against (a) every known-answer case the reference's own tests hold
for this path (``tests/test_oracle_reference_kat.py``) and (b) golden vectors produced by importing
the real reference in the build container (``tests/golden/make_golden.py`` ->
``tests/golden/*.npz``, checked by ``tests/test_oracle_golden.py``).
"""
