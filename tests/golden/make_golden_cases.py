"""Case tables shared by ``make_golden.py`` (generator, needs the reference) and the tests."""

SELECT_CASES = [
    # (method, S, A, n, masked, eps, deterministic, dtype)
    ("choose_actions", 100, 4, 1, False, 0.3, False, "f8"),  # -> choose_actions_iter (C1 train)
    ("choose_actions", 1000, 8, 128, False, 0.1, False, "f8"),  # -> choose_actions_vec (C2 train)
    ("choose_actions", 1000, 16, 300, False, 0.5, False, "f4"),  # -> choose_actions_vec (C3 train)
    ("choose_actions", 200, 128, 150, False, 0.2, False, "f8"),  # -> choose_actions_vec_iter
    ("choose_actions", 300, 9, 64, True, 0.3, False, "f8"),  # -> masked iter (TicTacToe shape)
    ("choose_actions", 500, 64, 200, True, 0.25, False, "f4"),  # -> masked vec_iter (C5 train)
    ("choose_actions", 100, 8, 50, False, 0.0, True, "f8"),  # eval -> iter
    ("choose_actions", 100, 16, 70, False, 0.0, True, "f4"),  # eval -> choose_actions_vec
    ("choose_actions", 100, 64, 70, True, 0.0, True, "f8"),  # eval -> choose_masked_actions_vec
    ("choose_actions_iter", 50, 6, 40, False, 0.4, False, "f8"),
    ("choose_actions_iter", 50, 6, 40, True, 0.4, False, "f8"),
    ("choose_actions_vec_iter", 50, 12, 40, False, 0.4, False, "f8"),
    ("choose_actions_vec_iter", 50, 12, 40, True, 0.4, False, "f4"),
    ("choose_actions_vec", 50, 12, 40, False, 0.4, False, "f8"),
    ("choose_actions_vec", 50, 5, 33, False, 1.0, False, "f8"),  # everyone explores, odd A
    ("choose_masked_actions_vec", 50, 12, 40, True, 0.4, False, "f8"),
    ("choose_masked_actions_vec", 50, 260, 10, True, 0.4, False, "f4"),  # A > 256
    ("choose_actions_vec", 20, 1000, 6, False, 0.2, False, "f8"),  # large A
]

LEARN_CASES = [
    # (S, A, n, masked, dtype, lr, gamma)
    (40, 4, 64, False, "f8", 0.1, 0.99),
    (40, 4, 64, False, "f4", 0.1, 0.99),
    (30, 16, 200, False, "f8", 0.5, 0.9),
    (30, 16, 200, False, "f4", 0.5, 0.9),
    (25, 64, 100, True, "f8", 0.3, 0.97),
    (25, 64, 100, True, "f4", 0.3, 0.97),
    (1, 2, 32, False, "f4", 1.0, 1.0),  # bandit shape: every transition depends on the previous
    (5000, 8, 128, False, "f4", 0.1, 0.99),  # sparse: almost no collisions
    (60, 9, 90, True, "f8", 0.2, 0.95),  # A not a multiple of 4
]

TRACE_CASES = [
    # (name, env spec, steps, dtype, schedule kind, learn fn)
    ("c1_grid_n1", ("grid", 1, 10), 80, "f8", "bench", "learn"),
    ("grid4_n1", ("grid", 1, 4), 300, "f8", "bench", "learn"),
    ("grid4_n16", ("grid", 16, 4), 100, "f4", "const", "learn"),
    ("grid4_n16_f8", ("grid", 16, 4), 100, "f8", "const", "learn"),
    ("c2_hash_n128", ("hash", 128, 10000, 8, False), 50, "f4", "bench", "learn"),
    ("c2_hash_n128_const", ("hash", 128, 10000, 8, False), 50, "f4", "const", "learn"),
    ("hash_dense_n256", ("hash", 256, 64, 16, False), 40, "f4", "const", "learn"),
    ("hash_dense_n256_f8", ("hash", 256, 64, 16, False), 40, "f8", "const", "learn"),
    ("hash_dense_n256_vec", ("hash", 256, 64, 16, False), 40, "f8", "const", "learn_vec"),
    ("c5_hash_masked_n128", ("hash", 128, 500, 64, True), 40, "f4", "const", "learn"),
    ("hash_masked_a9_n64", ("hash", 64, 300, 9, True), 40, "f8", "const", "learn"),
    ("bandit_n4", ("bandit", 4, 5), 23, "f8", "kat", "learn"),
    ("bandit_n128", ("bandit", 128, 7), 30, "f4", "const", "learn"),
    ("ttt_n64", ("ttt", 64), 60, "f8", "bench", "learn"),  # masked, A = 9: choose_masked_action (list) path
    ("ttt_n128_f4", ("ttt", 128), 50, "f4", "const", "learn"),
]

# ---- round 3 (make_golden_r3.py) -----------------------------------------------------------------------------------
SCALE_TRACE_CASES = [
    # the shapes SURVEY 8(c) lists beyond 256 agents
    ("c3_shrunk_n4096", ("hash", 4096, 1_000_000, 16, False), 30, "f4", "bench", "learn"),
    ("c3_shrunk_n4096_const", ("hash", 4096, 1_000_000, 16, False), 20, "f4", "const", "learn"),
    ("c5_small_n1024", ("hash", 1024, 4096, 64, True), 30, "f4", "const", "learn"),
    ("c5_small_n1024_bench", ("hash", 1024, 4096, 64, True), 20, "f4", "bench", "learn"),
]

NAN_SELECT_CASES = [
    # (method, S, A, n, masked, eps, deterministic, dtype, NaN cells, table seed)
    ("choose_actions_iter", 40, 6, 60, False, 0.3, False, "f4", 30, 1),     # list scan steps over NaN
    ("choose_actions_iter", 40, 6, 60, True, 0.3, False, "f8", 60, 2),
    ("choose_actions_iter", 12, 4, 40, False, 0.0, True, "f4", 30, 3),      # rows that are NaN throughout: -1
    ("choose_actions_vec_iter", 400, 12, 30, False, 1.0, False, "f4", 40, 4),  # everyone explores: no maximum taken
    ("choose_actions_vec_iter", 4000, 12, 8, False, 0.3, False, "f4", 3, 5),
    ("choose_actions_vec_iter", 40, 12, 30, False, 0.3, False, "f4", 40, 6),   # IndexError
    ("choose_actions_vec_iter", 4000, 12, 8, True, 0.3, False, "f8", 3, 7),
    ("choose_actions_vec", 400, 8, 120, False, 1.0, False, "f4", 30, 8),
    ("choose_actions_vec", 40, 8, 120, False, 0.2, False, "f4", 6, 9),        # IndexError
    ("choose_actions_vec", 90000, 8, 120, False, 0.2, False, "f4", 2, 10),
    ("choose_masked_actions_vec", 40, 12, 50, True, 0.2, False, "f4", 10, 11),  # IndexError unless the NaN is masked out
    ("choose_masked_actions_vec", 20000, 12, 50, True, 0.2, False, "f8", 4, 12),
    ("choose_actions", 50, 8, 64, False, 0.2, False, "f4", 20, 13),            # n < 100 -> list variant
    ("choose_actions", 50, 8, 128, False, 0.2, False, "f4", 4, 14),            # n >= 100 -> NumPy variant (IndexError)
    ("choose_actions", 50, 9, 128, True, 0.2, False, "f4", 20, 15),            # masked, A <= 10 -> list
    ("choose_actions", 50, 16, 128, True, 0.2, False, "f4", 4, 16),            # masked, A > 10 -> NumPy (IndexError)
    ("choose_actions", 50, 8, 128, False, 0.0, True, "f4", 20, 17),            # deterministic, A <= 10 -> list
    ("choose_actions", 50, 16, 128, False, 0.0, True, "f4", 4, 18),            # deterministic, A > 10 -> NumPy
]

NAN_LEARN_CASES = [
    # (S, A, n, masked, dtype, lr, gamma, NaN cells, table seed)
    (40, 4, 64, False, "f4", 0.1, 0.99, 8, 1),
    (40, 4, 64, False, "f8", 0.1, 0.99, 8, 2),
    (30, 16, 200, False, "f4", 0.5, 0.9, 20, 3),
    (25, 64, 100, True, "f4", 0.3, 0.97, 40, 4),   # a masked-out NaN does not reach the target
    (60, 9, 90, True, "f8", 0.2, 0.95, 25, 5),
    (5000, 8, 128, False, "f4", 0.1, 0.99, 300, 6),
]

NAN_TRACE_CASES = [
    # (name, env spec, chunks, dtype, schedule kind, learn fn, NaN cells (0: zero table), table seed)
    ("diverge_bandit_n600_vec", ("bandit", 600, 5), [10] * 6, "f4", "nan", "learn_vec", 0, 0),
    ("diverge_hash_n600_vec", ("hash", 600, 4, 4, False), [8] * 6, "f4", "nan", "learn_vec", 0, 0),
    ("diverge_hash_n256_masked_vec", ("hash", 256, 5, 16, True), [8] * 6, "f4", "nan", "learn_vec", 0, 0),
    ("nan_explore_n200", ("hash", 200, 50, 8, False), [10, 10, 10], "f4", "explore", "learn", 40, 3),
    ("nan_raise_n200", ("hash", 200, 2000, 8, False), [3, 3, 3, 3], "f4", "const", "learn", 4, 5),
    ("nan_raise_n1024", ("hash", 1024, 30000, 16, False), [3, 3, 3, 3], "f4", "const", "learn", 6, 5),
    ("nan_list_n64", ("hash", 64, 50, 8, False), [4, 4, 4], "f4", "const", "learn", 6, 4),
    ("nan_list_masked_a8", ("hash", 128, 300, 8, True), [5, 5], "f4", "const", "learn", 20, 4),
    ("nan_raise_masked_a16", ("hash", 128, 3000, 16, True), [3, 3, 3, 3], "f4", "const", "learn", 6, 3),
    ("nan_lean_n128", ("hash", 128, 2000, 16, False), [20, 20, 20, 20], "f4", "const", "learn", 1, 4),
    ("nan_ttt_n128", ("ttt", 128), [10, 10], "f4", "const", "learn", 300, 3),
]
