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
