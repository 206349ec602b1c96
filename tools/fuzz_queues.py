#!/usr/bin/env python3
"""Longer run of the queue-model parity cases (tests/test_gpu_queues.py) with fresh seeds (needs a GPU):
    python tools/fuzz_queues.py [first_seed] [n_seeds]
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import radiosaber_amd as rs  # noqa: E402
import test_gpu_queues as T  # noqa: E402
from oracle import oracle_py  # noqa: E402

oracle_py.lib()
first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
kinds_pool = ["B-", "Q-", "QQ", "BQ"]
for seed in range(first, first + n):
    rng = np.random.default_rng(seed)
    S = int(rng.integers(2, 9))
    ues = [int(x) for x in rng.integers(1, 12, S)]
    kinds = [kinds_pool[int(rng.integers(0, 4))] for _ in range(S)]
    alpha = [int(x) for x in rng.integers(0, 2, S)]
    beta = [int(x) for x in rng.integers(0, 2, S)]
    psi = [int(x) for x in rng.integers(0, 2, S)]
    sched = [9, 8, 7, 1, 101, 103][int(rng.integers(0, 6))]
    R, G = [(25, 4), (64, 8), (12, 2), (17, 3)][int(rng.integers(0, 4))]
    if sched == 1:
        alpha, beta, psi = [0] * S, [0] * S, None
    T._run_case(rs, oracle_py, sched, ues, kinds, alpha, beta, R, G, n_cells=2, launches=[int(rng.integers(1, 60)), int(rng.integers(40, 120))],
                jit=bool(seed % 2), seed=seed, threads=[0, 128, 256, 512][int(rng.integers(0, 4))],
                mean_gap_ms=int(rng.integers(3, 40)), mean_bytes=int(rng.integers(200, 20000)), psi=psi)
print(f"queue fuzz: seeds {first}..{first + n - 1} bit-exact")
