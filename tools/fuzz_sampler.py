#!/usr/bin/env python3
"""Random shapes through the NVS non-greedy sampler (scheduler 11) against the oracle (needs a GPU): ragged slices of 1 ... 70 users
(keys up to 64 users, doubles beyond), grids of 6 ... 64 RBGs, one to eight waves per cell, built-in and shape-specialised kernels,
error-model draws on and off.
    python tools/fuzz_sampler.py [first_seed] [n_seeds]
ORACLE UNPINNED for scheduler 11 (tests/PINS.md): this proves device == oracle, nothing more."""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import radiosaber_amd as rs  # noqa: E402
import test_gpu_parity as T  # noqa: E402
from oracle import oracle_py  # noqa: E402

oracle_py.lib()
first = int(sys.argv[1]) if len(sys.argv) > 1 else 17000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
GRIDS = [(6, 1), (12, 2), (17, 3), (25, 4), (27, 4), (33, 3), (34, 3), (40, 3), (50, 4), (63, 8), (64, 8)]
for seed in range(first, first + n):
    rng = np.random.default_rng(seed)
    S = int(rng.integers(1, 9))
    kind = int(rng.integers(0, 4))
    if kind == 0:
        ues = [int(x) for x in rng.integers(1, 16, S)]           # the reference's experiment sizes
    elif kind == 1:
        ues = [int(x) for x in rng.integers(0, 71, S)]           # around the 64-user limit of the keys, empty slices
    elif kind == 2:
        ues = [int(rng.integers(1, 65))] * S                     # equal slices
    else:
        ues = [int(x) for x in rng.choice([1, 8, 16, 32, 64, 5, 25], S)]
    if sum(ues) == 0:
        ues[0] = 3
    R, G = GRIDS[int(rng.integers(0, len(GRIDS)))]
    threads = int(rng.choice([0, 0, 64, 128, 256, 448, 512]))
    jit = bool(rng.integers(0, 2))
    T._check_batch(rs, oracle_py, 11, ues, R, G, n_cells=2, n_ttis=int(rng.integers(41, 70)), threads=threads,
                   phy=int(rng.integers(0, 2)), seed=seed, jit=jit)
print(f"sampler fuzz: seeds {first}..{first + n - 1} bit-exact")
