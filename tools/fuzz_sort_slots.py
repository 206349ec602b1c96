#!/usr/bin/env python3
"""Random shapes for the sort's position slots (needs a GPU): MaximizeCell and UpperBound on random (slices, RBGs, workgroup size)
so that the record array takes one to four slots per thread with every kind of last slot, built-in and run-time kernels, a few
dozen TTIs each, bit-exact against the oracle (tests/test_gpu_round6_sort.py is the fixed list of shapes; this is the campaign).
    python tools/fuzz_sort_slots.py [first_seed] [n_seeds]
"""
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
first = int(sys.argv[1]) if len(sys.argv) > 1 else 500
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
done = 0
for seed in range(first, first + n):
    rng = np.random.default_rng(seed)
    jit = bool(rng.integers(0, 2))
    threads = int(rng.choice([128, 192, 256, 320, 512] + ([640, 768, 1024] if jit else [])))
    slots = int(rng.integers(1, 5))
    lo, hi = (slots - 1) * threads + 1, slots * threads
    for _ in range(200):
        S = int(rng.integers(2, 65))
        R = int(rng.integers(6, 65))
        if lo <= S * R <= hi:
            break
    else:
        continue
    G = int(rng.choice([1, 2, 4, 8]))
    ues = [int(x) for x in rng.integers(0, 5, S)]
    if sum(ues) == 0:
        ues[0] = 3
    sched = 10 if (rng.random() < 0.3 and jit) else 9
    if rs.lds_bytes_per_cell(S, max(sum(ues), 1), R, sched, threads) > 160 * 1024:
        continue
    T._check_batch(rs, oracle_py, sched, ues, R, G, n_cells=2, n_ttis=int(rng.integers(20, 60)), threads=threads, jit=jit, seed=seed)
    done += 1
    print(f"seed {seed}: sched {sched}, {S} slices x {R} RBGs = {S * R} records on {threads} threads ({slots} slots), jit {int(jit)}", flush=True)
print(f"fuzz_sort_slots: {done} shapes of seeds {first}..{first + n - 1} bit-exact")
