#!/usr/bin/env python3
"""Longer run of tests/test_gpu_parity.py::test_random_shapes_all_schedulers (needs a GPU):
    python tools/fuzz_parity.py [first_seed] [n_seeds]
"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import radiosaber_amd as rs  # noqa: E402
import test_gpu_parity as T  # noqa: E402
from oracle import oracle_py  # noqa: E402

oracle_py.lib()
first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
for seed in range(first, first + n):
    T.test_random_shapes_all_schedulers(rs, oracle_py, seed)
print(f"fuzz: seeds {first}..{first + n - 1} bit-exact")
