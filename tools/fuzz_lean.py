#!/usr/bin/env python3
"""Random shapes through the LEAN build of the shape-specialised batch kernels (needs a GPU): unlogged launches on epoch grids with
RS_JIT_LEAN_MIN_TTIS = 1, uneven launch lengths, random CQI refresh periods (1 ... 40: streamed and resident modes); the final state
(bitwise PF averages, exact counters, exact slice state) against the CPU oracle.  Test infrastructure: uses oracle/.
    python tools/fuzz_lean.py [first_seed] [n_seeds]
"""
import os
import sys
from pathlib import Path

os.environ["RS_JIT_LEAN_MIN_TTIS"] = "1"
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import numpy as np  # noqa: E402
import radiosaber_amd as rs  # noqa: E402
from conftest import synth_cqi  # noqa: E402
from oracle import oracle_py  # noqa: E402
from test_gpu_parity import HIST  # noqa: E402

oracle_py.lib()
first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
for seed in range(first, first + n):
    rng = np.random.default_rng(seed)
    for trial in range(4):
        S = int(rng.integers(1, 24))
        ues = [int(x) for x in rng.integers(0, 30, S)]
        if sum(ues) == 0:
            ues[0] = 3
        R, G = [(25, 4), (64, 8), (12, 2), (50, 2), (17, 3), (33, 3)][int(rng.integers(0, 6))]
        w = rng.uniform(0.2, 1.0, S)
        w = [float(x) for x in w / w.sum()]
        sched = [9, 9, 9, 8, 8, 7, 1, 103, 10, 11, 101][int(rng.integers(0, 11))]
        threads = [0, 128, 256, 512][int(rng.integers(0, 4))]
        if sched == 10 and threads and R * S > 4 * threads:
            threads = 0
        psi = [int(x) for x in rng.integers(0, 2, S)] if sched != 1 else None
        refresh = [40, 40, 1, 2, 7, 3][int(rng.integers(0, 6))]
        n_ttis = int(rng.integers(50, 140))
        sc = rs.SliceConfig(ues, weight=w, algo_psi=psi or [])
        n_cells = 2
        grids = synth_cqi(seed * 100 + trial, (n_cells, (n_ttis + refresh - 1) // refresh, sc.n_users, R), HIST)
        seeds = np.arange(n_cells, dtype=np.uint32) * 7919 + 805290992
        b = rs.BatchScheduler(sc, R, G, n_cells, sched=sched, threads_per_cell=threads, jit=True, cqi_refresh=refresh)
        b.seed(seeds)
        b.upload_cqi_epochs(grids)
        done = 0
        while done < n_ttis:
            k = min(int(rng.integers(1, 60)), n_ttis - done)
            b.run(k)
            done += k
        st = b.state()
        for c in range(n_cells):
            cell = oracle_py.Cell(ues, R, G, sched, weights=w, psi=psi)
            cell.run_synth(grids[c], int(seeds[c]), n_ttis, refresh=refresh, log=False)
            o = cell.state()
            what = f"seed {seed} trial {trial} sched {sched} ues {ues} R {R} refresh {refresh} threads {threads} cell {c}"
            assert np.array_equal(st["cum_bytes"][c], o["cum_bytes"]) and np.array_equal(st["cum_rbs"][c], o["cum_rbs"]), what
            assert st["avg_rate"][c].tobytes() == o["avg_rate"].tobytes(), what
            assert st["slice_state"][c].tobytes() == o["slice_state"].tobytes(), what
        b.close()
print(f"fuzz (lean build): seeds {first}..{first + n - 1} bit-exact")
