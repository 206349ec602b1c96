#!/usr/bin/env python3
"""Long-run parity soak on the GPU box: many cells x thousands of TTIs, final state vs the CPU oracle
(bitwise PF averages, exact counters, exact slice state).  Exercises rare paths (heap fallback if it ever
fires, >110-PRB TBS rule, long tie chains).  Test infrastructure: uses oracle/."""
import argparse
import sys
import time
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import radiosaber_amd as rs  # noqa: E402
from oracle import oracle_py as O  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--cells", type=int, default=32)
ap.add_argument("--ttis", type=int, default=4000)
ap.add_argument("--sched", type=int, default=9)
ap.add_argument("--rbgs", type=int, default=25)
ap.add_argument("--rbg-size", type=int, default=4)
ap.add_argument("--ues-per-slice", type=int, default=25)
ap.add_argument("--slices", type=int, default=20)
ap.add_argument("--phy", type=int, default=0)
ap.add_argument("--jit", type=int, default=0)
ap.add_argument("--threads", type=int, default=0)
ap.add_argument("--launch", type=int, default=0, help="TTIs per launch (0: one launch)")
a = ap.parse_args()
ues = [a.ues_per_slice] * a.slices
w = [1.0 / a.slices] * a.slices
sc = rs.SliceConfig(ues, weight=w)
n_ep = (a.ttis + 39) // 40
b = rs.BatchScheduler(sc, a.rbgs, a.rbg_size, a.cells, sched=a.sched, phy_error_draws=bool(a.phy), jit=bool(a.jit),
                      threads_per_cell=a.threads)
seeds = (np.arange(a.cells, dtype=np.uint32) * 2654435761 + 99) % (2**31 - 1)
b.seed(seeds.astype(np.uint32))
b.synthesize_cqi(4242, n_ep)
t0 = time.time()
done = 0
while done < a.ttis:
    n = min(a.launch or a.ttis, a.ttis - done)
    b.run(n)
    done += n
st = b.state()
print(f"gpu: {a.cells} cells x {a.ttis} TTIs in {time.time() - t0:.2f} s")
grids = [b.download_cqi_epochs(c) for c in range(a.cells)]


def one(c):
    cell = O.Cell(ues, a.rbgs, a.rbg_size, a.sched, weights=w)
    cell.run_synth(grids[c], int(seeds[c]), a.ttis, phy_error_draws=a.phy, log=False)
    return cell.state()


t0 = time.time()
with ThreadPoolExecutor(16) as ex:
    ref = list(ex.map(one, range(a.cells)))
print(f"oracle: {time.time() - t0:.1f} s")
bad = 0
for c in range(a.cells):
    ok = (st["cum_bytes"][c] == ref[c]["cum_bytes"]).all() and (st["cum_rbs"][c] == ref[c]["cum_rbs"]).all() and \
        st["avg_rate"][c].tobytes() == ref[c]["avg_rate"].tobytes() and \
        st["slice_state"][c].tobytes() == ref[c]["slice_state"].tobytes()
    bad += not ok
print("SOAK", "OK" if bad == 0 else f"MISMATCH in {bad} cells", f"sched {a.sched} R {a.rbgs} U {sc.n_users} jit {a.jit} threads {a.threads} launch {a.launch}",
      "max RBGs of one UE-TTI not tracked; total bytes", int(st["cum_bytes"].sum()))
sys.exit(1 if bad else 0)
