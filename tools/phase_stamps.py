#!/usr/bin/env python3
"""Diagnostic: per-phase cycle shares of the cell kernel (needs the -DRS_STAMPS build + a GPU).

    RS_HIP_LIB=radiosaber_amd/libradiosaber_hip_stamps.so python tools/phase_stamps.py [--rbgs 25 --rbg-size 4 --sched 9 --threads 256]
"""
import argparse
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import radiosaber_amd as rs  # noqa: E402

NAMES = ["P0+P1 refresh/EWMA (+barrier)", "P2 quotas (if wave 0)", "P3 best user (+barrier)", "P4a introsort loop",
         "P4b counting sort", "P4c greedy (wave 0)", "wave 0 waits for the scanning waves", "P5 link adapt + counters", "barrier end of TTI",
         "-", "-", "loop head"]

ap = argparse.ArgumentParser()
ap.add_argument("--cells", type=int, default=512)
ap.add_argument("--ttis", type=int, default=400)
ap.add_argument("--rbgs", type=int, default=25)
ap.add_argument("--rbg-size", type=int, default=4)
ap.add_argument("--sched", type=int, default=9)
ap.add_argument("--threads", type=int, default=0)
ap.add_argument("--ues-per-slice", type=int, default=25)
ap.add_argument("--w1", action="store_true", help="-DRS_STAMPS_W1 build: the sub-stamp slots hold wave 1's serial-phase clock")
ap.add_argument("--jit", action="store_true", help="shape-specialised kernel (export RS_JIT_EXTRA=-DRS_STAMPS)")
ap.add_argument("--p5", action="store_true", help="-DRS_STAMPS_P5 build: sub-stamp slots = the steps of P5 on wave 0")
ap.add_argument("--hold", action="store_true", help="-DRS_STAMPS_HOLD build: sub-stamp slots 0 / 1 = the held-winner scan's list step / item passes (wave 0)")
ap.add_argument("--cqi-refresh", type=int, default=40, help="TTIs between two CQI grids (1: streamed-CQI mode; the epochs cycle through 64 grids)")
ap.add_argument("--queues", action="store_true", help="the queue model on exp-customize-20slices (tools/bench_queue_mode.py's workload)")
a = ap.parse_args()
if a.queues:
    import json
    sys.path.insert(0, str(Path(__file__).resolve().parent))
    import bench_queue_mode as Q
    cfg = json.loads((Q.GOLDEN / "experiment_configs.json").read_text())["exp-customization/exp-customize-20slices/config.json"]
    sc = rs.SliceConfig(cfg["ues_per_slice"], cfg["weight"], cfg["algo_alpha"], cfg["algo_beta"], cfg["algo_epsilon"],
                        cfg["algo_psi"], cfg["traffic"])
    per_cell = Q.build_bursts(cfg, sc, 8, 2 * a.ttis)
    bursts = {(c, u, k): v for c in range(a.cells) for (u, k), v in per_cell[c % 8].items()}
else:
    sc = rs.SliceConfig([a.ues_per_slice] * 20, weight=[0.05] * 20)
b = rs.BatchScheduler(sc, a.rbgs, a.rbg_size, a.cells, sched=a.sched, threads_per_cell=a.threads, jit=a.jit,
                      cqi_refresh=a.cqi_refresh, cqi_epoch_wrap=a.cqi_refresh != 40)
if a.queues:
    b.set_bearers(sc.bearer_kinds())
    b.set_arrivals(bursts)
    print("jit status:", b.jit_status())
b.seed(np.arange(a.cells, dtype=np.uint32) + 1)
b.synthesize_cqi(1, (2 * a.ttis + 39) // 40 if a.cqi_refresh == 40 else 64)
b.run(a.ttis)
ms = b.run_timed(a.ttis, 1)
st = np.stack([b.debug_stamps(c) for c in (0, a.cells // 2, a.cells - 1)]).astype(np.float64)
tot = st.sum(1)
print("greedy: RBGs assigned / TTI", st[:, 9] / a.ttis, " sorted position of the last assignment (mean)", st[:, 10] / a.ttis)
print(f"launch {ms[0]:.3f} ms, {ms[0] * 1e3 / a.ttis:.2f} us/TTI/cell; cycles/TTI (thread 0): {tot / a.ttis}")
if a.p5:
    SUB = ["P5: lanes per user (LDS atomics)", "P5: leaders, offsets, flags", "P5: E values + ordered sums", "P5: synthetic-exp bits", "P5: divide + classify",
           "P5: MCS / TBS / counters", "P5: exact EWMA of the served", "-"]
elif a.hold:
    SUB = ["hold: need + list", "hold: passes over the listed items", "-", "-", "-", "-", "-", "-"]
elif a.w1:
    SUB = ["scan: EWMA of every user", "scan: wait for the other scanning waves", "scan: items", "scan: wait for the allocation", "scan: (quotas,) served check, list", "-", "-", "-"]
else:
  SUB = ["sort F (pivot+ballots)", "sort barrier 1", "sort R (counts, exchange)", "sort barrier 2", "sort S (receive, descend)", "sort barrier 3", "-", "sort levels (count)"]
tot = st[:, :12].sum(1)
if not a.w1:
    hold = st[:, 18].astype(np.uint64)
    print("held winners: items wave 0 listed (of its 64) per TTI that was not a full scan", (hold & np.uint64(0xffffffff)) / np.maximum(a.ttis - (hold >> np.uint64(32)), 1),
          " full-scan TTIs", hold >> np.uint64(32), "of", a.ttis)
for i, n in enumerate(SUB):
    print(f"    {n:26s} " + "  ".join(f"{st[c, 12 + i] / a.ttis:9.1f}" for c in range(3)))
for i, n in enumerate(NAMES):
    if n != "-":
        print(f"  {n:28s} " + "  ".join(f"{st[c, i] / a.ttis:9.0f} ({100 * st[c, i] / tot[c]:4.1f}%)" for c in range(3)))
