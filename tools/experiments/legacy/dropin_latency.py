#!/usr/bin/env python3
"""Latency of the drop-in mode: one rs_schedule_tti (= one RBsAllocation()) per call, host buffers in and out."""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import radiosaber_amd as rs  # noqa: E402

for (ues, R, G) in (([5] * 20, 64, 8), ([25] * 20, 25, 4), ([25] * 20, 64, 8)):
    sc = rs.SliceConfig(ues, weight=[0.05] * 20)
    U = sc.n_users
    ts = rs.TtiScheduler(sc, R, G, sched=9)
    rng = np.random.default_rng(0)
    cqi = rng.integers(1, 16, (U, R)).astype(np.uint8)
    avg = rng.uniform(1e4, 1e6, U)
    for _ in range(20):
        ts.schedule_tti(cqi, avg, 123, 456)
    n = 300
    t0 = time.perf_counter()
    for i in range(n):
        ts.schedule_tti(cqi, avg, 123 + i, 456 + i)
    dt = (time.perf_counter() - t0) / n
    print(f"U={U} R={R}: {dt * 1e6:.1f} us per rs_schedule_tti (python ctypes call included)")
    ts.close()
