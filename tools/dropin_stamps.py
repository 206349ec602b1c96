#!/usr/bin/env python3
"""Diagnostic: phase cycles of ONE drop-in call (rs_schedule_tti) with the -DRS_STAMPS build.

    RS_HIP_LIB=radiosaber_amd/libradiosaber_hip_stamps.so python tools/dropin_stamps.py

The stamps start after the kernel's load phase and end with the TTI: what the kernel's duration (rocprofv3) has on top of
their sum is the load phase, the write-back and the launch itself."""
import ctypes as C
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import radiosaber_amd as rs  # noqa: E402
from radiosaber_amd.api import lib  # noqa: E402

NAMES = ["P0+P1", "P2", "P3", "introsort", "counting", "greedy", "wait", "P5", "end barrier", "load phase", "store phase", "load..loop head"]
import os  # noqa: E402
JIT = os.environ.get("RS_STAMPS_JIT", "0") == "1"      # the specialised kernel (needs RS_JIT_EXTRA=-DRS_STAMPS as well)
EPOCH = os.environ.get("RS_STAMPS_EPOCH", "0") == "1"  # rs_tti_in.cqi_epoch: the calls read the device-resident image
for sched, ues, R, G in ((9, 25, 25, 4), (9, 25, 64, 8), (9, 5, 64, 8), (8, 25, 25, 4), (1, 25, 25, 4)):
    sc = rs.SliceConfig([ues] * 20, weight=[0.05] * 20)
    U = 20 * ues
    ts = rs.TtiScheduler(sc, R, G, sched=sched, jit=JIT)
    rng = np.random.default_rng(0)
    cqi = rng.integers(1, 16, (U, R)).astype(np.uint8)
    avg = rng.uniform(1e4, 1e6, U)
    for i in range(50):
        ts.schedule_tti(cqi, avg, 123 + i, 456 + i, cqi_epoch=1 if EPOCH else 0)
    t0 = time.perf_counter()
    for i in range(200):
        ts.schedule_tti(cqi, avg, 1123 + i, 1456 + i, cqi_epoch=1 if EPOCH else 0)
    us = (time.perf_counter() - t0) / 200 * 1e6
    batch = C.cast(ts._h, C.POINTER(C.c_void_p))[0]  # rs_ctx's first member is its one-cell batch
    out = (C.c_uint64 * 20)()
    f = lib().rs_batch_debug_stamps
    f.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_uint64)]
    rc = f(batch, 0, out)
    st = np.array(out[:12], dtype=np.float64)
    print(f"sched {sched} {U} UEs x {R} RBGs ({'specialised' if JIT else 'built-in'}{', cqi_epoch' if EPOCH else ''}): {us:.1f} us per call (python); "
          f"stamped cycles {st.sum():.0f}: " + ", ".join(f"{n} {v:.0f}" for n, v in zip(NAMES, st) if v))
    ts.close()
