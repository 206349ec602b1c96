#!/usr/bin/env python3
"""tests/golden/ref_clock.json from the reference's own event core (build container only).

Runs oracle/_ref/libref_clock.so -- src/core/eventScheduler/{simulator.cc,calendar.cpp,event.cpp} compiled where they
lie -- for 700 subframes with one application-start event at 0.1 s, and oracle/_ref/libref_bw.so
(src/core/spectrum/bandwidth-manager.cpp) for the PRB counts.  Data only: hex floats and integers."""
import ctypes as C
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from oracle import oracle_py as O  # noqa: E402

N = 700
clk = O.ref_lib("libref_clock.so")
bw = O.ref_lib("libref_bw.so")
assert clk is not None and bw is not None, "run `make -C oracle` with /root/reference present"
clk.ref_clock_run.argtypes = [C.c_int, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int)]
out = np.zeros(N)
app_now, app_before = C.c_double(), C.c_int()
assert clk.ref_clock_run(N, 0.1, out.ctypes.data_as(C.POINTER(C.c_double)), C.byref(app_now), C.byref(app_before)) == N
bw.ref_dl_subchannels.argtypes = [C.c_double]
fix = {
    "source": "oracle/_ref/libref_clock.so + libref_bw.so (reference sources compiled in place, g++ -O0)",
    "subframe_start": [float(x).hex() for x in out],
    "app_start_now": float(app_now.value).hex(),
    "subframes_before_app_start": int(app_before.value),
    "dl_prbs": {str(b): int(bw.ref_dl_subchannels(b)) for b in (1.4, 3, 5, 10, 15, 20, 100, 7)},
}
(ROOT / "tests" / "golden" / "ref_clock.json").write_text(json.dumps(fix, indent=0) + "\n")
print("wrote tests/golden/ref_clock.json:", out[100].hex(), fix["dl_prbs"])
