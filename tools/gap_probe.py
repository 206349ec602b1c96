#!/usr/bin/env python3
"""The headline batch, launch by launch: HIP-event duration, the kernel's own duration and the SHADER CLOCK it ran at (the kernel
reads s_memtime and s_memrealtime at its first and last instruction: rs_batch_debug_clocks) -- with an optional host sleep between
launches.  Evidence for / against the two explanations of "rocprofv3 sees the cell kernel 2-3 % faster than a plain run"
(profiles/r04_rocprof_vs_events.md): a different clock under the profiler, or launches that start on a drained GPU.

    python3 tools/gap_probe.py [--sleep-ms X] [--launches 12] [--tag plain]      (also under rocprofv3: the program right after `--`)
"""
import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def sclk():
    """current shader clock level as the driver reports it (sysfs; None when not readable)"""
    out = []
    for f in sorted(Path("/sys/class/drm").glob("card*/device/pp_dpm_sclk")):
        try:
            out += [ln.strip() for ln in f.read_text().split("\n") if "*" in ln]
        except OSError:
            pass
    return out or None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sleep-ms", type=float, default=0.0)
    ap.add_argument("--launches", type=int, default=12)
    ap.add_argument("--ttis", type=int, default=8000)
    ap.add_argument("--tag", default="plain")
    a = ap.parse_args()
    import radiosaber_amd as rs
    sc = rs.SliceConfig([25] * 20, weight=[0.05] * 20)
    cells = 512
    b = rs.BatchScheduler(sc, 25, 4, cells, sched=9, jit=True, cqi_epoch_wrap=True)
    b.seed((np.arange(cells, dtype=np.uint64) * 2654435761 + 805290992).astype(np.uint32))
    b.synthesize_cqi(0x5AB3, 1200)
    b.prepare_launch(a.ttis)
    b.run(a.ttis)
    b.run(a.ttis)
    rows = []
    for i in range(a.launches):
        if a.sleep_ms:
            time.sleep(a.sleep_ms / 1e3)
        before = sclk()
        ms = float(b.run_timed(a.ttis, 1)[0])
        mhz, kms = b.debug_clocks()
        rows.append({"event_ms": ms, "cell_ms_max": float(kms.max()), "cell_ms_mean": float(kms.mean()),
                     "shader_mhz_mean": float(mhz.mean()), "shader_mhz_min": float(mhz.min()), "shader_mhz_max": float(mhz.max()),
                     "sclk_sysfs_before": before})
    b.close()
    ev = np.array([r["event_ms"] for r in rows])
    mh = np.array([r["shader_mhz_mean"] for r in rows])
    print(json.dumps({"tag": a.tag, "sleep_ms": a.sleep_ms, "launches": a.launches, "event_ms_mean": float(ev.mean()), "event_ms_min": float(ev.min()),
                      "event_ms_max": float(ev.max()), "shader_mhz_mean": float(mh.mean()), "shader_mhz_min": float(mh.min()),
                      "cycles_per_launch_mean_M": float((ev * mh * 1e3).mean() / 1e6), "rows": rows}))


if __name__ == "__main__":
    main()
