#!/usr/bin/env python3
"""Copy the rocprofv3 summaries gpurun merged into gpurun_out/ into profiles/ and derive
profiles/traffic.json (HBM bytes per launch of the cell kernel from the PMC passes).

    python tools/summarize_rocprof.py r02          (the workload key comes from the profiled bench line)

Counter handling per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE and
WRITE_SIZE are collected in separate passes, both are in KiB, and on gfx950 FETCH_SIZE reports half
of the bytes of wide (16 B/lane) coalesced reads, so the read side is doubled.  Other access widths
are uncalibrated; the figure is an upper-bound style estimate of fabric-side traffic.
"""
import csv
import glob
import json
import shutil
import statistics
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
import subprocess

tag = sys.argv[1]
out = ROOT / "profiles"
bench_line = json.loads([ln for ln in (ROOT / "gpurun_out" / "prof_kt.log").read_text().splitlines() if ln.startswith("{")][-1])
cfg = bench_line["config"]
key = f"sched{cfg['sched']}_S{cfg['slices']}_U{cfg['ues']}_R{cfg['rbgs']}_cells{cfg['cells_per_gpu']}"
if cfg.get("cqi_refresh", 40) != 40:
    key += f"_refresh{cfg['cqi_refresh']}"
cell_ttis = cfg["cells_per_gpu"] * cfg["ttis_per_step"]
commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()


def one(pattern):
    # newest file wins (gpurun merges new runs next to older ones)
    f = sorted(glob.glob(str(ROOT / "gpurun_out" / pattern)), key=lambda x: Path(x).stat().st_mtime)
    if not f:
        raise SystemExit(f"missing {pattern}")
    return Path(f[-1])


ks = one("prof_kt/*/*_kernel_stats.csv")
shutil.copy(ks, out / f"{tag}_kernel_stats.csv")
rows = list(csv.DictReader(ks.open()))
cell = max((r for r in rows if "rs_cell_kernel" in r["Name"]), key=lambda r: float(r["TotalDurationNs"]))


def counter(pattern, name):
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(one(pattern).open())
            if "rs_cell_kernel" in r["Kernel_Name"] and r["Counter_Name"] == name]
    return vals


fetch = counter("prof_fetch/*/*_counter_collection.csv", "FETCH_SIZE")
write = counter("prof_write/*/*_counter_collection.csv", "WRITE_SIZE")
fetch_b = statistics.mean(fetch) * 1024 * 2  # gfx950: wide reads are tallied at half their bytes
write_b = statistics.mean(write) * 1024
traffic = {"kernel": cell["Name"], "launches_profiled": len(fetch),
           "FETCH_SIZE_KiB_mean": statistics.mean(fetch), "WRITE_SIZE_KiB_mean": statistics.mean(write),
           "hbm_read_bytes_per_launch": fetch_b, "hbm_write_bytes_per_launch": write_b,
           "hbm_bytes_per_launch": fetch_b + write_b, "ttis_per_launch": cfg["ttis_per_step"],
           "hbm_bytes_per_cell_tti": (fetch_b + write_b) / cell_ttis, "commit": commit, "round": tag,
           "source_hash": bench_line.get("source_hash"),  # device sources of the profiled library (bench.py flags a mismatch as stale)
           "kernel_trace_avg_ns": float(cell["AverageNs"]), "kernel_trace_calls": int(cell["Calls"]),
           "note": "FETCH_SIZE doubled per the gfx950 correction; cumulative byte / RB counters ride in registers and "
                   "are flushed once per launch (round 1: two 8-byte atomics per served UE per TTI)"}
tf = out / "traffic.json"
tj = json.loads(tf.read_text()) if tf.exists() else {}
tj[key] = traffic
tf.write_text(json.dumps(tj, indent=1))
print(json.dumps(traffic, indent=1))
