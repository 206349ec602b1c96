#!/usr/bin/env python3
"""Summarise the queue-model passes of tools/profile_queue_mode.sh into profiles/queue_mode.json.

    python tools/summarize_queue_prof.py <tag> <label>       e.g. r03q0 before / r03q1 after

Per scheduler (run order 9, 7, 1; the first launch of every batch is the untimed warm-up): HIP-event TTIs/s of
tools/bench_queue_mode.py (with the backlogged twin of the same shape), rocprofv3 --kernel-trace durations of the same
command, FETCH_SIZE / WRITE_SIZE per launch from their own --pmc passes (KiB; FETCH doubled per the gfx950 correction of
MI355X_MICROARCH.md, as in tools/summarize_rocprof.py), VGPRs / LDS of the code object."""
import csv
import glob
import json
import statistics
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
tag, label = sys.argv[1], sys.argv[2]
G = ROOT / "gpurun_out"


def one(pattern):
    f = sorted(glob.glob(str(G / pattern)), key=lambda x: Path(x).stat().st_mtime)
    if not f:
        raise SystemExit(f"missing {pattern}")
    return Path(f[-1])


bench = [json.loads(ln) for ln in (G / f"{tag}_bench.log").read_text().splitlines() if ln.startswith("{")]
scheds = []
for b in bench:
    if b["sched"] not in scheds:
        scheds.append(b["sched"])


def groups(rows, per_batch):
    """cell-kernel dispatches in order -> one list per scheduler, warm-up launch dropped"""
    assert len(rows) == per_batch * len(scheds), (len(rows), per_batch, scheds)
    return {s: rows[i * per_batch + 1:(i + 1) * per_batch] for i, s in enumerate(scheds)}


kt = [r for r in csv.DictReader(one(f"{tag}_kt/*/*_kernel_trace.csv").open()) if "rs_cell_kernel" in r["Kernel_Name"]]
kt_launches = len(kt) // len(scheds) - 1
ktg = groups(kt, kt_launches + 1)


def counter(name, sub):
    rows = [r for r in csv.DictReader(one(f"{tag}_{sub}/*/*_counter_collection.csv").open())
            if "rs_cell_kernel" in r["Kernel_Name"] and r["Counter_Name"] == name]
    return groups(rows, len(rows) // len(scheds))


fg, wg = counter("FETCH_SIZE", "fetch"), counter("WRITE_SIZE", "write")
commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
out = {}
for s in scheds:
    q = next(b for b in bench if b["sched"] == s and b["queues"])
    bl = next((b for b in bench if b["sched"] == s and not b["queues"]), None)
    cell_ttis = q["cells"] * q["ttis_per_launch"]
    fetch_b = statistics.mean(float(r["Counter_Value"]) for r in fg[s]) * 1024 * 2
    write_b = statistics.mean(float(r["Counter_Value"]) for r in wg[s]) * 1024
    durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in ktg[s]]
    out[f"sched{s}"] = {
        "workload": q["workload"], "ttis_per_launch": q["ttis_per_launch"],
        "ttis_per_s": q["value"], "us_per_tti_per_cell": q["us_per_tti_per_cell"], "hip_event_ms_per_launch": q["ms_per_launch"],
        "backlogged_ttis_per_s": bl["value"] if bl else None, "backlogged_us_per_tti_per_cell": bl["us_per_tti_per_cell"] if bl else None,
        "kernel_trace_ms_per_launch": [round(d, 3) for d in durs],
        "hbm_read_bytes_per_cell_tti": fetch_b / cell_ttis, "hbm_write_bytes_per_cell_tti": write_b / cell_ttis,
        "hbm_gb_per_s": (fetch_b + write_b) / cell_ttis * q["value"] / 1e9,
        "vgprs": int(ktg[s][0]["VGPR_Count"]), "lds_bytes": int(ktg[s][0]["LDS_Block_Size"]), "scratch": int(ktg[s][0]["Scratch_Size"]),
        "source_hash": q["source_hash"], "commit": commit,
    }
pf = ROOT / "profiles" / "queue_mode.json"
pj = json.loads(pf.read_text()) if pf.exists() else {}
pj[label] = out
pf.write_text(json.dumps(pj, indent=1))
for k, v in out.items():
    print(k, json.dumps({a: (round(b, 2) if isinstance(b, float) else b) for a, b in v.items() if a not in ("workload", "hip_event_ms_per_launch")}))
