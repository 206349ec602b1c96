#!/usr/bin/env python3
"""SQ instruction counters of the cell kernel per cell-TTI from a tools/pmc_insts.sh run (gpurun_out/pmc_<tag>) ->
profiles/inst_counts.json, keyed by workload like profiles/traffic.json; bench.py's `roofline_issue` block reads it.

    python tools/summarize_insts.py <tag> [--print-only] [--phase-shares '{"introsort loop": 0.52, ...}']
"""
import collections
import csv
import glob
import json
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
tag = sys.argv[1]
f = max(glob.glob(str(ROOT / f"gpurun_out/pmc_{tag}/*/*_counter_collection.csv")), key=lambda q: Path(q).stat().st_mtime)  # newest run
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "rs_cell_kernel" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
d = json.loads([ln for ln in open(ROOT / f"gpurun_out/pmc_{tag}.log") if ln.startswith("{")][-1])
cfg = d["config"]
n = cfg["cells_per_gpu"] * cfg["ttis_per_step"]
per = {k: sum(v) / len(v) / n for k, v in sorted(acc.items())}
print(tag, "us/TTI/cell %.2f" % d["us_per_tti_per_cell"], " per cell-TTI:", {k: round(v) for k, v in per.items()})
if "--print-only" in sys.argv:
    raise SystemExit(0)
key = f"sched{cfg['sched']}_S{cfg['slices']}_U{cfg['ues']}_R{cfg['rbgs']}_cells{cfg['cells_per_gpu']}"
commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
ent = {"valu": per["SQ_INSTS_VALU"], "salu": per["SQ_INSTS_SALU"], "lds": per["SQ_INSTS_LDS"],
       "active_inst_any_over_wave_cycles": per["SQ_ACTIVE_INST_ANY"] / per["SQ_WAVE_CYCLES"],
       "us_per_tti_per_cell_under_pmc": d["us_per_tti_per_cell"], "kernel": d["kernel"], "commit": commit, "tag": tag,
       "source_hash": d.get("source_hash")}  # device sources of the library that was profiled (bench.py flags a mismatch as stale)
if "--phase-shares" in sys.argv:
    ent["phase_shares"] = json.loads(sys.argv[sys.argv.index("--phase-shares") + 1])
out = ROOT / "profiles" / "inst_counts.json"
allc = json.loads(out.read_text()) if out.exists() else {}
allc[key] = ent
out.write_text(json.dumps(allc, indent=1) + "\n")
print("wrote", out, key)
