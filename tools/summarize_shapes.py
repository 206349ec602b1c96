#!/usr/bin/env python3
"""tools/profile_shapes.sh -> profiles/traffic.json + profiles/inst_counts.json, one keyed entry per workload, each with the
source hash of the device code that was profiled (bench.py compares it with the library's and says "stale" when they differ).

    python tools/summarize_shapes.py [--phase-shares-headline '{"introsort loop": 0.55, ...}']

FETCH_SIZE / WRITE_SIZE are KiB, collected in separate passes; FETCH is doubled per the gfx950 correction of
MI355X_MICROARCH.md (wide coalesced reads are tallied at half their bytes), as in tools/summarize_rocprof.py."""
import collections
import csv
import glob
import json
import statistics
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
G = ROOT / "gpurun_out"
ROUND = next((a for a in sys.argv[1:] if a.startswith("r") and a[1:].isdigit()), "r05")
commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
tags = sorted({Path(p).name[3:-len("_insts.log")] for p in glob.glob(str(G / "ps_*_insts.log"))})
tf, inf = ROOT / "profiles" / "traffic.json", ROOT / "profiles" / "inst_counts.json"
traffic = json.loads(tf.read_text()) if tf.exists() else {}
insts = json.loads(inf.read_text()) if inf.exists() else {}


def counters(tag, sub):
    f = max(glob.glob(str(G / f"ps_{tag}_{sub}" / "*" / "*_counter_collection.csv")), key=lambda q: Path(q).stat().st_mtime)
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "rs_cell_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


def line(tag, sub):
    return json.loads([ln for ln in (G / f"ps_{tag}_{sub}.log").read_text().splitlines() if ln.startswith("{")][-1])


for tag in tags:
    d = line(tag, "insts")
    cfg = d["config"]
    key = f"sched{cfg['sched']}_S{cfg['slices']}_U{cfg['ues']}_R{cfg['rbgs']}_cells{cfg['cells_per_gpu']}"
    if cfg.get("cqi_refresh", 40) != 40:
        key += f"_refresh{cfg['cqi_refresh']}"  # bench.py's key for a batch that is not on the reference's 40-TTI report interval
    n = cfg["cells_per_gpu"] * cfg["ttis_per_step"]
    fetch, write, ic = counters(tag, "fetch")["FETCH_SIZE"], counters(tag, "write")["WRITE_SIZE"], counters(tag, "insts")
    fb, wb = statistics.mean(fetch) * 1024 * 2, statistics.mean(write) * 1024
    hashes = {line(tag, s).get("source_hash") for s in ("fetch", "write", "insts")}
    assert len(hashes) == 1, (tag, hashes)
    traffic[key] = {"kernel": d["kernel"], "launches_profiled": len(fetch), "FETCH_SIZE_KiB_mean": statistics.mean(fetch),
                    "WRITE_SIZE_KiB_mean": statistics.mean(write), "hbm_read_bytes_per_launch": fb, "hbm_write_bytes_per_launch": wb,
                    "hbm_bytes_per_launch": fb + wb, "ttis_per_launch": cfg["ttis_per_step"], "hbm_bytes_per_cell_tti": (fb + wb) / n,
                    "ttis_per_s_under_pmc": line(tag, "fetch")["value"], "commit": commit, "round": ROUND, "tag": tag,
                    "source_hash": d.get("source_hash"),
                    "note": "FETCH_SIZE doubled per the gfx950 correction; separate --pmc passes (tools/profile_shapes.sh)"}
    per = {k: sum(v) / len(v) / n for k, v in ic.items()}
    ent = {"valu": per["SQ_INSTS_VALU"], "salu": per["SQ_INSTS_SALU"], "lds": per["SQ_INSTS_LDS"],
           "active_inst_any_over_wave_cycles": per["SQ_ACTIVE_INST_ANY"] / per["SQ_WAVE_CYCLES"],
           "us_per_tti_per_cell_under_pmc": d["us_per_tti_per_cell"], "ttis_per_s_under_pmc": d["value"], "kernel": d["kernel"],
           "commit": commit, "tag": tag, "source_hash": d.get("source_hash")}
    if tag == "s9_r25" and "--phase-shares-headline" in sys.argv:
        ent["phase_shares"] = json.loads(sys.argv[sys.argv.index("--phase-shares-headline") + 1])
    elif key in insts and "phase_shares" in insts[key] and insts[key].get("source_hash") == d.get("source_hash"):
        ent["phase_shares"] = insts[key]["phase_shares"]
    insts[key] = ent
    print(f"{key:34s} {d['value'] / 1e6:7.2f} M TTIs/s  HBM {(fb + wb) / n:7.1f} B/cell-TTI  VALU {per['SQ_INSTS_VALU']:8.0f}  SALU {per['SQ_INSTS_SALU']:7.0f}  "
          f"LDS {per['SQ_INSTS_LDS']:6.0f}  active {ent['active_inst_any_over_wave_cycles']:.3f}  hash {d.get('source_hash')}")
tf.write_text(json.dumps(traffic, indent=1))
inf.write_text(json.dumps(insts, indent=1) + "\n")
