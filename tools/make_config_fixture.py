#!/usr/bin/env python3
"""Distil the slice configurations of the reference's experiment directories (NSDI23-radiosaber-experiments/**/*.json) into
tests/golden/experiment_configs.json: per file the numbers the scheduler constructors read (ues_per_slice and, per slice,
weight / algo_alpha / algo_beta / algo_epsilon / algo_psi; downlink-transport-scheduler.cpp:55-97) plus the scheduler numbers
the directory's run script passes on the command line.  Data only.  Runs in the build container (needs /root/reference)."""
import glob
import json
import re
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import radiosaber_amd as rs  # noqa: E402

REF = Path("/root/reference/NSDI23-radiosaber-experiments")
out = {}
for f in sorted(glob.glob(str(REF / "**" / "*.json"), recursive=True)):
    sc = rs.SliceConfig.from_json(f)
    rel = str(Path(f).relative_to(REF))
    top = REF / rel.split("/")[0]
    scheds = set()
    for sh in glob.glob(str(top / "*.sh")):
        scheds |= {int(x) for x in re.findall(r"SingleCellWithI\s+\d+\s+(\d+)", Path(sh).read_text())}
    out[rel] = {"ues_per_slice": list(sc.ues_per_slice), "weight": list(sc.weight), "algo_alpha": list(sc.algo_alpha),
                "algo_beta": list(sc.algo_beta), "algo_epsilon": list(sc.algo_epsilon), "algo_psi": list(sc.algo_psi),
                "traffic": list(sc.traffic),
                "schedulers_in_run_scripts": sorted(scheds)}
(ROOT / "tests" / "golden" / "experiment_configs.json").write_text(json.dumps(out, indent=0))
print(len(out), "configurations")
