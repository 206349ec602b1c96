#!/usr/bin/env python3
"""tests/golden/video_foreman_1280k.json: the first 320 frames (12.8 s) of the video trace the reference's customised-slice
experiment streams (src/flows/application/Trace/foreman_H264_1280k.dat; single-cell-with-interference.h:320-364,
video_bitrate 1280) as data: frame index, type, time stamp [ms], size [bytes].  Runs in the build container only."""
import json
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
src = Path("/root/reference/src/flows/application/Trace/foreman_H264_1280k.dat")
rows = [ln.split() for ln in src.read_text().splitlines()[:320]]
fix = {"source": "src/flows/application/Trace/foreman_H264_1280k.dat (first 320 frames)",
       "index": [int(r[0]) for r in rows], "type": "".join(r[1] for r in rows),
       "time_ms": [int(r[2]) for r in rows], "bytes": [int(r[3]) for r in rows]}
(ROOT / "tests" / "golden" / "video_foreman_1280k.json").write_text(json.dumps(fix) + "\n")
print(len(rows), "frames,", sum(fix["bytes"]) * 8 / (fix["time_ms"][-1] / 1000) / 1e3, "kbit/s")
