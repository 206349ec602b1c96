#!/usr/bin/env python3
"""TTIs/s of the queue model in batches: the reference's
NSDI23-radiosaber-experiments/exp-customization/exp-customize-20slices/config.json -- 20 slices, 194 UEs: backlogged, one / two
InternetFlow bearers, 1 280 kbit/s video -- as a batch of independent cells under the scheduler numbers of its run script
(run_customize.sh: 1, 7, 9), 64 RBGs of 8 PRBs like the script's 100 MHz grid.

Traffic: rs_internet_flow_arrivals (the reference's InternetFlow process) and the recorded video trace
(tests/golden/video_foreman_1280k.json), generated for `--distinct` cells and repeated over the batch (the CQI grids and rand()
seeds differ per cell, so the trajectories still differ).  Timing: HIP events around every launch (rs_batch_run_timed).

  python tools/bench_queue_mode.py [--cells 512] [--ttis 1000] [--launches 3] [--sched 9,7,1] [--distinct 8] [--no-jit]
Prints one JSON line per scheduler (and a backlogged line of the same shape for comparison with --with-backlogged)."""
import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import radiosaber_amd as rs  # noqa: E402

GOLDEN = ROOT / "tests" / "golden"


def build_bursts(cfg, sc, n_distinct, n_ttis):
    video = json.loads((GOLDEN / "video_foreman_1280k.json").read_text())
    U, u2s = sc.n_users, sc.user_to_slice
    stop = 0.1 + n_ttis / 1000.0 + 0.01
    per_cell = []
    for c in range(n_distinct):
        d = {}
        for u in range(U):
            tr = cfg["traffic"][u2s[u]]
            for j in range(int(tr["internet_flow"])):
                rate = tr["if_bitrate"][j] / cfg["ues_per_slice"][u2s[u]]  # single-cell-with-interference.h:415-416
                d[(u, j)] = rs.internet_flow_arrivals(rate, 0.1, stop, 1000 * c + 2 * u + j)
            if int(tr["video_app"]):
                t, ts = 0.1, []
                k = 0
                n = len(video["bytes"])
                while True:  # TraceBased::Send: the next frame TimeToSend * 0.001 after this one; the trace repeats
                    if k:
                        dt = video["time_ms"][k % n] - video["time_ms"][(k - 1) % n]
                        t = (dt if dt > 0 else 40) * 0.001 + t
                    if t >= stop:
                        break
                    ts.append(t)
                    k += 1
                d[(u, 0)] = rs.frames_to_bursts(ts, [video["bytes"][i % n] for i in range(len(ts))])
        per_cell.append(d)
    return per_cell


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cells", type=int, default=512)
    ap.add_argument("--ttis", type=int, default=1000)
    ap.add_argument("--launches", type=int, default=3)
    ap.add_argument("--sched", default="9,7,1")
    ap.add_argument("--distinct", type=int, default=8)
    ap.add_argument("--rbgs", type=int, default=64)
    ap.add_argument("--rbg-size", type=int, default=8)
    ap.add_argument("--no-jit", action="store_true")
    ap.add_argument("--with-backlogged", action="store_true", help="also time the same shape without the queue model")
    a = ap.parse_args()
    if rs.device_count() < 1:
        raise SystemExit("bench_queue_mode.py needs an MI355X: the product has no CPU path")
    cfg = json.loads((GOLDEN / "experiment_configs.json").read_text())["exp-customization/exp-customize-20slices/config.json"]
    sc = rs.SliceConfig(cfg["ues_per_slice"], cfg["weight"], cfg["algo_alpha"], cfg["algo_beta"], cfg["algo_epsilon"],
                        cfg["algo_psi"], cfg["traffic"])
    kinds = sc.bearer_kinds()
    total_ttis = a.ttis * (a.launches + 1)
    t0 = time.time()
    per_cell = build_bursts(cfg, sc, a.distinct, total_ttis)
    bursts = {(c, u, k): v for c in range(a.cells) for (u, k), v in per_cell[c % a.distinct].items()}
    n_bursts = sum(len(v[0]) for v in bursts.values())
    print(f"# arrivals: {n_bursts} bursts for {a.cells} cells ({a.distinct} distinct), {time.time() - t0:.1f} s on the host", file=sys.stderr)
    seeds = (np.arange(a.cells, dtype=np.uint64) * 2654435761 + 805290992) % (2**31 - 1)
    for sched in [int(s) for s in a.sched.split(",")]:
        for queues in ([True, False] if a.with_backlogged else [True]):
            sc_run = sc if queues else rs.SliceConfig(cfg["ues_per_slice"], cfg["weight"])
            b = rs.BatchScheduler(sc_run, a.rbgs, a.rbg_size, a.cells, sched=sched, jit=not a.no_jit)
            if queues:
                b.set_bearers(kinds)
                b.set_arrivals(bursts)
            b.seed(seeds.astype(np.uint32))
            b.synthesize_cqi(0x5AB3, (total_ttis + 39) // 40)
            b.run(a.ttis)  # warm-up launch: code object load, queues fill
            ms = b.run_timed(a.ttis, a.launches)
            jit = b.jit_status()
            sb = b.slice_bytes()
            b.close()
            best = float(np.median(ms))
            print(json.dumps({
                "tool": "bench_queue_mode", "workload": f"exp-customize-20slices x {a.cells} cells x {a.rbgs} RBGs",
                "queues": queues, "sched": sched, "cells": a.cells, "ttis_per_launch": a.ttis, "launches": a.launches,
                "ms_per_launch": [round(float(x), 3) for x in ms], "value": a.cells * a.ttis / (best / 1e3), "unit": "TTIs/s",
                "us_per_tti_per_cell": best * 1e3 / a.ttis, "jit": jit[0], "source_hash": rs.device_source_hash(),
                "slice_mbps_per_cell": [round(float(x) * 8 / 1e6 / ((a.launches + 1) * a.ttis / 1000.0) / a.cells, 3) for x in sb],
            }))
            sys.stdout.flush()


if __name__ == "__main__":
    main()
