#!/usr/bin/env python3
"""TTIs/s of the reference's OWN experiment shapes (tests/golden/experiment_configs.json: 64 RBGs, 53-600 UEs in 5-20 ragged
slices, weights and algo parameters as shipped; every flow backlogged) x schedulers 1 / 7 / 8 / 9, 512 cells per GPU, the
shape-specialised kernel (lean build) -- and, with --ab, the alternatives the JIT's rule table chooses between, one RS_JIT_EXTRA
(or RS_JIT_SCHED_STRATEGY) variant per build, same box, same process.  Run ON THE GPU BOX:

    python tools/sweep_shipped_shapes.py --out gpurun_out/r05_shipped.json [--ab] [--cells 512] [--only exp-fixranues]

ref: NSDI23-radiosaber-experiments/exp-fix20slices/run_exps.sh, exp-customization/run_backlogged.sh:6-14 (one process per seed there).
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

# variant name -> environment for the run-time compiler
AB = {
    9: {"no_spec": {"RS_JIT_EXTRA": "-DRS_NO_SPEC"}, "hold_always": {"RS_JIT_EXTRA": "-DRS_HOLD_ALWAYS"},
        "p3_block8": {"RS_JIT_EXTRA": "-DRS_P3_BLOCK=8"}, "p3_block32": {"RS_JIT_EXTRA": "-DRS_P3_BLOCK=32"},
        "sched_default": {"RS_JIT_SCHED_STRATEGY": "default"}, "threads256": {"threads": 256}},
    8: {"no_spec": {"RS_JIT_EXTRA": "-DRS_NO_SPEC"}, "hold_always": {"RS_JIT_EXTRA": "-DRS_HOLD_ALWAYS"},
        "p3_block16": {"RS_JIT_EXTRA": "-DRS_P3_BLOCK=16"}, "sched_ilp": {"RS_JIT_SCHED_STRATEGY": "iterative-ilp"},
        "threads256": {"threads": 256}},
    7: {"no_early17": {"RS_JIT_EXTRA": "-DRS_NO_EARLY17"}, "whole_slice32": {"RS_JIT_EXTRA": "-DRS_NVS_WHOLE_SLICE=32"},
        "threads256": {"threads": 256}},
    1: {"no_early17": {"RS_JIT_EXTRA": "-DRS_NO_EARLY17"}, "pf1_always": {"RS_JIT_EXTRA": "-DRS_PF1_ALWAYS"},
        "threads256": {"threads": 256}},
}


def distinct_shapes(cfgs, only=None):
    """one configuration per distinct ues_per_slice (the PF one when a directory ships both config-pf and config-mt)"""
    by = {}
    for name in sorted(cfgs):
        c = cfgs[name]
        if only and only not in name:
            continue
        if any(c["algo_alpha"]):
            continue  # customised slices run with the queue model (tools/bench_queue_mode.py)
        k = tuple(c["ues_per_slice"])
        if k not in by or ("config-pf" in name and "config-pf" not in by[k]):
            by[k] = name
    return sorted(by.values(), key=lambda n: (sum(cfgs[n]["ues_per_slice"]), n))


def measure(rs, sc, sched, cells, threads, ttis, launches, autotune=False):
    b = rs.BatchScheduler(sc, 64, 8, cells, sched=sched, threads_per_cell=threads, jit=True, cqi_epoch_wrap=True, autotune=autotune)
    try:
        code, msg = b.jit_status()
        if code != 1:
            return {"error": f"jit status {code}: {msg}"}
        b.seed((np.arange(cells, dtype=np.uint64) * 2654435761 + 805290992).astype(np.uint32))
        b.synthesize_cqi(0x5AB3, 128)  # 128 epochs (5 120 TTIs) cycling: 512 cells x 128 grids is far beyond the Infinity Cache
        b.prepare_launch(ttis)
        b.run(ttis)
        ms = b.run_timed(ttis, launches)
        code, msg = b.jit_status()
        return {"ttis_per_s": cells * ttis / (float(np.mean(ms)) / 1e3), "us_per_tti": float(np.mean(ms)) * 1e3 / ttis,
                "ms": [round(float(x), 3) for x in ms], "jit_msg": msg, "autotune": b.autotune_report()[1] if autotune else None}
    finally:
        b.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/r05_shipped.json")
    ap.add_argument("--cells", type=int, default=512)
    ap.add_argument("--ab", action="store_true")
    ap.add_argument("--autotune", action="store_true", help="one more run per (shape, scheduler 8 / 9) with rs_batch_config.autotune")
    ap.add_argument("--only", default=None)
    ap.add_argument("--scheds", default="9,8,7,1")
    args = ap.parse_args()
    import radiosaber_amd as rs
    cfgs = json.loads((ROOT / "tests" / "golden" / "experiment_configs.json").read_text())
    names = distinct_shapes(cfgs, args.only)
    out = {"source_hash": rs.device_source_hash(), "cells": args.cells, "rbgs": 64, "results": []}
    for name in names:
        c = cfgs[name]
        sc = rs.SliceConfig(c["ues_per_slice"], weight=c["weight"], algo_epsilon=c["algo_epsilon"], algo_psi=c["algo_psi"])
        for sched in [int(x) for x in args.scheds.split(",")]:
            ttis = 2000 if sched == 9 else 6000
            variants = [("default", {})] + (sorted(AB[sched].items()) if args.ab else [])
            if args.autotune and sched in (8, 9):
                variants.append(("autotune", {"autotune": True}))
            for vname, env in variants:
                for k in ("RS_JIT_EXTRA", "RS_JIT_SCHED_STRATEGY"):
                    os.environ.pop(k, None)
                for k, v in env.items():
                    if k not in ("threads", "autotune"):
                        os.environ[k] = v
                t0 = time.time()
                try:
                    r = measure(rs, sc, sched, args.cells, env.get("threads", 0), ttis, 3, autotune=bool(env.get("autotune")))
                except Exception as e:  # a variant that does not build or fit is a result too
                    r = {"error": str(e)[:300]}
                r.update({"config": name, "slices": sc.n_slices, "ues": sc.n_users, "max_slice": max(c["ues_per_slice"]), "sched": sched,
                          "variant": vname, "wall_s": round(time.time() - t0, 1)})
                out["results"].append(r)
                print(f"{name:55s} S{sc.n_slices:2d} U{sc.n_users:3d} sched {sched} {vname:14s} "
                      + (f"{r['ttis_per_s'] / 1e6:8.2f} M TTIs/s {r['us_per_tti']:7.2f} us" if "ttis_per_s" in r else r["error"]), flush=True)
                Path(args.out).parent.mkdir(parents=True, exist_ok=True)
                Path(args.out).write_text(json.dumps(out, indent=1))
    for k in ("RS_JIT_EXTRA", "RS_JIT_SCHED_STRATEGY"):
        os.environ.pop(k, None)


if __name__ == "__main__":
    main()
