#!/usr/bin/env python3
"""What one line of the reference's experiment scripts does (NSDI23-radiosaber-experiments/*/run_*.sh):

    LTE-Sim SingleCellWithI 1 <sched> 1 30 <seed> <duration_s> <config.json>   2> <log>

on the GPU: one cell, backlogged flows, the CQI traces and mapping file of the reference's cqi-traces-noise0 directory,
and the reference's stderr lines (what plot_*.py parse) written to --log.

    python tools/run_experiment.py --sched 9 --seed 0 --duration 12 --config <config.json> \\
        --traces <dir with ue*.log and mapping.config> --log maxcell_pf0.log [--cells 9: one per seed 0..8]

The scheduler's arithmetic is bit-exact; the position of the libc rand() stream at the first scheduled TTI depends on
simulator set-up code outside this path (103 300 draws for the 100-UE configuration) and is taken
from --rand-skip, so runs are statistically, not bitwise, those of the reference unless that number is known.
"""
import argparse
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import radiosaber_amd as rs  # noqa: E402
from radiosaber_amd import logfmt  # noqa: E402

COMMON_SEEDS = (805290992, 749913912, 965326802, 697084729, 1518010490, 56234558, 1511265396, 1412837728, 947674421)

ap = argparse.ArgumentParser()
ap.add_argument("--sched", type=int, default=9, help="the reference's CLI scheduler number: 1, 7, 8, 9, 10, 11")
ap.add_argument("--seed", type=int, default=0, help="index into the reference's nine common seeds")
ap.add_argument("--duration", type=float, default=12.0, help="simulated seconds after the 0.1 s start-up")
ap.add_argument("--config", required=True, help="slice configuration JSON of the experiment directory")
ap.add_argument("--traces", required=True, help="directory with ue<id>.log and the mapping file")
ap.add_argument("--mapping", default="mapping.config")
ap.add_argument("--nb-rbs", type=int, default=512)
ap.add_argument("--rbg-size", type=int, default=8)
ap.add_argument("--rand-skip", type=int, default=0)
ap.add_argument("--log", default="-", help="where the stderr-format lines go ('-' = stdout)")
a = ap.parse_args()

sc = rs.SliceConfig.from_json(a.config)
R = a.nb_rbs // a.rbg_size
n_ttis = int(round(a.duration * 1000))
mapping = rs.read_trace_mapping(Path(a.traces) / a.mapping)
n_traces = int(mapping.max()) + 1
trace, mixed = rs.load_trace_dir(a.traces, n_traces=n_traces, nb_rbs=a.nb_rbs, rbg_size=a.rbg_size)
if mixed:
    raise SystemExit(f"{mixed} RBGs carry different CQI on their PRBs: the batched replay needs uniform RBGs")
U = sc.n_users
b = rs.BatchScheduler(sc, R, a.rbg_size, 1, sched=a.sched, phy_error_draws=True, jit=True)
b.seed(np.array([COMMON_SEEDS[a.seed] if 0 <= a.seed < 9 else COMMON_SEEDS[0]], np.uint32),
       np.array([a.rand_skip], np.int64))
b.set_trace(trace, mapping[np.arange(U) % len(mapping)][None, :].astype(np.int32))
out = sys.stdout if a.log == "-" else open(a.log, "w")
cb = np.zeros(U, np.int64)
cr = np.zeros(U, np.int64)
done = 0
while done < n_ttis:  # logged launches of at most 2 000 TTIs keep the host log small
    n = min(2000, n_ttis - done)
    got = b.run_logged(n)
    lines = logfmt.stderr_lines(got["tbs_bits"][0], got["rbg_to_user"][0], sc.user_to_slice, a.rbg_size,
                                first_ts=100 + done, cum_bytes0=cb, cum_rbs0=cr, nprb=got["nprb"][0],
                                pf_format=a.sched == 1)
    out.write("\n".join(lines) + ("\n" if lines else ""))
    st = b.state()
    cb, cr = st["cum_bytes"][0].copy(), st["cum_rbs"][0].copy()
    done += n
if out is not sys.stdout:
    out.close()
print(f"{n_ttis} TTIs, {U} UEs, sched {a.sched}: per-slice Mbps " +
      " ".join(f"{x:.2f}" for x in (np.asarray(b.slice_bytes(), np.float64) * 8 / 1e6 / a.duration)), file=sys.stderr)
b.close()
