#!/usr/bin/env python3
"""Per-shape kernel durations of tools/dropin_latency under `rocprofv3 --kernel-trace` (the "why the call costs what it costs" table):

    python tools/summarize_dropin_prof.py <..._kernel_trace.csv> [calls per context incl. the 50 warm-up calls, default 1050]

tools/dropin_latency runs its contexts one after the other -- six built-in, six specialised, then five + five with rs_tti_in.cqi_epoch --
so the dispatches of one kernel family, in time order, fall into consecutive groups of `calls` launches.  Prints a markdown table."""
import csv
import sys

import numpy as np

trace = sys.argv[1]
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 1050
rows = list(csv.DictReader(open(trace)))


def groups(pred, names):
    g = sorted((r for r in rows if pred(r["Kernel_Name"])), key=lambda r: int(r["Start_Timestamp"]))
    out = {}
    for i, n in enumerate(names):
        part = g[i * calls + 50:(i + 1) * calls]
        if part:
            d = np.array([int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in part]) / 1e3
            out[n] = (d.mean(), np.median(d), d.min(), d.max(), len(d))
    return out, len(g)


plain = ["sched 9, 100 UEs x 64 RBGs", "sched 9, 500 x 25", "sched 9, 500 x 64", "sched 8, 500 x 25", "sched 1, 500 x 25", "sched 7, 25 of 500 UEs x 25"]
epoch = [n + ", cqi_epoch" for n in plain[:5]]
jit, n_jit = groups(lambda k: k == "rs_cell_kernel_jit", plain + epoch)
bi, n_bi = groups(lambda k: k.startswith("void rs_cell_kernel<"), plain + epoch)
print(f"| context of tools/dropin_latency | built-in kernel, µs (mean / p50 / min / max) | specialised kernel, µs (mean / p50 / min / max) |")
print("|---|---|---|")
for n in plain + epoch:
    f = lambda t: "%.2f / %.2f / %.2f / %.2f" % t[:4] if t else "—"  # noqa: E731
    print(f"| {n} | {f(bi.get(n))} | {f(jit.get(n))} |")
print(f"\n({n_bi} built-in and {n_jit} specialised dispatches in the trace; {calls - 50} timed launches per row after 50 warm-up launches.  A specialised "
      "context whose build still lacks the self-check mark adds up to 8 built-in launches of its shape to the built-in column's groups: run the "
      "tool once before profiling, as tools/experiments/r06/record_r06.sh does.)")
