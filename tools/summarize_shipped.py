#!/usr/bin/env python3
"""profiles/r05_shipped_shapes.md's tables from the JSON records of tools/sweep_shipped_shapes.py:
    python tools/summarize_shipped.py gpurun_out/r05_shipped.json [gpurun_out/r05_shipped_after.json]"""
import json
import sys
from collections import defaultdict


def table(path):
    d = json.load(open(path))
    by = defaultdict(dict)
    for r in d["results"]:
        by[(r["sched"], r["ues"], r["config"], r["slices"], r["max_slice"])][r["variant"]] = r
    return d, by


def main():
    d, by = table(sys.argv[1])
    after = table(sys.argv[2]) if len(sys.argv) > 2 else None
    print(f"device sources {d['source_hash']}, {d['cells']} cells, 64 RBGs\n")
    for sched in (9, 8, 7, 1):
        keys = sorted(k for k in by if k[0] == sched)
        names = sorted({v for k in keys for v in by[k]} - {"default"})
        hdr = ["configuration", "slices", "UEs", "longest slice", "M TTIs/s", "µs per TTI"] + names
        if after:
            hdr += ["after: M TTIs/s", "autotuned"]
        print(f"### sched {sched}\n\n| " + " | ".join(hdr) + " |\n|" + "---|" * len(hdr))
        for k in keys:
            base = by[k]["default"]
            row = [k[2].replace("/config", "/").replace(".json", ""), str(k[3]), str(k[1]), str(k[4]), f"{base['ttis_per_s'] / 1e6:.2f}", f"{base['us_per_tti']:.2f}"]
            for n in names:
                r = by[k].get(n)
                row.append("—" if not r or "ttis_per_s" not in r else f"{(r['ttis_per_s'] / base['ttis_per_s'] - 1) * 100:+.1f} %")
            if after:
                a = after[1].get(k, {})
                row.append(f"{a['default']['ttis_per_s'] / 1e6:.2f}" if "default" in a and "ttis_per_s" in a["default"] else "—")
                t = a.get("autotune")
                row.append(f"{t['ttis_per_s'] / 1e6:.2f}" if t and "ttis_per_s" in t else "—")
            print("| " + " | ".join(row) + " |")
        print()


if __name__ == "__main__":
    main()
