#!/usr/bin/env python3
"""Disassembly of a run-time compiled kernel (the code object hiprtc produced, through the disk cache): no GPU needed.

    python tools/jit_asm.py S U R G threads sched [--lean] [--streamed] [-o out.s] [--phases]

--phases: instruction counts between the kernel's s_setprio markers (rs_phase_p4.inc raises the issue priority to 1 for the sort
levels, 2 for the counting sort and back to 0), by class -- a static count of one pass through the code, not an execution count."""
import argparse
import re
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("shape", type=int, nargs=6)
    ap.add_argument("--lean", action="store_true")
    ap.add_argument("--streamed", action="store_true")
    ap.add_argument("-o", default=None)
    ap.add_argument("--phases", action="store_true")
    a = ap.parse_args()
    import radiosaber_amd as rs
    import lint_exec_restore as lint
    rs.jit_cache_warm(*a.shape, lean=a.lean, streamed=a.streamed)
    f = rs.jit_cache_file(*a.shape, lean=a.lean, streamed=a.streamed)
    key, code = lint.code_of_cache_file(f)
    text = lint.disassemble(code)
    if a.o:
        Path(a.o).write_text("; " + key.replace("\n", " ") + "\n" + text)
    if a.phases:
        prio, counts, order = None, {}, []
        for line in text.split("\n"):
            s = line.split("//")[0].strip()
            if not s or s.endswith(":") or re.match(r"^[0-9a-f]+ <", s):
                continue
            op = s.split()[0]
            m = re.match(r"s_setprio\s+(\d+)", s)
            if m:
                prio = f"after s_setprio {m.group(1)} #{len(order)}"
                order.append(prio)
                counts[prio] = {}
                continue
            if prio is None:
                continue
            cls = ("valu" if op.startswith("v_") else "lds" if op.startswith("ds_") else "salu" if op.startswith("s_") else
                   "vmem" if op.startswith(("global_", "scratch_", "flat_", "buffer_")) else "other")
            counts[prio][cls] = counts[prio].get(cls, 0) + 1
        for p in order:
            c = counts[p]
            print(f"{p:28s} total {sum(c.values()):5d}  " + "  ".join(f"{k} {v}" for k, v in sorted(c.items())))
    if not a.o and not a.phases:
        print(text)


if __name__ == "__main__":
    main()
