"""Write tests/golden/sort_killers.npz: key arrays (1..15) on which libstdc++'s std::sort, as the reference's MaximizeCell /
UpperBound call it (downlink-transport-scheduler.cpp:223-246, 351-376), runs out of its depth limit and heap-sorts a range
(std::__partial_sort, bits/stl_algo.h:1937-1957).  Found by tools/sort_killer.cpp (hill climb over the product's emulation, or
its structured start: runs of one key value at the range's right end); every array is checked there against the real std::sort.

    python tools/make_sort_killers.py            # ~1-2 minutes (one 500-record hill climb), deterministic

With one UE per slice the sort key of (rbg, slice) is cqi[ue = slice][rbg], so an array goes straight through
rs_batch_upload_cqi_epochs / rs_schedule_tti:  MaximizeCell (array index rbg * S + slice)  grid[u][r] = keys[r * S + u];
UpperBound (one sort per slice over its R RBGs)  grid[u][r] = keys[r].
"""
import json
import subprocess
import sys
import tempfile
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
OUT = ROOT / "tests" / "golden" / "sort_killers.npz"

# name: (n, mode, seed, max_steps, runs)   -- sort_killer's arguments; waves = 8 (512 threads), ept by the array length
SPECS = {
    "n500_wg": (500, "wg", 1, 0, "62"),            # runs of 62: a sub-range longer than 64 on every level, two heap sorts
    "n500_wg_long": (500, "wg", 1, 0, "30"),       # one heap sort of 380 records
    "n500_wave": (500, "wave", 1, 0, "230,200"),   # the last levels on single waves, four heap sorts there
    "n500_climb": (500, "any", 1, 20000000, "0"),  # hill climb from a uniform array (no structure put in by hand)
    "n1280_wg": (1280, "wg", 1, 0, "62"),          # the 64-RBG grid: one heap sort of 1 032 records
    "n1280_wave": (1280, "wave", 1, 0, "620,560"),
    "n64_seg": (64, "any", 1, 5000000, "0"),       # UpperBound: one slice's 64 RBGs
    "n64_seg_b": (64, "any", 2, 5000000, "0"),
    "n48_seg": (48, "any", 1, 1000000, "0"),
    "n40_seg": (40, "any", 1, 5000000, "0"),
}


def build_tool(tmp):
    exe = Path(tmp) / "sort_killer"
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", str(exe), str(ROOT / "tools" / "sort_killer.cpp")], check=True)
    return exe


def main():
    arrays, meta = {}, {}
    with tempfile.TemporaryDirectory() as tmp:
        exe = build_tool(tmp)
        for name, (n, mode, seed, steps, runs) in SPECS.items():
            cmd = [str(exe), str(n), mode, str(seed), str(steps), "8", "0", runs]
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                sys.exit(f"{name}: {' '.join(cmd[1:])} -> {r.stdout.strip()} {r.stderr.strip()}")
            head, keys = r.stdout.strip().split("\n")
            _, _, site, heap_calls, nsteps = head.split()
            a = np.array(keys.split(), np.uint8)
            assert a.shape == (n,) and a.min() >= 1 and a.max() <= 15
            arrays[name] = a
            meta[name] = {"args": " ".join(cmd[1:]), "site_512_threads": site, "heap_calls": int(heap_calls), "steps": int(nsteps)}
            print(name, meta[name])
    arrays["meta"] = np.frombuffer(json.dumps(meta, indent=1).encode(), np.uint8)
    np.savez_compressed(OUT, **arrays)
    print("wrote", OUT, OUT.stat().st_size, "bytes")


if __name__ == "__main__":
    main()
