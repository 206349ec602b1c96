#!/usr/bin/env python3
"""Study input of the std::sort emulation: MaximizeCell's per-TTI key arrays of the bench workload from the CPU oracle.

    python tools/sort_study.py [R] [n_ttis]     -> tools/microbench/keys_r<R>.bin (u8 [n_ttis][R*S], RBG-major) + statistics

The statistics replay libstdc++'s introsort loop on the keys (pure Python, the serial algorithm of rs_sort_emul.h) and print,
per recursion level, how many sub-ranges are alive and how long they are: what a task-per-wave schedule has to deal with.
Diagnostic tool; the oracle is used as a data source only."""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from conftest import synth_cqi  # noqa: E402
from oracle import oracle_py as O  # noqa: E402
import radiosaber_amd as rs  # noqa: E402


def keys_of_eff(eff):
    tab = np.asarray(O.tables()["eff"] if "eff" in O.tables() else rs.link_tables()["eff"])
    k = np.zeros(eff.shape, np.uint8)
    for c in range(1, 16):
        k[eff == tab[c]] = c
    return k


def dump(R, G, n_ttis, ues=(25,) * 20, seed=5):
    cell = O.Cell(list(ues), R, G, O.SCHED_MAXCELL)
    grids = synth_cqi(seed, ((n_ttis + 39) // 40, cell.U, R), rs.TRACE_CQI_HISTOGRAM)
    g = O.Rng(4242)
    ticks = O.clock_ticks(100, n_ttis)
    cell.set_last_update(0.1)
    out = cell.new_out()
    eff_tab = None
    res = np.zeros((n_ttis, R * len(ues)), np.uint8)
    for n in range(n_ttis):
        if n % 40 == 0:
            cell.set_cqi(grids[n // 40])
        assert cell.step(float(ticks[n]), g.rand(), g.rand(), out) == 0
        if eff_tab is None:
            eff_tab = np.unique(np.concatenate([[0.0], np.asarray(rs.link_tables()["eff"], np.float64)]))
        res[n] = np.searchsorted(eff_tab, out.slice_eff.reshape(-1)).astype(np.uint8)
    return res


def levels(keys):
    """sub-range lengths per recursion level of std::__introsort_loop (threshold 16), serial replay"""
    v = [(-int(k), i) for i, k in enumerate(keys)]  # before(a, b) <=> key(a) > key(b)  <=> -key(a) < -key(b)
    lt = lambda a, b: a[0] < b[0]
    out = []
    cur = [(0, len(v))]
    while cur:
        out.append([l - f for f, l in cur])
        nxt = []
        for f, l in cur:
            a, b, c = f + 1, f + (l - f) // 2, l - 1
            if lt(v[a], v[b]):
                m = b if lt(v[b], v[c]) else (c if lt(v[a], v[c]) else a)
            elif lt(v[a], v[c]):
                m = a
            elif lt(v[b], v[c]):
                m = c
            else:
                m = b
            v[f], v[m] = v[m], v[f]
            first, last, pv = f + 1, l, v[f]
            while True:
                while lt(v[first], pv):
                    first += 1
                last -= 1
                while lt(pv, v[last]):
                    last -= 1
                if not first < last:
                    break
                v[first], v[last] = v[last], v[first]
                first += 1
            cut = first
            for ff, ll in ((f, cut), (cut, l)):
                if ll - ff > 16:
                    nxt.append((ff, ll))
        cur = nxt
    return out


if __name__ == "__main__":
    R = int(sys.argv[1]) if len(sys.argv) > 1 else 25
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    G = 4 if R == 25 else 8
    k = dump(R, G, n)
    outp = ROOT / "tools" / "microbench" / f"keys_r{R}.bin"
    k.tofile(outp)
    print("wrote", outp, k.shape, "key histogram", np.bincount(k.reshape(-1), minlength=16).tolist())
    depth, per_level = [], {}
    for row in k[::4]:
        lv = levels(row)
        depth.append(len(lv))
        for i, ls in enumerate(lv):
            per_level.setdefault(i, []).append(ls)
    print("levels per sort: mean %.2f max %d" % (np.mean(depth), max(depth)))
    for i in sorted(per_level):
        cnt = [len(x) for x in per_level[i]]
        allv = np.concatenate([np.asarray(x) for x in per_level[i]])
        big = [max(x) for x in per_level[i]]
        print(f"level {i}: sorts reaching it {len(cnt)}, sub-ranges mean {np.mean(cnt):.1f} max {max(cnt)}, length mean {allv.mean():.0f} "
              f"max {allv.max()}, largest-per-sort mean {np.mean(big):.0f}, >64: {np.mean([sum(1 for y in x if y > 64) for x in per_level[i]]):.2f}, "
              f">128: {np.mean([sum(1 for y in x if y > 128) for x in per_level[i]]):.2f}")
