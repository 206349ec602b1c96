#!/usr/bin/env python3
"""Fuzz of the drop-in entry point as round 6 left it (needs a GPU):

    python tools/fuzz_dropin.py [first_seed] [n_seeds] [calls_per_seed]

Per seed one random cell shape (1 … 12 ragged slices incl. empty ones, 6 … 64 RBGs, 1 … 8 PRBs per RBG), one scheduler of
1 / 7 / 8 / 9 / 10 / 101 / 103, and a sequence of calls in which the caller's CQI block changes every few calls and says so through
rs_tti_in.cqi_epoch, the user list changes now and then WITHOUT a new number (the library has to notice), and some seeds pass per-PRB
reports.  Three parties on every call:

    A  a specialised context (rs_ctx_specialize) with cqi_epoch, RS_JIT_SELFCHECK=2 (its first calls run beside the built-in kernel)
    B  a built-in context that is handed the true reports with cqi_epoch = 0        (every output field and the slice state: A == B)
    O  the oracle, on the calls that schedule every user (schedulers other than NVS)  (A == O)
"""
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import radiosaber_amd as rs  # noqa: E402
from conftest import synth_cqi  # noqa: E402
from oracle import oracle_py as O  # noqa: E402

HIST = rs.TRACE_CQI_HISTOGRAM
FIELDS = ("target_rbs", "quota_rbgs", "rbg_to_user", "user_nprb", "user_final_cqi", "user_mcs", "user_tbs_bits")
GRIDS = [(6, 1), (12, 2), (15, 3), (25, 4), (25, 2), (32, 8), (50, 8), (64, 8), (64, 4), (40, 3)]


def one_seed(seed, n_calls):
    rng = np.random.default_rng(seed)
    S = int(rng.integers(1, 13))
    ues = [int(x) for x in rng.integers(0, 41, S)]
    if sum(ues) == 0:
        ues[int(rng.integers(0, S))] = 3
    R, G = GRIDS[int(rng.integers(0, len(GRIDS)))]
    sched = int(rng.choice([1, 7, 8, 9, 10, 101, 103]))
    if sched == 10 and R * S > 2048:
        sched = 9
    w = rng.uniform(0.2, 1.0, S)
    w = (w / w.sum()).tolist()
    sc = rs.SliceConfig(ues, weight=w)
    U = sc.n_users
    u2s = np.asarray(sc.user_to_slice)
    per_prb = bool(rng.random() < 0.25) and sched != 7
    a = rs.TtiScheduler(sc, R, G, sched=sched, jit=True)
    b = rs.TtiScheduler(sc, R, G, sched=sched)
    cell = O.Cell(ues, R, G, sched if sched != 7 else 9, weights=w)
    epoch, cqi, prb, ids = 0, None, None, None
    n_oracle = n_reused = 0
    for it in range(n_calls):
        new_reports = it == 0 or rng.random() < 0.3
        new_users = it == 0 or rng.random() < 0.2
        if new_users:
            if sched == 7:
                sl = int(rng.choice([s for s in range(S) if ues[s] > 0]))
                ids = np.flatnonzero(u2s == sl).astype(np.int32)
            elif rng.random() < 0.6:
                ids = np.arange(U, dtype=np.int32)
            else:
                ids = np.sort(rng.choice(U, int(rng.integers(1, U + 1)), replace=False)).astype(np.int32)
        if new_reports:
            epoch += 1
            cqi = synth_cqi(seed * 997 + it, (U, R), HIST)
            if per_prb:
                prb = np.repeat(cqi, G, axis=1)
                noise = rng.integers(0, 3, prb.shape).astype(np.int64) - 1
                noise[:, ::G] = 0
                prb = np.clip(prb.astype(np.int64) + noise, 1, 15).astype(np.uint8)
        if not new_reports and not new_users:
            n_reused += 1
        avg = rng.choice([1.0, 98000.0, 5e5], U) if rng.random() < 0.1 else rng.uniform(1.0, 2e6, U)
        r0, r1 = int(rng.integers(0, 2**31 - 1)), int(rng.integers(0, 2**31 - 1))
        kw = dict(user_id=ids, rand0=r0, rand1=r1)
        if per_prb:
            ra = a.schedule_tti(None, avg[ids], cqi_prb=prb[ids], cqi_epoch=epoch, **kw)
            rb = b.schedule_tti(None, avg[ids], cqi_prb=prb[ids], **kw)
        else:
            ra = a.schedule_tti(cqi[ids], avg[ids], cqi_epoch=epoch, **kw)
            rb = b.schedule_tti(cqi[ids], avg[ids], **kw)
        for f in FIELDS:
            np.testing.assert_array_equal(getattr(ra, f), getattr(rb, f), err_msg=f"seed {seed} sched {sched} call {it}: {f} (specialised + epoch vs built-in)")
        if sched == 10:
            np.testing.assert_array_equal(ra.upper_rbg, rb.upper_rbg)
            np.testing.assert_array_equal(ra.upper_user, rb.upper_user)
        assert a.slice_offset.tobytes() == b.slice_offset.tobytes(), f"seed {seed} call {it}: slice state"
        if sched != 7 and len(ids) == U:
            # the oracle carries its own slice_rbs_offset_: bring it to where the contexts were BEFORE this call
            cell.set_cqi_prb(prb) if per_prb else cell.set_cqi(cqi)
            out = cell.new_out()
            cell.set_slice_offset(prev_offset) if it else None
            assert cell.allocate(avg, r0, r1, out) == 0
            for f in FIELDS:
                np.testing.assert_array_equal(getattr(ra, f), getattr(out, f), err_msg=f"seed {seed} sched {sched} call {it}: {f} (vs oracle)")
            n_oracle += 1
        prev_offset = a.slice_offset.copy()
    code, msg = a.jit_status()
    assert code == 1, (seed, code, msg)
    a.close()
    b.close()
    return sched, S, U, R, G, per_prb, n_oracle, n_reused


def main():
    os.environ.setdefault("RS_JIT_SELFCHECK", "2")  # every build is checked during its first calls, marked or not
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    calls = int(sys.argv[3]) if len(sys.argv) > 3 else 30
    if not hasattr(O.Cell, "set_slice_offset"):
        raise SystemExit("the oracle binding has no set_slice_offset")
    tot_o = tot_r = 0
    for seed in range(first, first + n):
        sched, S, U, R, G, per_prb, n_o, n_r = one_seed(seed, calls)
        tot_o += n_o
        tot_r += n_r
        print(f"seed {seed}: sched {sched}, {S} slices, {U} UEs, {R} x {G} PRBs{', per-PRB reports' if per_prb else ''}: {calls} calls, "
              f"{n_o} against the oracle, {n_r} served from the device image", flush=True)
    print(f"fuzz_dropin: seeds {first}..{first + n - 1} x {calls} calls bit-exact (specialised + cqi_epoch == built-in on every call; "
          f"{tot_o} calls also == oracle; {tot_r} calls read the device-resident CQI image)")


if __name__ == "__main__":
    main()
