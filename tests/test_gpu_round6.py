"""Round 6 on the GPU: the drop-in surface under the reference's own launch pattern -- many simulators at once (run_backlogged.sh:6-14
starts one OS process per (scheduler, seed, mapping) with `&`) --, run-time builds that are verified before they are trusted (batches
by default, specialised drop-in contexts during their first calls, the mark that lets the next process skip the check), the
device-resident CQI image behind rs_tti_in.cqi_epoch, and ADVICE r05's findings (self-check with the queue model, a rejection that
meets packed grant words, checkpoints of batches configured differently)."""
import json
import os
import subprocess
import sys
import threading
from pathlib import Path

import numpy as np
import pytest

from conftest import synth_cqi

pytestmark = pytest.mark.gpu

ROOT = Path(__file__).resolve().parents[1]
HIST = (152600, 56656, 270880, 2088792, 3509504, 1595568, 4145392, 5295816, 1903424,
        6890232, 4770864, 2842552, 3579624, 96000, 1227696)
FIELDS = ("rbg_to_user", "target_rbs", "quota_rbgs", "user_nprb", "user_final_cqi", "user_mcs", "user_tbs_bits")


def _same(res, out, what):
    for f in FIELDS:
        np.testing.assert_array_equal(getattr(res, f), getattr(out, f), err_msg=f"{what}: {f}")


def _drive(rs, oracle, ts, sched, ues, R, G, n_calls, seed, use_epoch=True, refresh=40, per_prb=False):
    """n_calls RBsAllocation() calls on one context against the oracle; the reports change every `refresh` calls as in the reference
    (CQI_INTERVAL 40) and, use_epoch, the caller says so through cqi_epoch.  Returns the number of calls compared."""
    U = sum(ues)
    S = len(ues)
    cell = oracle.Cell(ues, R, G, sched if sched != 7 else 9, weights=[1.0 / S] * S)
    rng = np.random.default_rng(seed)
    kb = rs.link_tables()["kbps"]
    avg = rng.uniform(1e3, 5e6, U)
    cqi = prb = None
    for it in range(n_calls):
        if it % refresh == 0:
            cqi = synth_cqi(seed * 1000 + it, (U, R), HIST)
            if per_prb:  # reports that differ inside an RBG; the metric reads the first PRB of each RBG
                prb = np.repeat(cqi, G, axis=1)
                noise = rng.integers(0, 3, prb.shape).astype(np.int64) - 1
                noise[:, ::G] = 0
                prb = np.clip(prb.astype(np.int64) + noise, 1, 15).astype(np.uint8)
        epoch = (1 + it // refresh) if use_epoch else 0
        avg = np.maximum(1.0, avg * rng.uniform(0.9, 1.1, U))  # the averages move every TTI, the reports do not
        r0, r1 = int(rng.integers(0, 2**31 - 1)), int(rng.integers(0, 2**31 - 1))
        if sched == 7:
            # NVS: the users of one slice; two consecutive calls serve the same slice now and then (the image is reused only then)
            sl = (it // 2) % S
            lo = int(np.sum(ues[:sl]))
            ids = np.arange(lo, lo + ues[sl])
            res = ts.schedule_tti(cqi[ids], avg[ids], user_id=ids, cqi_epoch=epoch)
            met = kb[cqi[ids]] / ((1 + avg[ids]) / 1000.0)[:, None]
            np.testing.assert_array_equal(res.rbg_to_user, ids[np.argmax(met, axis=0)], err_msg=f"sched 7 call {it}")
            continue
        if per_prb:
            cell.set_cqi_prb(prb)
        else:
            cell.set_cqi(cqi)
        out = cell.new_out()
        assert cell.allocate(avg, r0, r1, out) == 0
        res = ts.schedule_tti(None if per_prb else cqi, avg, r0, r1, cqi_prb=prb if per_prb else None, cqi_epoch=epoch)
        _same(res, out, f"sched {sched} call {it}")
    return n_calls


# ---------------------------------------------------------------------------------------------------------------------------
# VERDICT r05 #1b: eight host threads, each with its own context, mixed schedulers, all at once
# ---------------------------------------------------------------------------------------------------------------------------

def test_eight_threads_each_with_its_own_context_mixed_schedulers(rs, oracle):
    """include/radiosaber_hip.h promises "one context per host thread / HIP stream"; the reference runs its simulators side by side
    (run_backlogged.sh:10-12).  Eight threads x 200 calls each, schedulers 1 / 7 / 8 / 9, built-in and specialised kernels (four of them
    are compiled concurrently -- and self-checked during their first calls), every output of every call against the oracle."""
    ues, R, G, n_calls = [5] * 20, 64, 8, 200  # the shipped exp-fix20slices/5ues shape
    plan = [(9, True), (8, True), (1, True), (7, True), (9, False), (8, False), (1, False), (7, False)]
    start = threading.Barrier(len(plan))
    errors, done = [], [0] * len(plan)

    def worker(i, sched, jit):
        try:
            sc = rs.SliceConfig(ues, weight=[0.05] * 20)
            start.wait(timeout=120)
            ts = rs.TtiScheduler(sc, R, G, sched=sched, jit=jit)  # (create and specialise concurrently too)
            done[i] = _drive(rs, oracle, ts, sched, ues, R, G, n_calls, seed=50 + i)
            if jit:
                code, msg = ts.jit_status()
                assert code == 1, (sched, code, msg)
            ts.close()
        except BaseException as e:  # noqa: BLE001 -- reported by the main thread
            errors.append((i, sched, jit, repr(e)))

    threads = [threading.Thread(target=worker, args=(i, s, j)) for i, (s, j) in enumerate(plan)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=900)
    assert not errors, errors
    assert done == [n_calls] * len(plan)


def test_rs_create_says_when_contexts_outnumber_the_hardware_queues(rs, monkeypatch):
    """profiles/r06_dropin_concurrency.md: the HIP runtime maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues (4 by default);
    the fifth context of a process still works, and rs_last_error() after its creation names the variable to export."""
    monkeypatch.delenv("GPU_MAX_HW_QUEUES", raising=False)
    sc = rs.SliceConfig([2, 2])
    import gc
    gc.collect()  # (contexts other tests dropped without close() are destroyed by their __del__)
    ctxs = [rs.TtiScheduler(sc, 12, 2) for _ in range(5)]
    warned = [i for i, c in enumerate(ctxs) if "GPU_MAX_HW_QUEUES" in c.create_warning]
    assert warned and warned[-1] == 4, [c.create_warning for c in ctxs]        # the fifth at the latest
    assert warned == list(range(warned[0], 5))                                 # ... and every one after the first that was told
    w = ctxs[4].create_warning
    assert "drop-in contexts in this process share 4 hardware queues" in w and "export GPU_MAX_HW_QUEUES=" in w, w
    cqi = np.full((4, 12), 7, np.uint8)
    for c in ctxs:
        assert c.schedule_tti(cqi, np.full(4, 1e5), 1, 2).rbg_to_user.min() >= 0
        c.close()
    if warned[0] == 4:  # nothing else was alive in this process: the closed ones no longer count
        again = rs.TtiScheduler(sc, 12, 2)
        assert "GPU_MAX_HW_QUEUES" not in again.create_warning
        again.close()


# ---------------------------------------------------------------------------------------------------------------------------
# rs_tti_in.cqi_epoch: the device-resident CQI image
# ---------------------------------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("sched,jit", [(9, False), (9, True), (8, True), (1, True), (7, True), (103, False), (10, True)])
def test_cqi_epoch_serves_unchanged_reports_from_the_device_image(rs, oracle, sched, jit):
    """39 calls of 40 see the reports of the call before: the kernel reads the context's image, the caller's block is not touched --
    results are the oracle's on every call, with the image and (control) without it."""
    ues, R, G = [25] * 20, 25, 4
    sc = rs.SliceConfig(ues)
    for use_epoch in (True, False):
        ts = rs.TtiScheduler(sc, R, G, sched=sched, jit=jit)
        _drive(rs, oracle, ts, sched, ues, R, G, 95, seed=7 + sched, use_epoch=use_epoch)
        ts.close()


def test_cqi_epoch_never_trusts_more_than_it_can_check(rs, oracle):
    """The promise covers the BLOCK; the library checks what it can itself: another user list or another user count under the same
    number is a new image, and a block the caller changed WITHOUT bumping the number is -- as promised -- not read.  Against a twin
    context that is handed the true reports with cqi_epoch = 0 on every call (that path is the oracle's, tested above)."""
    ues, R, G = [5] * 20, 64, 8
    U = 100
    sc = rs.SliceConfig(ues)
    ts = rs.TtiScheduler(sc, R, G, sched=9, jit=True)
    twin = rs.TtiScheduler(sc, R, G, sched=9)
    rng = np.random.default_rng(3)
    cqi = synth_cqi(1, (U, R), HIST)
    avg = rng.uniform(1e3, 5e6, U)
    poisoned = np.full_like(cqi, 15)
    all_ids = np.arange(U)
    some = all_ids[::2].copy()
    fewer = all_ids[:60].copy()
    steps = [(all_ids, cqi, cqi, 5, "new number: read"),
             (all_ids, poisoned, cqi, 5, "same number: the image serves the call, the caller's block is not read"),
             (some, cqi[some], cqi[some], 5, "same number, other users: read again"),
             (some, poisoned[some], cqi[some], 5, "... and then served from the image"),
             (fewer, cqi[fewer], cqi[fewer], 5, "same number, other user count: read again"),
             (all_ids, poisoned, poisoned, 6, "new number: the new block is read"),
             (all_ids, cqi, poisoned, 6, "... and kept"),
             (all_ids, cqi, cqi, 0, "no promise: read"),
             (all_ids, poisoned, poisoned, 0, "no promise: read again")]
    for k, (ids, given, truth, epoch, what) in enumerate(steps):
        res = ts.schedule_tti(given, avg[ids], 11 + k, 22 + k, user_id=ids, cqi_epoch=epoch)
        ref = twin.schedule_tti(truth, avg[ids], 11 + k, 22 + k, user_id=ids)
        _same(res, ref, what)
    cell = oracle.Cell(ues, R, G, 9)   # (and the first step against the oracle, so that the twin is anchored)
    cell.set_cqi(cqi)
    out = cell.new_out()
    assert cell.allocate(avg, 11, 22, out) == 0
    t2 = rs.TtiScheduler(sc, R, G, sched=9)
    _same(t2.schedule_tti(cqi, avg, 11, 22, cqi_epoch=9), out, "anchor")
    for t in (ts, twin, t2):
        t.close()


@pytest.mark.parametrize("sched", [9, 8])
def test_cqi_epoch_with_per_prb_reports_and_the_copy_path(rs, oracle, sched, monkeypatch):
    """Per-PRB reports travel as a device copy (not zero-copy): with an unchanged number neither the grid nor the per-PRB block is sent
    again, only the per-call words between them."""
    ues, R, G = [5] * 20, 25, 4
    sc = rs.SliceConfig(ues)
    ts = rs.TtiScheduler(sc, R, G, sched=sched, jit=True)
    _drive(rs, oracle, ts, sched, ues, R, G, 90, seed=21, per_prb=True)
    ts.close()
    monkeypatch.setenv("RS_DROPIN_COPY", "1")  # the staged-copy form of the plain call
    ts = rs.TtiScheduler(sc, R, G, sched=sched)
    _drive(rs, oracle, ts, sched, ues, R, G, 90, seed=22)
    ts.close()


# ---------------------------------------------------------------------------------------------------------------------------
# VERDICT r05 #1c / #2: run-time builds are verified before they are trusted; the mark travels in the cache file
# ---------------------------------------------------------------------------------------------------------------------------

CHILD = r"""
import json, sys
import numpy as np
sys.path.insert(0, %(root)r)
sys.path.insert(0, %(root)r + "/tests")
import radiosaber_amd as rs
from oracle import oracle_py as O
from conftest import synth_cqi
HIST = %(hist)r
mode = %(mode)r
ues, R, G = [5] * 20, 25, 4
out = {}
if mode == "dropin":
    ts = rs.TtiScheduler(rs.SliceConfig(ues), R, G, sched=9, jit=True)
    out["status0"] = ts.jit_status()
    cell = O.Cell(ues, R, G, 9)
    rng = np.random.default_rng(1)
    ok = True
    for it in range(12):
        cqi = synth_cqi(it, (100, R), HIST)
        avg = rng.uniform(1e3, 5e6, 100)
        cell.set_cqi(cqi)
        o = cell.new_out()
        cell.allocate(avg, 5 + it, 6 + it, o)
        r = ts.schedule_tti(cqi, avg, 5 + it, 6 + it)
        ok &= bool((r.rbg_to_user == o.rbg_to_user).all() and (r.user_tbs_bits == o.user_tbs_bits).all() and (r.user_final_cqi == o.user_final_cqi).all())
    out["ok"] = ok
    out["status"] = ts.jit_status()
    out["last_error"] = rs.lib().rs_last_error().decode()
    ts.close()
else:
    b = rs.BatchScheduler(rs.SliceConfig(ues), R, G, 2, sched=9, jit=True)
    b.seed(np.array([3, 4], np.uint32))
    b.synthesize_cqi(9, 10)
    grids = [b.download_cqi_epochs(c) for c in range(2)]
    b.run(300)
    out["status"] = b.jit_status()
    out["kernel"] = b.kernel_name
    st = b.state()
    ok = True
    for c in range(2):
        cell = O.Cell(ues, R, G, 9)
        cell.run_synth(grids[c], 3 + c, 300, log=False)
        ok &= bool((st["cum_bytes"][c] == cell.state()["cum_bytes"]).all() and st["avg_rate"][c].tobytes() == cell.state()["avg_rate"].tobytes())
    out["ok"] = ok
    b.close()
out["stats"] = rs.jit_cache_stats()
print(json.dumps(out))
"""


def _child(mode, cache_dir, env_extra=None):
    env = dict(os.environ, RS_JIT_CACHE_DIR=str(cache_dir), AMD_COMGR_CACHE="0")
    for k in ("RS_JIT_CACHE", "RS_JIT_SELFCHECK", "RS_JIT_EXTRA"):
        env.pop(k, None)
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": str(ROOT), "hist": HIST, "mode": mode}], capture_output=True, text=True, env=env,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads(r.stdout.strip().split("\n")[-1])


def _marks(cache_dir):
    return sorted(f.read_bytes()[-8:].decode() for f in Path(cache_dir).glob("*.rsco"))


def test_specialised_context_checks_itself_once_and_the_next_process_skips_it(rs, tmp_path):
    first = _child("dropin", tmp_path)
    # 12 plain calls: the lean build served them all and passed its 8; the general build, never called, stays unchecked
    assert first["ok"] and first["status"][0] == 1, first
    assert "lean build: verified (8 checked calls agreed with the built-in kernel field by field)" in first["status"][1], first
    assert "general build: 0 checked call(s) agreed" in first["status"][1] and "8 to go" in first["status"][1], first
    assert first["stats"]["misses"] == 2 and _marks(tmp_path) == ["UNCHECKD", "VERIFIED"], (first, _marks(tmp_path))
    second = _child("dropin", tmp_path)
    assert second["ok"] and second["stats"] == {"hits": 2, "misses": 0, "stores": 0, "rejected": 0}
    assert second["status"][0] == 1 and "lean build: carries the self-check mark" in second["status"][1], second
    assert "general build: 0 checked call(s) agreed" in second["status"][1], second  # still unchecked: its first call will be


def test_batch_builds_are_verified_by_default_and_the_mark_travels(rs, tmp_path):
    first = _child("batch", tmp_path)
    assert first["ok"] and first["kernel"] == "rs_cell_kernel_jit"
    assert first["status"][0] == 1 and "selfcheck over 256 TTIs" in first["status"][1] and "general and lean builds agree" in first["status"][1], first
    assert _marks(tmp_path) == ["VERIFIED", "VERIFIED"]
    second = _child("batch", tmp_path)
    assert second["ok"] and second["stats"]["hits"] == 2 and second["stats"]["misses"] == 0
    assert second["status"][0] == 1 and "carries the self-check mark" in second["status"][1], second
    # opting out: nothing is checked, nothing is marked
    d2 = tmp_path / "optout"
    third = _child("batch", d2, {"RS_JIT_SELFCHECK": "0"})
    assert third["ok"] and _marks(d2) == ["UNCHECKD", "UNCHECKD"] and "selfcheck" not in third["status"][1]


def test_a_wrong_object_in_the_cache_is_caught_by_the_process_that_loads_it(rs, tmp_path):
    """A process that opted out of the check leaves an unmarked (and, here, deliberately wrong) object behind; the next process loads it
    from the cache -- no hiprtc run -- checks it because it carries no mark, rejects it, unlinks it, and serves the batch from the
    built-in kernels with the oracle's results."""
    wrong = {"RS_JIT_EXTRA": "-DRS_FAULT_INJECT_JIT"}
    first = _child("batch", tmp_path, dict(wrong, RS_JIT_SELFCHECK="0"))
    assert not first["ok"], "the fault injection does not bite"
    assert _marks(tmp_path) == ["UNCHECKD", "UNCHECKD"]
    second = _child("batch", tmp_path, wrong)
    assert second["stats"]["hits"] >= 1 and second["stats"]["misses"] == 0, second
    assert second["ok"] and second["status"][0] == -2 and "differs from the" in second["status"][1], second
    assert second["kernel"] != "rs_cell_kernel_jit"
    assert len(_marks(tmp_path)) == 1, "the rejected object is still in the cache"


def test_specialised_context_drops_a_wrong_build_on_its_first_call(rs, oracle, monkeypatch, tmp_path):
    """-DRS_FAULT_INJECT_DIRECT: the one-TTI kernel reports a byte more in every transport block.  The first call runs beside the
    built-in kernel, differs, and is served by it: every call returns the oracle's numbers, the status is -2 and names the field."""
    monkeypatch.setenv("RS_JIT_EXTRA", "-DRS_FAULT_INJECT_DIRECT")
    monkeypatch.setenv("RS_JIT_CACHE_DIR", str(tmp_path))
    ues, R, G = [5] * 20, 25, 4
    sc = rs.SliceConfig(ues)
    ts = rs.TtiScheduler(sc, R, G, sched=9, jit=True)
    assert ts.jit_status()[0] == 1
    assert len(list(tmp_path.glob("*.rsco"))) == 2
    _drive(rs, oracle, ts, 9, ues, R, G, 30, seed=5)
    code, msg = ts.jit_status()
    assert code == -2 and "user_tbs_bits" in msg and "checked call 1" in msg and "built-in kernel serves" in msg, (code, msg)
    assert not list(tmp_path.glob("*.rsco")), "the rejected builds are still in the cache"
    ts.close()
    # the same wrong build without the check really returns wrong numbers (the injection bites)
    monkeypatch.setenv("RS_JIT_SELFCHECK", "0")
    monkeypatch.setenv("RS_JIT_EXTRA", "-DRS_FAULT_INJECT_DIRECT -DRS_UNCHECKED_TWIN")  # (another key: the first one is rejected for this process)
    ts = rs.TtiScheduler(sc, R, G, sched=9, jit=True)
    with pytest.raises(AssertionError):
        _drive(rs, oracle, ts, 9, ues, R, G, 3, seed=5)
    ts.close()


@pytest.mark.parametrize("sched", [9, 8, 1, 7])
def test_specialised_context_check_covers_every_scheduler(rs, oracle, sched, monkeypatch):
    """RS_JIT_SELFCHECK=2: the first calls are checked even when the build carries the mark; 40 calls, all the oracle's, status says
    how many agreed."""
    monkeypatch.setenv("RS_JIT_SELFCHECK", "2")
    ues, R, G = [25] * 20, 25, 4
    ts = rs.TtiScheduler(rs.SliceConfig(ues), R, G, sched=sched, jit=True)
    _drive(rs, oracle, ts, sched, ues, R, G, 40, seed=9)
    code, msg = ts.jit_status()
    assert code == 1 and "verified (8 checked calls agreed with the built-in kernel" in msg, (code, msg)
    ts.close()


# ---------------------------------------------------------------------------------------------------------------------------
# ADVICE r05
# ---------------------------------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("sched", [9, 8, 7, 1])
def test_selfcheck_with_the_queue_model_leaves_no_trace(rs, oracle, sched, monkeypatch):
    """the self-check's snapshot left out the bearers' queues, averages and counters, so a checked queue-model batch
    ran 2-3 trial launches ahead of itself.  Now they are part of the snapshot and of the comparison: checked batches (general build at
    the first launch, lean build at the first long one) end where the oracle ends."""
    from test_gpu_queues import _run_case
    monkeypatch.setenv("RS_JIT_SELFCHECK", "2")      # every build, marked or not
    monkeypatch.setenv("RS_JIT_LEAN_MIN_TTIS", "50")
    kinds = ["B-", "Q-", "QQ", "BQ", "Q-", "QQ"]
    custom = sched in (9, 8, 7)
    alpha = [0, 1, 1, 1, 0, 1] if custom else [0] * 6
    beta = [0, 0, 1, 1, 0, 0] if custom else [0] * 6
    _run_case(rs, oracle, sched, [9, 12, 7, 10, 3, 11], kinds, alpha, beta, 25, 4, n_cells=3, launches=[30, 70, 47], jit=True, seed=177 + sched,
              logged=False)
    _run_case(rs, oracle, sched, [9, 12, 7, 10, 3, 11], kinds, alpha, beta, 25, 4, n_cells=2, launches=[40, 60], jit=True, seed=178 + sched,
              logged=True)


def test_a_wrong_lean_build_is_dropped_alone(rs, oracle, monkeypatch):
    """-DRS_FAULT_INJECT_LEAN breaks the lean build only.  The general build passes at the first (short) launch and serves it; the lean
    build is checked when the first long launch wants it -- after launches that left PACKED grant words behind --, disagrees and is
    dropped alone: the general build keeps serving, status stays 1 and says so, results are the oracle's."""
    monkeypatch.setenv("RS_JIT_EXTRA", "-DRS_FAULT_INJECT_LEAN")
    ues, R, G = [25] * 20, 25, 4
    b = rs.BatchScheduler(rs.SliceConfig(ues), R, G, 3, sched=9, jit=True, selfcheck=1)
    b.seed(np.arange(3, dtype=np.uint32) + 5)
    b.synthesize_cqi(11, 24)
    grids = [b.download_cqi_epochs(c) for c in range(3)]
    b.run(30)
    assert b.jit_status()[0] == 1
    b.run(300)
    code, msg = b.jit_status()
    assert code == 1 and "lean build" in msg and "dropped" in msg and "general build" in msg, (code, msg)
    assert b.kernel_name == "rs_cell_kernel_jit"
    b.run(270)
    st = b.state()
    b.close()
    for c in range(3):
        cell = oracle.Cell(ues, R, G, 9)
        cell.run_synth(grids[c], 5 + c, 600, log=False)
        ost = cell.state()
        np.testing.assert_array_equal(st["cum_bytes"][c], ost["cum_bytes"])
        np.testing.assert_array_equal(st["cum_rbs"][c], ost["cum_rbs"])
        assert st["avg_rate"][c].tobytes() == ost["avg_rate"].tobytes()


def test_a_rejection_that_meets_packed_grant_words(rs, oracle, monkeypatch):
    """A wrong GENERAL build rejected from a state a shape-specialised kernel left (here: a checkpoint of a healthy batch): the built-in
    kernels that take over must read plain bytes, not bytes | PRBs << 20 | counted (the first EWMA after the rejection was
    wrong).  150 + 250 TTIs end where the oracle ends."""
    ues, R, G, n1, n2 = [25] * 20, 25, 4, 150, 250
    grids = synth_cqi(61, (2, (n1 + n2 + 39) // 40, 500, R), HIST)
    seeds = np.array([8, 9], np.uint32)

    def make(threads=0):
        b = rs.BatchScheduler(rs.SliceConfig(ues), R, G, 2, sched=9, jit=True, threads_per_cell=threads)
        b.seed(seeds)
        b.upload_cqi_epochs(grids)
        return b
    a = make()
    a.run(n1)
    blob = a.checkpoint()
    a.close()
    monkeypatch.setenv("RS_JIT_EXTRA", "-DRS_FAULT_INJECT_JIT")
    b = make(threads=448)  # (a workgroup size no other test rejects this fault-injected build at: a rejection holds for the whole process)
    assert b.kernel_name == "rs_cell_kernel_jit", b.jit_status()
    b.restore(blob)
    b.run(n2)
    code, msg = b.jit_status()
    assert code == -2, (code, msg)
    st = b.state()
    b.close()
    for c in range(2):
        cell = oracle.Cell(ues, R, G, 9)
        cell.run_synth(grids[c], int(seeds[c]), n1 + n2, log=False)
        ost = cell.state()
        np.testing.assert_array_equal(st["cum_bytes"][c], ost["cum_bytes"])
        assert st["avg_rate"][c].tobytes() == ost["avg_rate"].tobytes(), "the PF averages after the rejection differ"


def test_checkpoint_of_a_differently_configured_batch_is_refused(rs):
    """the header carried the shape only.  Same S, U, R, G, scheduler and cells -- but another split of the users,
    other weights, another refresh period, error-model draws on: refused, with a message that says why."""
    def make(ues=(5, 5, 5, 5), weight=None, **kw):
        b = rs.BatchScheduler(rs.SliceConfig(list(ues), weight=list(weight) if weight else []), 12, 2, 2, sched=9, **kw)
        b.seed(np.array([1, 2], np.uint32))
        b.synthesize_cqi(1, 8)
        return b
    a = make()
    a.run(50)
    blob = a.checkpoint()
    a.close()
    for other in (make(ues=(4, 6, 5, 5)), make(weight=(0.4, 0.2, 0.2, 0.2)), make(cqi_refresh=20), make(phy_error_draws=True), make(first_tti=0)):
        with pytest.raises(rs.RadioSaberError) as e:
            other.restore(blob)
        assert "configured differently" in str(e.value), str(e.value)
        other.close()
    same = make(threads_per_cell=128, jit=True)  # workgroup size and kernel family are free
    same.restore(blob)
    assert same.ttis_done == 50
    same.run(10)
    same.close()


# ---------------------------------------------------------------------------------------------------------------------------
# the bench line proves its own run
# ---------------------------------------------------------------------------------------------------------------------------

def _bench(args, env_extra=None, timeout=900):
    env = dict(os.environ)
    for k in ("RS_JIT_EXTRA", "RS_JIT", "RS_JIT_SELFCHECK"):
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, str(ROOT / "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)


def test_bench_line_carries_a_parity_sample_and_fails_when_it_is_wrong():
    common = ["--steps", "1", "--warmup", "0", "--cells", "64", "--ttis", "400", "--no-r64", "--no-streamed", "--no-cells1024",
              "--cpu-baseline-seconds", "0.5"]
    r = _bench(common)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    ps = d["parity_sample"]
    assert ps["cells"] == 32 and ps["ttis"] == 2000 and ps["bit_exact"] is True and ps["kernel"] == "rs_cell_kernel_jit", ps
    assert d["occupancy"]["waves_per_cu"] == 64 / d["compute_units"] * 8 and d["cpu_baseline"]["value"] > 0
    # a deliberately wrong kernel (and the self-check switched off, or it would have been dropped): the line says false, the run fails
    r = _bench(common + ["--allow-variant"], {"RS_JIT_EXTRA": "-DRS_FAULT_INJECT_JIT", "RS_JIT_SELFCHECK": "0"})
    assert r.returncode == 3, (r.returncode, r.stderr[-2000:])
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["parity_sample"]["bit_exact"] is False and "cum_bytes" in d["parity_sample"]["differs_in"]
    assert "DIFFER on the parity sample" in r.stderr
    # with the default self-check the same wrong build never serves: the batch falls back, and bench.py refuses a headline on the built-in kernels
    r = _bench(common + ["--allow-variant"], {"RS_JIT_EXTRA": "-DRS_FAULT_INJECT_JIT"})
    assert r.returncode != 0


@pytest.mark.parametrize("seed", [3, 4, 5, 6])
def test_drop_in_fuzz_seeds(rs, oracle, seed, monkeypatch):
    """tools/fuzz_dropin.py (random shapes and schedulers; the CQI block and the user list change independently; per-PRB reports on some
    seeds): a specialised, self-checked context with cqi_epoch == a built-in context without == the oracle where every user is listed."""
    import importlib.util
    monkeypatch.setenv("RS_JIT_SELFCHECK", "2")
    spec = importlib.util.spec_from_file_location("fuzz_dropin", ROOT / "tools" / "fuzz_dropin.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.one_seed(seed, 24)
