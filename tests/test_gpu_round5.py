"""GPU tests added in round 5 (through the C ABI, bit-exact against the oracle): the rewritten counting sorts at two, three and four
chunks per wave and in the LDS form, the NVS carve rule keyed on the longest slice window (ragged batches), the built-in NVS
threshold, the batch autotune (state untouched, results identical), the drop-in call's polled completion word."""
import os

import numpy as np
import pytest

from conftest import synth_cqi
from test_gpu_parity import HIST, _check_batch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("jit", [False, True])
@pytest.mark.parametrize("ues,R,G,threads", [
    ([3] * 40, 25, 4, 0),      # 1 000 sort records: two chunks per wave
    ([5] * 20, 64, 8, 0),      # 1 280: three (the as-shipped grid)
    ([2] * 32, 64, 8, 0),      # 2 048: four, every chunk full
    ([2] * 30, 61, 8, 0),      # 1 830: four, the last chunk partial
    ([5] * 20, 64, 8, 256),    # 1 280 records on four waves: five chunks per wave -> the LDS form (lanes = chunks prefix)
    ([1] * 64, 64, 8, 128),    # 4 096 records: 64 chunks, the LDS form's widest table
])
def test_counting_sort_forms(rs, oracle, ues, R, G, threads, jit):
    if threads == 128 and not jit:
        pytest.skip("one shape of the LDS form per build is enough")
    _check_batch(rs, oracle, 9, ues, R, G, n_cells=2, n_ttis=85, threads=threads, jit=jit, seed=5)


def test_counting_sort_round4_form_still_agrees(rs, oracle, monkeypatch):
    """-DRS_COUNTING_SORT_V1 keeps the previous form for A/B runs: same results."""
    monkeypatch.setenv("RS_JIT_EXTRA", "-DRS_COUNTING_SORT_V1")
    _check_batch(rs, oracle, 9, [5] * 20, 64, 8, n_cells=2, n_ttis=60, jit=True, seed=6)


@pytest.mark.parametrize("jit", [False, True])
@pytest.mark.parametrize("ues", [[70, 5, 5, 5, 5, 5, 5, 5, 5, 5], [10, 90, 10, 10], [40, 41, 39]])
def test_nvs_ragged_batch_with_one_long_slice(rs, oracle, ues, jit):
    """The average slice is short, one window is longer than 64 users: split runs (round 5: rs_carve keyed on the longest window)."""
    _check_batch(rs, oracle, 7, ues, 25, 4, n_cells=3, n_ttis=130, jit=jit, seed=7)
    _check_batch(rs, oracle, 7, ues, 64, 8, n_cells=2, n_ttis=90, jit=jit, seed=8)


@pytest.mark.parametrize("ues", [[50] * 20, [33] * 8])
def test_builtin_nvs_kernels_split_from_32_users_per_slice(rs, oracle, ues):
    _check_batch(rs, oracle, 7, ues, 25, 4, n_cells=2, n_ttis=90, jit=False, seed=9)
    _check_batch(rs, oracle, 7, ues, 25, 4, n_cells=2, n_ttis=90, jit=True, seed=9)


@pytest.mark.parametrize("sched,ues,R,G", [(9, [25] * 20, 25, 4), (8, [10] * 20, 64, 8), (9, [13, 13, 6, 11, 10], 64, 8), (103, [5] * 20, 25, 4)])
def test_autotune_leaves_no_trace_and_keeps_results(rs, oracle, sched, ues, R, G):
    """rs_batch_config.autotune: the trials run on the batch's own next TTIs from a snapshot that is put back; the batch then
    continues exactly like one that was never tuned, whatever variant was kept -- counters, PF averages, slice state, clock and the
    run that follows, all against the oracle."""
    n_cells, n1, n2 = 3, 300, 340
    sc = rs.SliceConfig(ues)
    U = sc.n_users
    grids = synth_cqi(77, (n_cells, (n1 + n2 + 39) // 40 + 1, U, R), HIST)
    seeds = np.arange(n_cells, dtype=np.uint32) + 99
    b = rs.BatchScheduler(sc, R, G, n_cells, sched=sched, jit=True, autotune=True)
    b.seed(seeds)
    b.upload_cqi_epochs(grids)
    assert b.autotune_report()[0] == 0
    b.run(40)                       # a short launch first: below the lean build's threshold, no tuning yet
    assert b.autotune_report()[0] == 0
    b.prepare_launch(n1)            # tunes here: >= 3 candidates timed on TTIs 40 .. 40 + 300, state put back
    n_cand, text = b.autotune_report()
    assert n_cand >= 3 and "rule table" in text and "kept" in text, text
    assert b.ttis_done == 40
    t, lu = b.clock()
    b.run(n1 - 40)
    got = b.run_logged(n2)          # logged launches use the general build
    st = b.state()
    b.close()
    for c in range(n_cells):
        cell = oracle.Cell(ues, R, G, sched)
        logs = cell.run_synth(grids[c], int(seeds[c]), n1 + n2)
        ost = cell.state()
        np.testing.assert_array_equal(got["rbg_to_user"][c], logs["rbg_to_user"][n1:], err_msg=f"cell {c}")
        np.testing.assert_array_equal(got["tbs_bits"][c], logs["tbs_bits"][n1:])
        np.testing.assert_array_equal(st["cum_bytes"][c], ost["cum_bytes"])
        np.testing.assert_array_equal(st["cum_rbs"][c], ost["cum_rbs"])
        assert st["avg_rate"][c].tobytes() == ost["avg_rate"].tobytes()
        assert st["slice_state"][c].tobytes() == ost["slice_state"].tobytes()


def test_autotune_is_a_no_op_where_it_does_not_apply(rs):
    sc = rs.SliceConfig([5] * 20)
    for sched, jit in ((7, True), (1, True), (9, False)):
        b = rs.BatchScheduler(sc, 25, 4, 2, sched=sched, jit=jit, autotune=True)
        b.seed(np.array([1, 2], np.uint32))
        b.synthesize_cqi(3, 16)
        b.prepare_launch(400)
        b.run(400)
        assert b.autotune_report() == (0, "")
        b.close()


@pytest.mark.parametrize("poll", ["1", "0"])
@pytest.mark.parametrize("jit", [False, True])
def test_drop_in_completion_word(rs, oracle, monkeypatch, poll, jit):
    """RS_DROPIN_POLL: the kernel's last store publishes the call's sequence number in the pinned output block and the host spins
    on it (default), or the stream's completion is waited for (0): the same results either way, over many back-to-back calls."""
    monkeypatch.setenv("RS_DROPIN_POLL", poll)
    ues, R, G = [25] * 20, 25, 4
    sc = rs.SliceConfig(ues)
    U = sc.n_users
    ts = rs.TtiScheduler(sc, R, G, sched=9, jit=jit)
    cell = oracle.Cell(ues, R, G, 9)
    rng = np.random.default_rng(13)
    for it in range(300):
        cqi = synth_cqi(500 + it % 7, (U, R), HIST)
        avg = rng.uniform(1e3, 5e6, U)
        r0, r1 = int(rng.integers(0, 2**31 - 1)), int(rng.integers(0, 2**31 - 1))
        res = ts.schedule_tti(cqi, avg, r0, r1)
        cell.set_cqi(cqi)
        out = cell.new_out()
        assert cell.allocate(avg, r0, r1, out) == 0
        np.testing.assert_array_equal(res.rbg_to_user, out.rbg_to_user, err_msg=f"call {it}")
        np.testing.assert_array_equal(res.user_tbs_bits, out.user_tbs_bits, err_msg=f"call {it}")
        np.testing.assert_array_equal(res.quota_rbgs, out.quota_rbgs, err_msg=f"call {it}")
    ts.close()


def _selfcheck_batch(rs, sched, ues, R, G, **kw):
    sc = rs.SliceConfig(ues)
    b = rs.BatchScheduler(sc, R, G, 3, sched=sched, jit=True, selfcheck=True, **kw)
    b.seed(np.arange(3, dtype=np.uint32) + 5)
    b.synthesize_cqi(11, 24)
    return b


@pytest.mark.parametrize("sched,ues,R,G", [(9, [25] * 20, 25, 4), (9, [5] * 20, 64, 8), (8, [10] * 20, 64, 8), (7, [25] * 20, 25, 4),
                                           (1, [30] * 20, 25, 4), (103, [5] * 20, 25, 4), (101, [5] * 20, 25, 4), (10, [5] * 20, 25, 4),
                                           (11, [5] * 20, 25, 4)])
def test_selfcheck_passes_and_leaves_no_trace(rs, oracle, sched, ues, R, G):
    """rs_batch_config.selfcheck: built-in, general and lean build agree on the batch's own next TTIs; the run afterwards is the
    oracle's, as if nothing had happened."""
    b = _selfcheck_batch(rs, sched, ues, R, G)
    grids = [b.download_cqi_epochs(c) for c in range(3)]
    b.run(30)                       # the general build is checked here (30 TTIs; too short a launch for the lean build)
    code, msg = b.jit_status()
    assert code == 1 and "selfcheck over 30 TTIs" in msg and "general build agree" in msg, (code, msg)
    b.prepare_launch(400)           # ... the lean build here
    code, msg = b.jit_status()
    assert code == 1 and "selfcheck over 256 TTIs" in msg and "general and lean builds agree" in msg, (code, msg)
    assert b.ttis_done == 30
    b.run(400)
    st = b.state()
    b.close()
    for c in range(3):
        cell = oracle.Cell(ues, R, G, sched)
        cell.run_synth(grids[c], 5 + c, 430, log=False)
        ost = cell.state()
        np.testing.assert_array_equal(st["cum_bytes"][c], ost["cum_bytes"])
        assert st["avg_rate"][c].tobytes() == ost["avg_rate"].tobytes()
        assert st["slice_state"][c].tobytes() == ost["slice_state"].tobytes()


def test_selfcheck_drops_a_wrong_run_time_build(rs, oracle, monkeypatch):
    """A deliberately wrong shape-specialised build (one byte more per grant, -DRS_FAULT_INJECT_JIT): the self-check notices, the batch
    falls back to the built-in kernels, says so (-2), and its results are the oracle's."""
    monkeypatch.setenv("RS_JIT_EXTRA", "-DRS_FAULT_INJECT_JIT")
    ues, R, G = [25] * 20, 25, 4
    b = _selfcheck_batch(rs, 9, ues, R, G)
    grids = [b.download_cqi_epochs(c) for c in range(3)]
    assert b.kernel_name == "rs_cell_kernel_jit"
    b.prepare_launch(300)
    code, msg = b.jit_status()
    assert code == -2 and "differs from the built-in" in msg, (code, msg)
    assert b.kernel_name != "rs_cell_kernel_jit"
    b.run(300)
    st = b.state()
    b.close()
    for c in range(3):
        cell = oracle.Cell(ues, R, G, 9)
        cell.run_synth(grids[c], 5 + c, 300, log=False)
        np.testing.assert_array_equal(st["cum_bytes"][c], cell.state()["cum_bytes"])
    # without the self-check the wrong build would have served the batch -- the fault injection really bites (selfcheck = -1: since
    # round 6 the check is on by default; another workgroup size = another code object: the first one is rejected for this process)
    sc = rs.SliceConfig(ues)
    b = rs.BatchScheduler(sc, R, G, 1, sched=9, jit=True, selfcheck=-1, threads_per_cell=256)
    b.seed(np.array([5], np.uint32))
    b.synthesize_cqi(11, 24)
    b.run(300)
    cell = oracle.Cell(ues, R, G, 9)
    cell.run_synth(grids[0], 5, 300, log=False)
    assert (b.state()["cum_bytes"][0] != cell.state()["cum_bytes"]).any()
    b.close()


def test_autotune_never_keeps_a_variant_whose_state_differs(rs, oracle, monkeypatch):
    """-DRS_FAULT_INJECT_P3B8 breaks exactly one autotune candidate of MaximizeCell (the 8-users-per-block build): the report says
    REJECTED and the batch's results stay the oracle's."""
    monkeypatch.setenv("RS_JIT_EXTRA", "-DRS_FAULT_INJECT_P3B8")
    ues, R, G = [25] * 20, 25, 4
    sc = rs.SliceConfig(ues)
    b = rs.BatchScheduler(sc, R, G, 2, sched=9, jit=True, autotune=True)
    b.seed(np.array([5, 6], np.uint32))
    b.synthesize_cqi(11, 24)
    grids = [b.download_cqi_epochs(c) for c in range(2)]
    b.prepare_launch(600)
    n, text = b.autotune_report()
    assert "-DRS_P3_BLOCK=8" in text and "REJECTED" in text, text
    b.run(600)
    st = b.state()
    b.close()
    for c in range(2):
        cell = oracle.Cell(ues, R, G, 9)
        cell.run_synth(grids[c], 5 + c, 600, log=False)
        np.testing.assert_array_equal(st["cum_bytes"][c], cell.state()["cum_bytes"])
        assert st["avg_rate"][c].tobytes() == cell.state()["avg_rate"].tobytes()


def _fresh(rs, sched, ues, R, G, n_cells, grids, seeds, jit, threads=0, phy=False):
    sc = rs.SliceConfig(ues)
    b = rs.BatchScheduler(sc, R, G, n_cells, sched=sched, jit=jit, threads_per_cell=threads, phy_error_draws=phy)
    b.seed(seeds)
    b.upload_cqi_epochs(grids)
    return b


@pytest.mark.parametrize("sched,ues,R,G,phy", [(9, [25] * 20, 25, 4, False), (9, [5] * 20, 64, 8, True), (8, [10] * 20, 25, 4, False),
                                               (7, [25] * 20, 25, 4, True), (1, [30] * 20, 25, 4, False), (11, [5] * 20, 25, 4, False)])
def test_checkpoint_resume_is_exact(rs, oracle, sched, ues, R, G, phy):
    """run, save, run == load into a new batch, run -- bit for bit, across kernel families (the pending-grant words are converted) and
    workgroup sizes, and the whole run is the oracle's."""
    n_cells, n1, n2 = 2, 137, 263
    U = sum(ues)
    grids = synth_cqi(31, (n_cells, (n1 + n2 + 39) // 40, U, R), HIST)
    seeds = np.arange(n_cells, dtype=np.uint32) + 77
    ref = []
    for c in range(n_cells):
        cell = oracle.Cell(ues, R, G, sched)
        cell.run_synth(grids[c], int(seeds[c]), n1 + n2, phy_error_draws=int(phy), log=False)
        ref.append(cell.state())
    for jit_a, jit_b, thr_b in ((True, True, 0), (True, False, 0), (False, True, 0), (True, True, 256), (False, False, 128)):
        a = _fresh(rs, sched, ues, R, G, n_cells, grids, seeds, jit_a, phy=phy)
        a.run(n1)
        blob = a.checkpoint()
        a.run(n2)
        st_a = a.state()
        a.close()
        b = _fresh(rs, sched, ues, R, G, n_cells, grids, np.zeros(n_cells, np.uint32), jit_b, threads=thr_b, phy=phy)  # (another seed: the ring comes from the block)
        b.restore(blob)
        assert b.ttis_done == n1
        b.run(n2)
        st_b = b.state()
        t_b = b.clock()
        b.close()
        for k in ("cum_bytes", "cum_rbs"):
            np.testing.assert_array_equal(st_a[k], st_b[k], err_msg=f"{k} {jit_a}->{jit_b}")
        assert st_a["avg_rate"].tobytes() == st_b["avg_rate"].tobytes() and st_a["slice_state"].tobytes() == st_b["slice_state"].tobytes()
        for c in range(n_cells):
            np.testing.assert_array_equal(st_b["cum_bytes"][c], ref[c]["cum_bytes"])
            np.testing.assert_array_equal(st_b["cum_rbs"][c], ref[c]["cum_rbs"])
            assert st_b["avg_rate"][c].tobytes() == ref[c]["avg_rate"].tobytes()


def test_checkpoint_refuses_what_does_not_fit(rs):
    sc = rs.SliceConfig([5] * 4)
    a = rs.BatchScheduler(sc, 12, 2, 2, sched=9)
    a.seed(np.array([1, 2], np.uint32))
    a.synthesize_cqi(1, 4)
    a.run(50)
    blob = a.checkpoint()
    for other in (rs.BatchScheduler(sc, 12, 2, 3, sched=9), rs.BatchScheduler(sc, 12, 2, 2, sched=8), rs.BatchScheduler(rs.SliceConfig([5] * 3 + [6]), 12, 2, 2, sched=9)):
        with pytest.raises(rs.RadioSaberError) as e:
            other.restore(blob)
        assert "another batch" in str(e.value)
        other.close()
    with pytest.raises(rs.RadioSaberError):
        a.restore(blob[:100])
    with pytest.raises(rs.RadioSaberError):
        a.restore(b"x" * len(blob))
    a.restore(blob)  # its own checkpoint: back to TTI 50
    assert a.ttis_done == 50
    a.close()


def test_checkpoint_resume_with_the_queue_model(rs, oracle):
    """The queue model's checkpoint carries the bearers' queues, averages and counters too."""
    from test_gpu_queues import _random_bursts
    ues, R, G, n_cells, n1, n2 = [4, 4, 4], 25, 4, 2, 90, 150
    sc = rs.SliceConfig(ues, algo_alpha=[1, 0, 0], algo_beta=[1, 0, 0])
    U = sc.n_users
    kinds = np.zeros((U, 2), np.uint8)
    kinds[:, 0] = rs.BEARER_QUEUE
    kinds[4:, 0] = rs.BEARER_BACKLOG
    kinds[:4, 1] = rs.BEARER_QUEUE
    rng = np.random.default_rng(5)
    bursts = {(c, u, k): _random_bursts(rng, n1 + n2, 5, 2500) for c in range(n_cells) for u in range(U) for k in range(2) if kinds[u, k] == rs.BEARER_QUEUE}
    grids = synth_cqi(41, (n_cells, (n1 + n2 + 39) // 40, U, R), HIST)
    seeds = np.array([3, 4], np.uint32)

    def make(jit):
        b = rs.BatchScheduler(sc, R, G, n_cells, sched=9, jit=jit)
        b.seed(seeds)
        b.upload_cqi_epochs(grids)
        b.set_bearers(kinds)
        b.set_arrivals(bursts)
        return b
    for jit in (False, True):
        a = make(jit)
        a.run(n1)
        blob = a.checkpoint()
        a.run(n2)
        sa = a.bearer_state()
        a.close()
        b = make(jit)
        b.restore(blob)
        b.run(n2)
        sb = b.bearer_state()
        b.close()
        for k in sa:
            assert sa[k].tobytes() == sb[k].tobytes(), (k, jit)


@pytest.mark.parametrize("mode", ["0", "1", "2"])
def test_issue_priority_sharing_changes_no_result(rs, oracle, monkeypatch, mode):
    """RS_PRIO_BALANCE (off / time windows / progress feedback) is scheduling only: more cells than compute units, every mode, bit-exact."""
    monkeypatch.setenv("RS_PRIO_BALANCE", mode)
    ues, R, G, n_cells, n_ttis = [3] * 6, 12, 2, 600, 70
    sc = rs.SliceConfig(ues)
    grids = synth_cqi(5, (n_cells, 2, sc.n_users, R), HIST)
    seeds = np.arange(n_cells, dtype=np.uint32) + 1
    for sched, jit in ((9, True), (9, False), (7, True), (8, True)):
        b = rs.BatchScheduler(sc, R, G, n_cells, sched=sched, jit=jit)
        b.seed(seeds)
        b.upload_cqi_epochs(grids)
        b.run(n_ttis)
        st = b.state()
        b.close()
        for c in (0, 1, 255, 256, 511, 599):
            cell = oracle.Cell(ues, R, G, sched)
            cell.run_synth(grids[c], int(seeds[c]), n_ttis, log=False)
            ost = cell.state()
            np.testing.assert_array_equal(st["cum_bytes"][c], ost["cum_bytes"])
            assert st["avg_rate"][c].tobytes() == ost["avg_rate"].tobytes()


def test_co_resident_cells_finish_together(rs, monkeypatch):
    """The headline batch (512 cells on 256 CUs): without the feedback the cells dispatched second finish ~8 % after the first ones
    (they lose every issue tie by age); with it all cells finish within a fraction of that (each cell's own run time: rs_batch_debug_clocks)."""
    spread = {}
    for mode in ("0", "2"):
        monkeypatch.setenv("RS_PRIO_BALANCE", mode)
        sc = rs.SliceConfig([25] * 20)
        b = rs.BatchScheduler(sc, 25, 4, 512, sched=9, jit=True, cqi_epoch_wrap=True)
        b.seed(np.arange(512, dtype=np.uint32) + 9)
        b.synthesize_cqi(3, 64)
        b.prepare_launch(2000)
        b.run(2000)
        b.run(2000)
        _, ms = b.debug_clocks()
        b.close()
        spread[mode] = (float(ms[:256].mean()), float(ms[256:].mean()), float(ms.max() - ms.min()))
    first0, second0, sp0 = spread["0"]
    first2, second2, sp2 = spread["2"]
    if rs.lib().rs_device_count() and second0 < 1.03 * first0:
        pytest.skip(f"this device does not show the age effect ({first0:.2f} / {second0:.2f} ms)")
    assert sp2 < 0.6 * sp0, spread
    assert abs(second2 - first2) < 0.5 * (second0 - first0), spread


@pytest.mark.gpu
@pytest.mark.parametrize("ues,R,G,threads,jit", [
    ([25] * 20, 64, 8, 0, True),        # the 64-RBG sweep shape: 16 lanes per sample, four samples per wave and pass
    ([25] * 20, 64, 8, 0, False),
    # (the key table u16[n][4][ceil(R / 4)][4] lives behind the slice's metrics: the array is sized for U users on doubles and for the
    #  batch's longest slice of <= 64 users with its keys, rs_nvs_val_bytes)
    ([8, 16, 32, 5] + [60] * 9, 64, 8, 0, True),   # slice sizes whose u16[R][n][4] rows of round 4 sat 16- to 32-way on one LDS bank
    ([13, 40, 1, 250, 240], 50, 4, 0, True),       # 50 RBGs: 13 units, the last one half padding; a 1-user slice; 250: on doubles, few samples per step
    ([6, 0, 11, 3, 100], 34, 3, 64, False),        # one wave per cell: it draws, scans and adds up in turn
    ([29, 28, 31, 200, 210], 64, 8, 256, True),    # 210 > 64: the array is 32 U = 15 936 B; 29 users x 544 B just fit (15 776), 31 scan on doubles
    ([40, 10, 210, 200], 40, 3, 128, True),        # 10 units per row; the 40-user slice's keys (14 080 B) under 32 U = 14 720
    ([10, 20, 30], 64, 8, 0, True),                # few slices: the key table is sized for the longest slice, not for 32 U bytes
    ([10, 20, 30], 64, 8, 0, False),
    ([64, 3], 25, 4, 0, True),                     # 64 users exactly; 7 units of four RBGs, the last with three of padding
])
def test_sampler_wide_grids(rs, oracle, ues, R, G, threads, jit):
    """ORACLE UNPINNED (sched 11).  Round 5: the sampler's packed scan (a lane owns four RBGs of one sample: one draw byte, one 8-byte
    unit of keys, two packed 16-bit maxima per user), 16-bit winner indices in rows rotated by the sample index instead of metric
    doubles, draws and winner rows in one pool split per served slice, the key table built beside the first step's draws
    (rs_phase_nvs_sampler.inc); device == oracle bit for bit on every form, error-model draws included."""
    _check_batch(rs, oracle, 11, ues, R, G, n_cells=2, n_ttis=50, threads=threads, jit=jit, seed=31)
    _check_batch(rs, oracle, 11, ues, R, G, n_cells=1, n_ttis=42, threads=threads, jit=jit, seed=32, phy=1)


@pytest.mark.gpu
def test_sampler_wide_grid_drop_in(rs, oracle):
    """ORACLE UNPINNED (sched 11): the 64-RBG drop-in call (the caller's rand() values), built-in and shape-specialised."""
    ues, R, G = [9] * 4, 64, 8
    sc = rs.SliceConfig(ues)
    for jit in (False, True):
        ts = rs.TtiScheduler(sc, R, G, sched=11, jit=jit)
        cell = oracle.Cell(ues, R, G, 11)
        rng = np.random.default_rng(77)
        for it in range(6):
            cqi = synth_cqi(900 + it, (sc.n_users, R), HIST)
            avg = rng.uniform(1e3, 5e6, sc.n_users)
            sl = it % 4
            ids = np.arange(sl * 9, sl * 9 + 9)
            draws = rng.integers(0, 2**31 - 1, 300 * len(ids)).astype(np.int32)
            cell.set_cqi(cqi)
            out = cell.new_out()
            assert cell.allocate_nongreedy(avg, sl, draws, out) == 0
            res = ts.schedule_tti(cqi[ids], avg[ids], user_id=ids, rand_draws=draws)
            np.testing.assert_array_equal(res.rbg_to_user, out.rbg_to_user)
            np.testing.assert_array_equal(res.user_tbs_bits, out.user_tbs_bits[ids])
        ts.close()
