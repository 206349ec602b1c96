"""GPU parity of the batched queue model (customised slices, finite MAC queues, two bearers per user) against
the CPU oracle's literal per-packet restatement (oracle/rs_oracle.cpp, rso_cell_step_queues).

PARITY UNPINNED: neither side has a reference output to be checked against in this image (the simulator cannot be built); these
tests prove device == oracle.  The device keeps a queue as a window over the uploaded arrival bursts with closed-form dequeue
arithmetic, the oracle a std::deque of packets: two independent formulations of flows/MacQueue.cpp + um-rlc-entity.cpp."""
import json

import numpy as np
import pytest

from conftest import GOLDEN, synth_cqi

pytestmark = pytest.mark.gpu

HIST = (152600, 56656, 270880, 2088792, 3509504, 1595568, 4145392, 5295816, 1903424,
        6890232, 4770864, 2842552, 3579624, 96000, 1227696)


def _random_bursts(rng, n_ttis, mean_gap_ms, mean_bytes, start_after=0):
    """Arrival bursts on the millisecond grid of the applications (0.1 + k ms by repeated addition like Simulator::DoSchedule)."""
    t, out_t, out_b = 0.1, [], []
    k = start_after
    first = True
    while True:
        gap = 0 if (first and start_after == 0) else int(rng.geometric(1.0 / mean_gap_ms))
        first = False
        k += gap
        if k >= n_ttis + 5:
            break
        t = t + gap / 1000.0 if gap else t
        out_t.append(t)
        out_b.append(int(max(40, rng.exponential(mean_bytes))))
    b = np.array(out_b, np.int64)
    return np.array(out_t), (b // 1490).astype(np.int32), (b % 1490).astype(np.int32)


def _run_case(rs, oracle, sched, ues, slice_kinds, alpha, beta, R, G, n_cells, launches, jit, seed, threads=0,
              mean_gap_ms=6, mean_bytes=2500, psi=None, logged=True):
    S = len(ues)
    sc = rs.SliceConfig(ues, algo_alpha=alpha, algo_beta=beta, algo_psi=psi or [])
    U = sc.n_users
    u2s = sc.user_to_slice
    kinds = np.zeros((U, 2), np.uint8)
    code = {"B": rs.BEARER_BACKLOG, "Q": rs.BEARER_QUEUE, "-": rs.BEARER_NONE}
    for u in range(U):
        kinds[u, 0], kinds[u, 1] = code[slice_kinds[u2s[u]][0]], code[slice_kinds[u2s[u]][1]]
    n_ttis = sum(launches)
    rng = np.random.default_rng(seed)
    bursts = {}
    for c in range(n_cells):
        for u in range(U):
            for k in range(2):
                if kinds[u, k] == rs.BEARER_QUEUE:
                    # a few bearers start late or stay nearly idle: users and whole slices drop out of UsersToSchedule
                    late = int(rng.integers(0, n_ttis // 2)) if rng.random() < 0.25 else 0
                    gap = mean_gap_ms * (8 if rng.random() < 0.2 else 1)
                    bursts[(c, u, k)] = _random_bursts(rng, n_ttis, gap, mean_bytes, late)
    grids = synth_cqi(seed + 1, (n_cells, (n_ttis + 39) // 40, U, R), HIST)
    seeds = np.arange(n_cells, dtype=np.uint32) * 31 + 1000 + seed
    b = rs.BatchScheduler(sc, R, G, n_cells, sched=sched, jit=jit, threads_per_cell=threads)
    b.set_bearers(kinds)
    b.set_arrivals(bursts)
    if jit:
        assert b.jit_status()[0] == 1, b.jit_status()
    b.seed(seeds)
    b.upload_cqi_epochs(grids)
    maps = tbs = None
    if logged:
        got = [b.run_logged(n) for n in launches]
        maps = np.concatenate([g["rbg_to_user"] for g in got], axis=1)
        tbs = np.concatenate([g["tbs_bits"] for g in got], axis=1)
    else:  # unlogged launches (the lean build of the kernel when RS_JIT_LEAN_MIN_TTIS allows): final state only
        for n in launches:
            b.run(n)
    st, bst = b.state(), b.bearer_state()
    b.close()
    idle_ttis = 0
    for c in range(n_cells):
        cell = oracle.Cell(ues, R, G, sched, alpha=alpha, beta=beta, psi=psi)
        cell.enable_queues(kinds)
        for (cc, u, k), (t, nf, la) in bursts.items():
            if cc == c:
                cell.set_arrivals(u, k, t, nf, la)
        logs = cell.run_synth_queues(grids[c], int(seeds[c]), n_ttis)
        ob = cell.bearer_state()
        if logged:
            np.testing.assert_array_equal(maps[c], logs["rbg_to_user"], err_msg=f"cell {c} RBG map")
            np.testing.assert_array_equal(tbs[c], logs["tbs_bits"], err_msg=f"cell {c} TBS")
        for key in ("cum_bytes", "cum_rbs", "queue_bytes", "queue_packets"):
            np.testing.assert_array_equal(bst[key][c], ob[key], err_msg=f"cell {c} {key}")
        live = kinds != rs.BEARER_NONE
        assert bst["avg_rate"][c][live].tobytes() == ob["avg_rate"][live].tobytes(), f"cell {c} bearer PF averages differ"
        assert st["slice_state"][c].tobytes() == cell.state()["slice_state"].tobytes(), f"cell {c} slice offsets differ"
        assert (st["cum_bytes"][c] == ob["cum_bytes"].sum(1)).all()
        idle_ttis += int((logs["rbg_to_user"] < 0).all(1).sum())
    return idle_ttis, maps


@pytest.mark.parametrize("sched", [9, 8, 7, 1, 103])
def test_lean_build_of_the_queue_model_kernels(rs, oracle, sched, monkeypatch):
    """Round 4: unlogged launches of a queue-model batch run the lean build of ITS shape-specialised kernel too (the launch's unused
    options as constants; RS_JIT_LEAN_MIN_TTIS = 1 here): bearers' counters, queues, averages and slice offsets against the oracle."""
    monkeypatch.setenv("RS_JIT_LEAN_MIN_TTIS", "1")
    kinds = ["B-", "Q-", "QQ", "BQ", "Q-", "QQ"]
    custom = sched in (9, 8, 7, 103)
    alpha = [0, 1, 1, 1, 0, 1] if custom else [0] * 6
    beta = [0, 0, 1, 1, 0, 0] if custom else [0] * 6
    _run_case(rs, oracle, sched, [9, 12, 7, 10, 3, 11], kinds, alpha, beta, 25, 4, n_cells=3, launches=[1, 39, 60, 47], jit=True, seed=77 + sched,
              logged=False)
    _run_case(rs, oracle, sched, [20] * 3, ["Q-", "QQ", "B-"], [0, 0, 0], [0, 0, 0], 64, 8, n_cells=2, launches=[120], jit=True, seed=80 + sched,
              mean_gap_ms=5, mean_bytes=1500, logged=False)


@pytest.mark.parametrize("sched", [9, 8])
@pytest.mark.parametrize("jit", [False, True])
def test_customised_slices_with_queues(rs, oracle, sched, jit):
    """Backlogged, single-queue (alpha = 1), two-bearer (alpha = 1) and HoL-weighted (alpha = beta = 1) slices side by side."""
    ues = [4, 5, 3, 6, 4]
    _run_case(rs, oracle, sched, ues, ["B-", "Q-", "QQ", "Q-", "QQ"], [0, 1, 1, 1, 1], [0, 0, 0, 1, 1], 25, 4,
              n_cells=3, launches=[1, 39, 60, 100], jit=jit, seed=3)


def test_finite_queues_without_customisation(rs, oracle):
    """alpha = 0 everywhere, every flow rate-limited (the exp-*/config.json files with internet_flow = 1): users join and leave
    UsersToSchedule, slices lose their targets while empty, and there are TTIs in which nothing is scheduled (no rand() drawn)."""
    idle, _ = _run_case(rs, oracle, 9, [3, 4, 3], ["Q-", "Q-", "Q-"], [0, 0, 0], [0, 0, 0], 25, 4, n_cells=4,
                        launches=[150, 150], jit=False, seed=11, mean_gap_ms=25, mean_bytes=1200)
    assert idle > 0, "no TTI without any queued data: the skipped-allocation path was not exercised"
    _run_case(rs, oracle, 8, [3, 4, 3], ["Q-", "QQ", "B-"], [0, 0, 0], [0, 0, 0], 64, 8, n_cells=2, launches=[120], jit=True,
              seed=12, mean_gap_ms=10, mean_bytes=30000)


@pytest.mark.parametrize("sched", [101, 103])
def test_queues_with_the_other_inter_slice_policies(rs, oracle, sched):
    _run_case(rs, oracle, sched, [4, 4, 4], ["B-", "Q-", "QQ"], [0, 1, 1], [0, 0, 1], 25, 4, n_cells=2, launches=[90],
              jit=True, seed=21)


def test_queues_headline_shape_and_workgroup_sizes(rs, oracle):
    """20 slices x 25 UEs with mixed traffic; psi = 0 in some slices; 256- and 512-thread cells (two register forms of the sort)."""
    S = 20
    kinds = ["B-"] * 5 + ["Q-"] * 5 + ["QQ"] * 5 + ["Q-"] * 5
    alpha = [0] * 5 + [1] * 15
    beta = [0] * 15 + [1] * 5
    psi = [1, 0] * 10
    for threads, jit in ((512, True), (256, False)):
        _run_case(rs, oracle, 9, [25] * S, kinds, alpha, beta, 25, 4, n_cells=1, launches=[45, 45], jit=jit, seed=40 + threads,
                  threads=threads, mean_gap_ms=12, mean_bytes=6000, psi=psi)


def test_fragmentation_and_overhead_accounting(rs, oracle):
    """Large flows, small grants: most dequeues end in a fragment, and grants of at most 8 bytes of room send nothing."""
    _run_case(rs, oracle, 9, [12, 12], ["Q-", "QQ"], [1, 1], [0, 1], 12, 2, n_cells=2, launches=[200], jit=False, seed=5,
              mean_gap_ms=40, mean_bytes=90000)


@pytest.mark.parametrize("sched", [9, 7, 1])
def test_customize_20slices_experiment_runs_as_a_batch(rs, oracle, sched):
    """exp-customization/exp-customize-20slices/config.json with the three scheduler numbers its run script passes (1, 7, 9;
    run_customize.sh): backlogged,
    one / two InternetFlow bearers and video slices as ONE batch.  Traffic: rs_internet_flow_arrivals with the config's rates
    and the reference's 1 280 kbit/s video trace (tests/golden/video_foreman_1280k.json); 64 RBGs of 8 PRBs like the run script."""
    cfg = json.loads((GOLDEN / "experiment_configs.json").read_text())["exp-customization/exp-customize-20slices/config.json"]
    sc = rs.SliceConfig(cfg["ues_per_slice"], cfg["weight"], cfg["algo_alpha"], cfg["algo_beta"], cfg["algo_epsilon"],
                        cfg["algo_psi"], cfg["traffic"])
    assert any(sc.algo_alpha) and any(sc.algo_beta)
    kinds = sc.bearer_kinds()
    U, u2s = sc.n_users, sc.user_to_slice
    video = json.loads((GOLDEN / "video_foreman_1280k.json").read_text())
    n_cells, n_ttis, R, G = 2, 240, 64, 8
    stop = 0.1 + n_ttis / 1000.0 + 0.01
    bursts = {}
    for c in range(n_cells):
        for u in range(U):
            tr = cfg["traffic"][u2s[u]]
            for j in range(int(tr["internet_flow"])):
                rate = tr["if_bitrate"][j] / cfg["ues_per_slice"][u2s[u]]  # single-cell-with-interference.h:415-416
                bursts[(c, u, j)] = rs.internet_flow_arrivals(rate, 0.1, stop, 1000 * c + 2 * u + j)
            if int(tr["video_app"]):
                t, ts = 0.1, []
                for k in range(len(video["bytes"])):  # TraceBased::Send: next frame TimeToSend * 0.001 after this one
                    if k:
                        t = (video["time_ms"][k] - video["time_ms"][k - 1]) * 0.001 + t
                    if t >= stop:
                        break
                    ts.append(t)
                bursts[(c, u, 0)] = rs.frames_to_bursts(ts, video["bytes"][:len(ts)])
    grids = synth_cqi(77, (n_cells, (n_ttis + 39) // 40, U, R), HIST)
    seeds = np.array([5, 6], np.uint32)
    b = rs.BatchScheduler(sc, R, G, n_cells, sched=sched, jit=True)
    b.set_bearers(kinds)
    b.set_arrivals(bursts)
    b.seed(seeds)
    b.upload_cqi_epochs(grids)
    got = b.run_logged(n_ttis)
    bst = b.bearer_state()
    b.close()
    for c in range(n_cells):
        cell = oracle.Cell(cfg["ues_per_slice"], R, G, sched, weights=cfg["weight"], alpha=cfg["algo_alpha"], beta=cfg["algo_beta"],
                           epsilon=cfg["algo_epsilon"], psi=cfg["algo_psi"])
        cell.enable_queues(kinds)
        for (cc, u, k), (t, nf, la) in bursts.items():
            if cc == c:
                cell.set_arrivals(u, k, t, nf, la)
        logs = cell.run_synth_queues(grids[c], int(seeds[c]), n_ttis)
        ob = cell.bearer_state()
        np.testing.assert_array_equal(got["rbg_to_user"][c], logs["rbg_to_user"])
        np.testing.assert_array_equal(got["tbs_bits"][c], logs["tbs_bits"])
        for key in ("cum_bytes", "cum_rbs", "queue_bytes", "queue_packets"):
            np.testing.assert_array_equal(bst[key][c], ob[key], err_msg=key)
    # the video slices (alpha = beta = 1) and the prioritized flows got service
    vid = np.flatnonzero(np.array([int(cfg["traffic"][s]["video_app"]) for s in u2s]) > 0)
    assert bst["cum_bytes"][:, vid, 0].sum() > 0
    two = np.flatnonzero(kinds[:, 1] == rs.BEARER_QUEUE)
    assert bst["cum_bytes"][:, two, 1].sum() > 0


@pytest.mark.parametrize("jit", [False, True])
def test_nvs_with_queues_and_the_required_rbs_gate(rs, oracle, jit):
    """Sched 7 with finite queues: SelectSliceToServe over the slices that have queued data, the customised metric (always
    HoL-weighted here), and the m_requiredRBs gate -- with a few thousand bytes queued a user needs only a few RBGs and drops
    out of the race for the rest of the TTI (downlink-nvs-scheduler.cpp:299-300)."""
    _, maps = _run_case(rs, oracle, 7, [4, 5, 3, 6], ["B-", "Q-", "QQ", "Q-"], [0, 1, 1, 0], [0, 0, 1, 0], 25, 4, n_cells=3,
                        launches=[1, 59, 90], jit=jit, seed=31, mean_gap_ms=20, mean_bytes=400)
    # the gate binds: TTIs in which the served slice's users are satisfied before the RBGs run out leave RBGs unallocated
    partial = ((maps >= 0).any(2) & (maps < 0).any(2)).sum()
    assert partial > 0, "no TTI with unallocated RBGs: the m_requiredRBs gate never bound"
    _run_case(rs, oracle, 7, [25] * 4, ["Q-", "QQ", "B-", "Q-"], [1, 1, 0, 0], [1, 0, 0, 0], 64, 8, n_cells=1, launches=[100],
              jit=jit, seed=32, mean_gap_ms=8, mean_bytes=9000)


@pytest.mark.parametrize("jit", [False, True])
def test_per_flow_pf_with_queues_and_the_satisfied_flow_break(rs, oracle, jit):
    """Sched 1 with finite queues: every bearer with packets is a flow with its own PF average; a flow leaves the TTI's
    competition once the transport block of its PRBs so far carries its queue (downlink-packet-scheduler.cpp:253-265); the
    RBG map holds flow ids 2 * user + bearer."""
    _, maps = _run_case(rs, oracle, 1, [4, 5, 3, 6], ["B-", "Q-", "QQ", "Q-"], [0, 0, 0, 0], [0, 0, 0, 0], 25, 4, n_cells=3,
                        launches=[1, 59, 90], jit=jit, seed=41, mean_gap_ms=5, mean_bytes=1500)
    assert (maps % 2 == 1).any(), "no RBG went to a second bearer"
    _run_case(rs, oracle, 1, [20] * 3, ["Q-", "QQ", "Q-"], [0, 0, 0], [0, 0, 0], 64, 8, n_cells=1, launches=[120], jit=jit, seed=42,
              mean_gap_ms=10, mean_bytes=4000)


def test_queue_mode_is_required_for_customised_batches_and_validated(rs):
    sc = rs.SliceConfig([3, 3], algo_alpha=[0, 1])
    b = rs.BatchScheduler(sc, 12, 2, 1, sched=9)
    b.seed(np.array([1], np.uint32))
    b.synthesize_cqi(1, 2)
    with pytest.raises(rs.RadioSaberError) as e:
        b.run(5)
    assert "rs_batch_set_bearers" in str(e.value)
    kinds = np.array([[1, 0]] * 3 + [[2, 2]] * 3, np.uint8)
    b.set_bearers(kinds)
    with pytest.raises(rs.RadioSaberError) as e:
        b.run(5)
    assert "rs_batch_set_arrivals" in str(e.value)
    b.set_arrivals({})
    b.run(5)  # the queue bearers simply never receive anything: only the backlogged slice is served
    bs = b.bearer_state()
    assert bs["cum_bytes"][0, :3, 0].sum() > 0 and bs["cum_bytes"][0, 3:].sum() == 0
    b.close()
    b10 = rs.BatchScheduler(rs.SliceConfig([3, 3]), 12, 2, 1, sched=10)
    with pytest.raises(rs.RadioSaberError):
        b10.set_bearers(kinds)
    b10.close()
