"""GPU parity: the HIP path (through the C ABI) against the CPU oracle, bit-exact.

Integer outputs (RBG->UE map, quotas, TBS bits, cumulative bytes/RBs) must be identical; the PF
averages are FP64 and must be bitwise identical too (same IEEE operations in the same order).
"""
import json

import numpy as np
import pytest

from conftest import GOLDEN, synth_cqi

pytestmark = pytest.mark.gpu

HIST = (152600, 56656, 270880, 2088792, 3509504, 1595568, 4145392, 5295816, 1903424,
        6890232, 4770864, 2842552, 3579624, 96000, 1227696)


def _oracle_cells(oracle, ues, R, G, sched, weights, grids, seeds, n_ttis, eps=None, psi=None, phy=0):
    out = []
    for c in range(grids.shape[0]):
        cell = oracle.Cell(ues, R, G, sched, weights=weights, epsilon=eps, psi=psi)
        logs = cell.run_synth(grids[c], int(seeds[c]), n_ttis, phy_error_draws=phy)
        out.append((logs, cell.state()))
    return out


def _check_batch(rs, oracle, sched, ues, R, G, n_cells, n_ttis, threads=0, eps=None, psi=None, phy=0, seed=1,
                 weights=None, jit=False):
    S = len(ues)
    weights = weights or [1.0 / S] * S
    sc = rs.SliceConfig(ues, weight=weights, algo_epsilon=eps or [], algo_psi=psi or [])
    U = sc.n_users
    n_epochs = (n_ttis + 39) // 40
    grids = synth_cqi(seed, (n_cells, n_epochs, U, R), HIST)
    seeds = np.arange(n_cells, dtype=np.uint32) * 7919 + 805290992
    b = rs.BatchScheduler(sc, R, G, n_cells, sched=sched, phy_error_draws=bool(phy), threads_per_cell=threads, jit=jit)
    if jit:
        assert b.kernel_name == "rs_cell_kernel_jit", rs.lib().rs_last_error().decode()
    b.seed(seeds)
    b.upload_cqi_epochs(grids)
    got = b.run_logged(n_ttis)
    st = b.state()
    ref = _oracle_cells(oracle, ues, R, G, sched, weights, grids, seeds, n_ttis, eps, psi, phy)
    for c in range(n_cells):
        logs, ost = ref[c]
        np.testing.assert_array_equal(got["rbg_to_user"][c], logs["rbg_to_user"], err_msg=f"cell {c} RBG map")
        np.testing.assert_array_equal(got["tbs_bits"][c], logs["tbs_bits"], err_msg=f"cell {c} TBS")
        np.testing.assert_array_equal(st["cum_bytes"][c], ost["cum_bytes"])
        np.testing.assert_array_equal(st["cum_rbs"][c], ost["cum_rbs"])
        assert st["avg_rate"][c].tobytes() == ost["avg_rate"].tobytes(), f"cell {c} PF averages differ"
        assert st["slice_state"][c].tobytes() == ost["slice_state"].tobytes(), f"cell {c} slice state differs"
    b.close()


@pytest.mark.parametrize("sched", [9, 8, 7, 1])
def test_small_grid_all_schedulers(rs, oracle, sched):
    _check_batch(rs, oracle, sched, [5] * 20, 25, 4, n_cells=3, n_ttis=90)


@pytest.mark.parametrize("sched", [9, 8, 7, 1])
def test_shipped_grid_64_rbgs(rs, oracle, sched):
    _check_batch(rs, oracle, sched, [5] * 20, 64, 8, n_cells=2, n_ttis=85)


def test_headline_shape_500_ues_25_rbgs(rs, oracle):
    _check_batch(rs, oracle, 9, [25] * 20, 25, 4, n_cells=4, n_ttis=120)


def test_500_ues_64_rbgs(rs, oracle):
    _check_batch(rs, oracle, 9, [25] * 20, 64, 8, n_cells=2, n_ttis=60)


def test_ragged_slices_and_empty_slice(rs, oracle):
    # slice 2 has no UE at all, sizes differ
    _check_batch(rs, oracle, 9, [3, 7, 0, 1, 12], 25, 4, n_cells=2, n_ttis=50)
    _check_batch(rs, oracle, 8, [3, 7, 0, 1, 12], 25, 4, n_cells=2, n_ttis=50)


def test_skewed_weights_negative_quotas(rs, oracle):
    # a slice whose weight is far below its share keeps a negative slice_rbs_offset_ / quota
    # (the reference's "(2, -20, -2)" lines): C truncation toward zero must be reproduced
    w = [0.62, 0.3, 0.05, 0.02, 0.01]
    _check_batch(rs, oracle, 9, [10, 10, 10, 10, 10], 25, 4, n_cells=3, n_ttis=120, weights=w)
    _check_batch(rs, oracle, 8, [10, 10, 10, 10, 10], 64, 8, n_cells=2, n_ttis=80, weights=w)


def test_many_slices_and_big_slices(rs, oracle):
    _check_batch(rs, oracle, 9, [2] * 64, 25, 4, n_cells=2, n_ttis=50)      # S = 64 (lane limit)
    _check_batch(rs, oracle, 9, [70, 45, 5], 25, 4, n_cells=2, n_ttis=50)   # slices longer than one 32-user block
    _check_batch(rs, oracle, 1, [70, 45, 5], 25, 4, n_cells=2, n_ttis=50)
    _check_batch(rs, oracle, 7, [70, 45, 5], 25, 4, n_cells=2, n_ttis=50)


def test_long_run_few_slices_over_110_prbs(rs, oracle):
    """3 slices share 64 RBGs: UEs regularly hold > 110 PRBs, incl. multiples of 5 (the reference's
    out-of-bounds TBS row, pinned to the as-shipped build), over 1 500 TTIs; final state bitwise."""
    ues, R, G, n_cells, n_ttis = [4, 4, 4], 64, 8, 6, 1500
    for sched in (9, 7):
        sc = rs.SliceConfig(ues)
        b = rs.BatchScheduler(sc, R, G, n_cells, sched=sched)
        seeds = np.arange(n_cells, dtype=np.uint32) + 5
        b.seed(seeds)
        b.synthesize_cqi(77, (n_ttis + 39) // 40)
        b.run(n_ttis)
        st = b.state()
        big = 0
        for c in range(n_cells):
            cell = oracle.Cell(ues, R, G, sched)
            logs = cell.run_synth(b.download_cqi_epochs(c), int(seeds[c]), n_ttis)
            ost = cell.state()
            np.testing.assert_array_equal(st["cum_bytes"][c], ost["cum_bytes"])
            np.testing.assert_array_equal(st["cum_rbs"][c], ost["cum_rbs"])
            assert st["avg_rate"][c].tobytes() == ost["avg_rate"].tobytes()
            nprb = np.stack([np.bincount(m[m >= 0], minlength=12) for m in logs["rbg_to_user"]]) * G
            big += int(((nprb > 110) & (nprb % 5 == 0)).sum())
        assert big > 0, "the >110-PRB multiple-of-5 case was not exercised"
        b.close()


def test_single_user_single_slice(rs, oracle):
    _check_batch(rs, oracle, 9, [1], 12, 2, n_cells=1, n_ttis=45)


def test_psi_zero_max_rate(rs, oracle):
    S = 4
    _check_batch(rs, oracle, 9, [6] * S, 25, 4, n_cells=2, n_ttis=50, eps=[1] * S, psi=[0, 1, 0, 1])


def test_phy_error_draws_stream(rs, oracle):
    _check_batch(rs, oracle, 9, [5] * 20, 64, 8, n_cells=2, n_ttis=60, phy=1)


@pytest.mark.parametrize("threads", [64, 128, 320, 512])
def test_workgroup_sizes(rs, oracle, threads):
    _check_batch(rs, oracle, 9, [5] * 20, 25, 4, n_cells=2, n_ttis=45, threads=threads)


def test_trace_replay_appendix_a(rs, oracle, traces):
    """The reference's own run: seed 0, 20x5 UEs, 64 RBGs, mapping0, sched 9."""
    ka = json.loads((GOLDEN / "appendix_a.json").read_text())
    cfg = ka["config"]
    sc = rs.SliceConfig(cfg["ues_per_slice"], weight=[cfg["weight"]] * 20)
    U = sc.n_users
    b = rs.BatchScheduler(sc, 64, 8, 1, sched=9, phy_error_draws=True)
    b.seed(np.array([cfg["seed"]], np.uint32), np.array([cfg["rand_skip"]], np.int64))
    user_trace = traces["mapping"][0][np.arange(U) % 474][None, :]
    b.set_trace(traces["cqi"], user_trace)
    got = b.run_logged(200)
    first = ka["first_tti"]
    for s, (t, q) in first["quota"].items():
        assert got["quota"][0, 0, int(s)] == q
    for u, info in first["users"].items():
        rb = np.nonzero(got["rbg_to_user"][0, 0] == int(u))[0].tolist()
        assert rb == [x[0] for x in info["rbgs"]]
    assert (got["tbs_bits"][0, 0] > 0).sum() == first["n_users_served"]
    st = b.state()
    for u, (cb, cr) in ka["after_200_ttis"]["cumu"].items():
        assert st["cum_bytes"][0, int(u)] == cb and st["cum_rbs"][0, int(u)] == cr
    # and the whole run against the oracle
    cell = oracle.Cell(cfg["ues_per_slice"], 64, 8, 9, weights=[cfg["weight"]] * 20)
    logs = cell.run_trace(traces["cqi"], traces["mapping"][0], cfg["seed"], cfg["rand_skip"], 200)
    np.testing.assert_array_equal(got["rbg_to_user"][0], logs["rbg_to_user"])
    np.testing.assert_array_equal(got["tbs_bits"][0], logs["tbs_bits"])
    np.testing.assert_array_equal(got["quota"][0], logs["quota"])
    np.testing.assert_array_equal(got["target"][0], logs["target"])
    np.testing.assert_array_equal(got["final_cqi"][0], logs["final_cqi"])
    # the reference's stderr lines rebuilt from the device log
    from radiosaber_amd import logfmt
    err = logfmt.stderr_lines(got["tbs_bits"][0], got["rbg_to_user"][0], sc.user_to_slice, 8)
    assert "299 app: 5 cumu_bytes: 110838 cumu_rbs: 1336 hol_delay: 0 user: 5 slice: 1" in err
    b.close()


def test_trace_replay_500_ues_all_schedulers(rs, oracle, traces):
    """BASELINE configs[1]: 20 x 25 UEs on the real CQI traces (64 RBGs), GPU vs CPU oracle bit for bit.
    (The reference's own rand() position for this config is not recorded anywhere; a fixed skip is used.)"""
    ues = [25] * 20
    sc = rs.SliceConfig(ues, weight=[0.05] * 20)
    U = sc.n_users
    for sched, mapping in ((9, 1), (8, 2), (7, 3), (1, 0)):
        b = rs.BatchScheduler(sc, 64, 8, 1, sched=sched, phy_error_draws=True)
        b.seed(np.array([749913912], np.uint32), np.array([5000], np.int64))
        b.set_trace(traces["cqi"], traces["mapping"][mapping][np.arange(U) % 474][None, :])
        got = b.run_logged(130)
        st = b.state()
        cell = oracle.Cell(ues, 64, 8, sched, weights=[0.05] * 20)
        logs = cell.run_trace(traces["cqi"], traces["mapping"][mapping], 749913912, 5000, 130)
        np.testing.assert_array_equal(got["rbg_to_user"][0], logs["rbg_to_user"], err_msg=f"sched {sched}")
        np.testing.assert_array_equal(got["tbs_bits"][0], logs["tbs_bits"], err_msg=f"sched {sched}")
        np.testing.assert_array_equal(st["cum_bytes"][0], cell.state()["cum_bytes"])
        assert st["avg_rate"][0].tobytes() == cell.state()["avg_rate"].tobytes()
        b.close()


def test_trace_replay_500_ues_25_rbgs(rs, oracle, traces):
    """BASELINE configs[1] as written: 20 x 25 UEs x 25 RBGs on the real CQI traces.  A 100-PRB carrier reads the first 100
    values of every trace line in RBGs of 4 (enb-mac-entity.cc:178-183, get_rbg_size); the traces are constant per 8 PRBs, so
    RBG r of 4 carries the fixture's 8-PRB value r // 2.  Every scheduler the run scripts use, device vs oracle."""
    ues = [25] * 20
    sc = rs.SliceConfig(ues, weight=[0.05] * 20)
    U = sc.n_users
    cqi25 = np.ascontiguousarray(traces["cqi"][:, :, np.arange(25) // 2])
    for sched, mapping in ((9, 0), (8, 1), (7, 2), (1, 3), (10, 0), (11, 1)):
        b = rs.BatchScheduler(sc, 25, 4, 1, sched=sched, phy_error_draws=True)
        b.seed(np.array([805290992], np.uint32), np.array([7000], np.int64))
        b.set_trace(cqi25, traces["mapping"][mapping][np.arange(U) % 474][None, :])
        got = b.run_logged(130)
        st = b.state()
        cell = oracle.Cell(ues, 25, 4, sched, weights=[0.05] * 20)
        logs = cell.run_trace(cqi25, traces["mapping"][mapping], 805290992, 7000, 130)
        np.testing.assert_array_equal(got["rbg_to_user"][0], logs["rbg_to_user"], err_msg=f"sched {sched}")
        np.testing.assert_array_equal(got["tbs_bits"][0], logs["tbs_bits"], err_msg=f"sched {sched}")
        np.testing.assert_array_equal(st["cum_bytes"][0], cell.state()["cum_bytes"])
        assert st["avg_rate"][0].tobytes() == cell.state()["avg_rate"].tobytes()
        b.close()


def test_1000_ues_config3(rs, oracle):
    """BASELINE configs[2] shape: 20 slices x 50 UEs, schedulers 1/7/8/9 on synthetic sub-band CQI."""
    for sched in (9, 8, 7, 1):
        _check_batch(rs, oracle, sched, [50] * 20, 25, 4, n_cells=1, n_ttis=45)


def test_device_synth_grids_in_range_and_runs(rs, oracle):
    ues, R, G = [25] * 20, 25, 4
    sc = rs.SliceConfig(ues, weight=[0.05] * 20)
    b = rs.BatchScheduler(sc, R, G, 8, sched=9)
    seeds = np.arange(8, dtype=np.uint32) + 1
    b.seed(seeds)
    b.synthesize_cqi(12345, 3)
    g = b.download_cqi_epochs(5)
    assert g.min() >= 1 and g.max() <= 15
    hist = np.bincount(g.ravel(), minlength=16)[1:] / g.size
    p = np.asarray(HIST) / np.sum(HIST)
    assert np.abs(hist - p).max() < 0.02
    got = b.run_logged(100)
    cell = oracle.Cell(ues, R, G, 9, weights=[0.05] * 20)
    logs = cell.run_synth(g, int(seeds[5]), 100)
    np.testing.assert_array_equal(got["rbg_to_user"][5], logs["rbg_to_user"])
    np.testing.assert_array_equal(got["tbs_bits"][5], logs["tbs_bits"])
    # per-slice byte reduction (the vector the multi-GPU run all-reduces)
    st = b.state()
    per_slice = np.add.reduceat(st["cum_bytes"].sum(0), np.arange(0, 500, 25))
    np.testing.assert_array_equal(b.slice_bytes().astype(np.int64), per_slice)
    b.close()


@pytest.mark.parametrize("sched", [9, 8, 1, 7, 103, 10, 101])
def test_drop_in_single_tti(rs, oracle, sched):
    """rs_schedule_tti == RBsAllocation() of the oracle, carrying slice_rbs_offset_ across calls."""
    ues, R, G = [5] * 20, 64, 8
    sc = rs.SliceConfig(ues, weight=[0.05] * 20)
    U = sc.n_users
    ts = rs.TtiScheduler(sc, R, G, sched=sched)
    cell = oracle.Cell(ues, R, G, sched, weights=[0.05] * 20)
    rng = np.random.default_rng(3)
    for it in range(12):
        cqi = synth_cqi(100 + it, (U, R), HIST)
        avg = rng.uniform(1e3, 5e6, U)
        if it % 3 == 0:
            avg[:] = 98000.0  # exact ties everywhere
        r0, r1 = int(rng.integers(0, 2**31 - 1)), int(rng.integers(0, 2**31 - 1))
        if sched == 7:
            sl = it % 20
            ids = np.arange(sl * 5, sl * 5 + 5)
            res = ts.schedule_tti(cqi[ids], avg[ids], user_id=ids)
            # oracle: NVS allocation within a slice = per RBG first max of the slice metric
            kb = rs.link_tables()["kbps"]
            met = kb[cqi[ids]] / ((1 + avg[ids]) / 1000.0)[:, None]
            exp = ids[np.argmax(met, axis=0)]
            np.testing.assert_array_equal(res.rbg_to_user, exp)
            continue
        cell.set_cqi(cqi)
        out = cell.new_out()
        assert cell.allocate(avg, r0, r1, out) == 0
        res = ts.schedule_tti(cqi, avg, r0, r1)
        np.testing.assert_array_equal(res.rbg_to_user, out.rbg_to_user)
        np.testing.assert_array_equal(res.target_rbs, out.target_rbs)
        np.testing.assert_array_equal(res.quota_rbgs, out.quota_rbgs)
        np.testing.assert_array_equal(res.user_nprb, out.user_nprb)
        np.testing.assert_array_equal(res.user_final_cqi, out.user_final_cqi)
        np.testing.assert_array_equal(res.user_mcs, out.user_mcs)
        np.testing.assert_array_equal(res.user_tbs_bits, out.user_tbs_bits)
    ts.close()


@pytest.mark.parametrize("sched", [9, 8, 1])
def test_drop_in_per_prb_cqi(rs, oracle, sched):
    """Per-PRB CQI reports that differ inside an RBG (the simulated-channel case): the metric reads PRB
    rbg*rbg_size, link adaptation every allocated PRB -- against the oracle's per-PRB path."""
    ues, R, G = [5] * 20, 64, 8
    sc = rs.SliceConfig(ues, weight=[0.05] * 20)
    U = sc.n_users
    ts = rs.TtiScheduler(sc, R, G, sched=sched)
    cell = oracle.Cell(ues, R, G, sched, weights=[0.05] * 20)
    rng = np.random.default_rng(17)
    for it in range(8):
        base = synth_cqi(300 + it, (U, R), HIST).astype(np.int16)
        prb = np.clip(np.repeat(base, G, axis=1) + rng.integers(-2, 3, (U, R * G)), 1, 15).astype(np.uint8)
        avg = rng.uniform(1e3, 5e6, U)
        r0, r1 = int(rng.integers(0, 2**31 - 1)), int(rng.integers(0, 2**31 - 1))
        cell.set_cqi_prb(prb)
        out = cell.new_out()
        assert cell.allocate(avg, r0, r1, out) == 0
        res = ts.schedule_tti(None, avg, r0, r1, cqi_prb=prb)
        np.testing.assert_array_equal(res.rbg_to_user, out.rbg_to_user)
        np.testing.assert_array_equal(res.user_nprb, out.user_nprb)
        np.testing.assert_array_equal(res.user_final_cqi, out.user_final_cqi)
        np.testing.assert_array_equal(res.user_tbs_bits, out.user_tbs_bits)
        # and the per-RBG entry point on the same metric grid differs only in link adaptation
        res2 = ts.schedule_tti(prb[:, ::G], avg, r0, r1)
        if sched != 1:
            ts.slice_offset = cell.state()["slice_state"]  # res2 advanced slice_rbs_offset_ a second time
    ts.close()


def test_device_side_range_errors_are_reported(rs, traces):
    sc = rs.SliceConfig([2, 2])
    b = rs.BatchScheduler(sc, 12, 2, 1, sched=9)
    b.seed(np.array([1], np.uint32))
    with pytest.raises(rs.RadioSaberError) as e:
        b.run(10)
    assert "no CQI source" in str(e.value)
    b.upload_cqi_epochs(np.full((1, 1, 4, 12), 7, np.uint8))
    b.run(40)  # exactly the uploaded epoch
    with pytest.raises(rs.RadioSaberError) as e:
        b.run(1)  # TTI 40 needs epoch 1
    assert e.value.code == -5 and "epoch" in str(e.value)
    b.close()
    # trace replay past the uploaded rows
    sc = rs.SliceConfig([2, 2])
    b = rs.BatchScheduler(sc, 64, 8, 1, sched=9)
    b.seed(np.array([1], np.uint32))
    b.set_trace(traces["cqi"][:4, :3], np.zeros((1, 4), np.int32))  # rows 0..2 only: the first report (row 2) fits
    b.run(40)
    with pytest.raises(rs.RadioSaberError) as e:
        b.run(1)  # TTI 140 reads row 3
    assert e.value.code == -5
    b.close()
    with pytest.raises(rs.RadioSaberError) as e:
        rs.TtiScheduler(rs.SliceConfig([2, 2]), 12, 2).schedule_tti(np.zeros((4, 12), np.uint8), np.ones(4))
    assert "outside 1..15" in str(e.value)


@pytest.mark.parametrize("sched,ues,R,G,threads", [(9, [25] * 20, 25, 4, 0), (9, [5] * 20, 64, 8, 0), (9, [3, 7, 0, 1, 12], 25, 4, 128),
                                                    (8, [25] * 20, 25, 4, 0), (7, [25] * 20, 64, 8, 256), (1, [50] * 20, 25, 4, 0)])
def test_shape_specialised_kernels(rs, oracle, sched, ues, R, G, threads):
    """The hiprtc build of the same source with the cell shape as compile-time constants."""
    _check_batch(rs, oracle, sched, ues, R, G, n_cells=2, n_ttis=90, threads=threads, jit=True)


def test_full_size_batch_properties_and_sampled_parity(rs, oracle):
    """BASELINE configs[3] at full size (512 cells x 20x25 UEs x 25 RBGs, the bench workload, shape-specialised
    kernel): size-independent properties on every cell + bit-exact parity on a sample of cells."""
    ues, R, G, n_cells, n_ttis = [25] * 20, 25, 4, 512, 400
    sc = rs.SliceConfig(ues, weight=[0.05] * 20)
    b = rs.BatchScheduler(sc, R, G, n_cells, sched=9, jit=True)
    seeds = (np.arange(n_cells, dtype=np.uint64) * 2654435761 + 805290992) % (2**31 - 1)
    b.seed(seeds.astype(np.uint32))
    b.synthesize_cqi(99, n_ttis // 40)
    b.run(n_ttis)
    st = b.state()
    # every RBG of every TTI is allocated exactly once
    assert (st["cum_rbs"].sum(axis=1) == R * G * n_ttis).all()
    # weighted fairness: slice_rbs_offset_ carries the remainder, so every slice's PRB total stays within
    # one cell-width of its share
    per_slice = st["cum_rbs"].reshape(n_cells, 20, 25).sum(axis=2)
    assert np.abs(per_slice - 0.05 * R * G * n_ttis).max() <= R * G
    # the carried offsets are integers and sum to ~0
    assert (st["slice_state"] == np.round(st["slice_state"])).all()
    assert np.abs(st["slice_state"].sum(axis=1)).max() <= R * G
    # bytes are consistent with the per-slice reduction the multi-GPU run all-reduces
    np.testing.assert_array_equal(b.slice_bytes().astype(np.int64),
                                  st["cum_bytes"].reshape(n_cells, 20, 25).sum(axis=(0, 2)))
    assert (st["avg_rate"] >= 1).all() and np.isfinite(st["avg_rate"]).all()
    for c in (0, 1, 63, 64, 255, 256, 510, 511):
        cell = oracle.Cell(ues, R, G, 9, weights=[0.05] * 20)
        cell.run_synth(b.download_cqi_epochs(c), int(seeds[c]), n_ttis, log=False)
        ost = cell.state()
        np.testing.assert_array_equal(st["cum_bytes"][c], ost["cum_bytes"], err_msg=f"cell {c}")
        assert st["avg_rate"][c].tobytes() == ost["avg_rate"].tobytes(), f"cell {c}"
        assert st["slice_state"][c].tobytes() == ost["slice_state"].tobytes(), f"cell {c}"
    b.close()


@pytest.mark.parametrize("sched", [9, 8, 7])
def test_drop_in_customised_slices(rs, oracle, sched):
    """algo_alpha = 1 slices (metric 0 while the prioritized bearer is empty, times the
    head-of-line delay when algo_beta = 1; sched 7 always multiplies it) -- drop-in mode vs the oracle."""
    ues = [6, 5, 7, 4, 8, 6]
    alpha = [0, 1, 1, 1, 1, 0]
    beta = [0, 0, 1, 1, 1, 0]
    eps = [1, 1, 1, 1, 0, 1]
    psi = [1, 1, 1, 0, 1, 0]
    S, R, G = len(ues), 25, 4
    w = [1.0 / S] * S
    sc = rs.SliceConfig(ues, weight=w, algo_alpha=alpha, algo_beta=beta, algo_epsilon=eps, algo_psi=psi)
    U = sc.n_users
    ts = rs.TtiScheduler(sc, R, G, sched=sched)
    cell = oracle.Cell(ues, R, G, sched, weights=w, epsilon=eps, psi=psi, alpha=alpha, beta=beta)
    rng = np.random.default_rng(23)
    starts = np.cumsum([0] + ues)
    for it in range(14):
        cqi = synth_cqi(500 + it, (U, R), HIST)
        avg = rng.uniform(1e3, 5e6, U)
        hol = rng.uniform(1e-5, 0.4, U)
        prio = (rng.random(U) < 0.7).astype(np.uint8)
        if it % 4 == 1:
            prio[starts[2]:starts[3]] = 0          # a whole customised slice without prioritized data
        if it % 4 == 2:
            avg[:] = 98000.0                       # exact ties: HoL alone separates the users
        if it % 4 == 3:
            hol[starts[3]:starts[4]] = hol[starts[3]]  # equal delays inside a slice
        r0, r1 = int(rng.integers(0, 2**31 - 1)), int(rng.integers(0, 2**31 - 1))
        cell.set_cqi(cqi)
        cell.set_queue_state(hol, prio)
        if sched == 7:
            sl = it % S
            ids = np.arange(starts[sl], starts[sl + 1])
            res = ts.schedule_tti(cqi[ids], avg[ids], user_id=ids, hol_delay=hol[ids], prio_has_data=prio[ids])
            # oracle: per RBG the first maximum of the NVS slice metric from lowest()
            kb = rs.link_tables()["kbps"]
            num = kb[cqi[ids]] if eps[sl] else np.ones((len(ids), R))
            den = ((1 + avg[ids]) / 1000.0)[:, None] if psi[sl] else np.ones((len(ids), 1))
            if alpha[sl]:
                met = np.where(prio[ids][:, None] != 0, hol[ids][:, None] * num / den, 0.0)
            else:
                met = num / den
            np.testing.assert_array_equal(res.rbg_to_user, ids[np.argmax(met, axis=0)], err_msg=f"it {it}")
            continue
        out = cell.new_out()
        assert cell.allocate(avg, r0, r1, out) == 0
        res = ts.schedule_tti(cqi, avg, r0, r1, hol_delay=hol, prio_has_data=prio)
        np.testing.assert_array_equal(res.rbg_to_user, out.rbg_to_user, err_msg=f"it {it}")
        np.testing.assert_array_equal(res.quota_rbgs, out.quota_rbgs)
        np.testing.assert_array_equal(res.user_tbs_bits, out.user_tbs_bits)
    with pytest.raises(rs.RadioSaberError) as e:
        ids = np.arange(starts[2], starts[3])  # a customised slice, no hol_delay given
        ts.schedule_tti(cqi[ids], avg[ids], 1, 2, user_id=ids)
    assert "hol_delay" in str(e.value)
    ts.close()


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_random_shapes_all_schedulers(rs, oracle, seed):
    """Randomised cell shapes (ragged slices, empty slices, random weights, every workgroup size class and with it every
    register/LDS variant of the sort), built-in and shape-specialised kernels, bit-exact against the oracle."""
    rng = np.random.default_rng(seed)
    for trial in range(6):
        S = int(rng.integers(1, 24))
        ues = [int(x) for x in rng.integers(0, 14, S)]
        if sum(ues) == 0:
            ues[0] = 3
        R, G = [(25, 4), (64, 8), (12, 2), (50, 2), (17, 3), (33, 3)][int(rng.integers(0, 6))]
        w = rng.uniform(0.2, 1.0, S)
        w = [float(x) for x in w / w.sum()]
        sched = [9, 9, 9, 8, 7, 1, 103, 10, 11, 101][int(rng.integers(0, 10))]
        threads = [0, 64, 128, 256, 512][int(rng.integers(0, 5))]
        if sched == 10 and threads and R * S > 4 * threads:
            threads = 0  # UpperBound needs R*S <= 4 * threads; 0 lets the library choose
        psi = [int(x) for x in rng.integers(0, 2, S)] if sched != 1 else None
        eps = [1] * S if sched != 1 else None
        _check_batch(rs, oracle, sched, ues, R, G, n_cells=2, n_ttis=int(rng.integers(41, 90)), threads=threads,
                     eps=eps, psi=psi, seed=seed * 100 + trial, weights=w, jit=bool(trial % 2), phy=int(rng.integers(0, 2)))


def test_maximum_sizes(rs, oracle):
    """The limits of the ABI: 64 slices x 64 RBGs (4 096 sort records: the any-size LDS form of the sort),
    1 024 UEs, and both at once as far as 160 KiB of LDS allow."""
    _check_batch(rs, oracle, 9, [2] * 64, 64, 8, n_cells=2, n_ttis=45)            # R*S = 4096, U = 128
    _check_batch(rs, oracle, 9, [2] * 64, 64, 8, n_cells=1, n_ttis=45, jit=True)
    _check_batch(rs, oracle, 9, [16] * 64, 25, 4, n_cells=1, n_ttis=45)           # U = 1024 (RS_MAX_USERS), S = 64
    _check_batch(rs, oracle, 8, [512, 512], 64, 8, n_cells=1, n_ttis=45)          # U = 1024 x 64 RBGs
    _check_batch(rs, oracle, 1, [512, 512], 64, 8, n_cells=1, n_ttis=45)
    _check_batch(rs, oracle, 7, [1024], 64, 8, n_cells=1, n_ttis=45)


def test_subopt_policy_on_the_device(rs, oracle):
    """ORACLE UNPINNED (SubOpt has no reference output, tests/PINS.md): this proves device == oracle, nothing more.
    N4: SubOpt (ref: downlink-transport-scheduler.cpp:274-349).  Ties between equal efficiency losses follow the order
    libstdc++'s unordered_map yields the under-quota slices: the oracle uses the real container, the device its own
    restatement of the hashtable (rs_umap_order, checked against the container on the CPU).  More than 13 / 29 / 59
    under-quota slices exercise the rehashes; skewed weights make slices over and under quota every TTI."""
    _check_batch(rs, oracle, 101, [5] * 20, 25, 4, n_cells=3, n_ttis=90)
    _check_batch(rs, oracle, 101, [5] * 20, 64, 8, n_cells=2, n_ttis=50)
    _check_batch(rs, oracle, 101, [3, 7, 0, 1, 12], 25, 4, n_cells=2, n_ttis=50)                 # ragged, one empty slice
    _check_batch(rs, oracle, 101, [10] * 5, 25, 4, n_cells=2, n_ttis=90, weights=[0.62, 0.3, 0.05, 0.02, 0.01])
    _check_batch(rs, oracle, 101, [2] * 64, 64, 2, n_cells=2, n_ttis=60)                          # 64 slices: 13 -> 29 -> 59 -> 127 buckets
    _check_batch(rs, oracle, 101, [3] * 40, 50, 2, n_cells=2, n_ttis=60)
    _check_batch(rs, oracle, 101, [2] * 64, 33, 3, n_cells=1, n_ttis=45, jit=True)
    _check_batch(rs, oracle, 101, [25] * 20, 25, 4, n_cells=1, n_ttis=45, jit=True)


def test_vogel_policy_on_the_device(rs, oracle):
    """N4: VogelApproximate (ref: downlink-transport-scheduler.cpp:378-451).  The oracle's Vogel is pinned against the
    reference's own unit code (tests/test_oracle_pins.py); here the device against the oracle, whole TTI loops."""
    _check_batch(rs, oracle, 103, [5] * 20, 25, 4, n_cells=3, n_ttis=90)
    _check_batch(rs, oracle, 103, [5] * 20, 64, 8, n_cells=2, n_ttis=50)
    _check_batch(rs, oracle, 103, [3, 7, 0, 1, 12], 25, 4, n_cells=2, n_ttis=50)                 # ragged, one empty slice
    _check_batch(rs, oracle, 103, [10] * 5, 25, 4, n_cells=2, n_ttis=90, weights=[0.62, 0.3, 0.05, 0.02, 0.01])
    _check_batch(rs, oracle, 103, [2] * 64, 33, 3, n_cells=1, n_ttis=45, jit=True)                # 64 slices
    _check_batch(rs, oracle, 103, [25] * 20, 25, 4, n_cells=1, n_ttis=45, jit=True)



def test_upper_bound_policy_on_the_device(rs, oracle):
    """ORACLE UNPINNED (no reference output for sched 10; the rbg_to_user "lowest slice wins" report is this build's own convention,
    tests/PINS.md): this proves device == oracle, nothing more.
    N4: UpperBound (sched 10; ref: downlink-transport-scheduler.cpp:223-246, :603-616): one unstable std::sort per
    slice (run as one segmented level-synchronous pass on the device), the top-quota RBGs per slice, link adaptation over
    each UE's RBGs in push order.  Ties at the quota boundary are the norm with 16 key levels, so per-UE PRB counts, TBS
    and the cumulative counters pin the sort order.  Oracle restated from the cited lines (no reference output exists)."""
    _check_batch(rs, oracle, 10, [5] * 20, 25, 4, n_cells=3, n_ttis=90)
    _check_batch(rs, oracle, 10, [5] * 20, 64, 8, n_cells=2, n_ttis=60)                  # R = 64: real introsort levels
    _check_batch(rs, oracle, 10, [3, 7, 0, 1, 12], 64, 8, n_cells=2, n_ttis=50)           # ragged, one empty slice
    _check_batch(rs, oracle, 10, [10] * 5, 33, 3, n_cells=2, n_ttis=90, weights=[0.62, 0.3, 0.05, 0.02, 0.01])
    _check_batch(rs, oracle, 10, [2] * 32, 64, 8, n_cells=1, n_ttis=45, jit=True)         # 2 048 records
    _check_batch(rs, oracle, 10, [25] * 20, 25, 4, n_cells=1, n_ttis=45, jit=True)
    _check_batch(rs, oracle, 10, [4] * 6, 64, 8, n_cells=2, n_ttis=45, threads=128)       # three positions per thread
    with pytest.raises(rs.RadioSaberError, match="exceeds"):
        rs.BatchScheduler(rs.SliceConfig([2] * 64), 64, 8, 1, sched=10)                   # 4 096 records


def test_nvs_nongreedy_sampler_on_the_device(rs, oracle):
    """ORACLE UNPINNED (no reference output for sched 11, tests/PINS.md): this proves device == oracle, nothing more.
    N4: the CLI's scheduler 11 (DownlinkNVSScheduler, is_nongreedy_; ref: downlink-nvs-scheduler.cpp:405-528): 300 sampled
    CQI-index vectors per TTI from the libc rand() stream (generated on the device 31 ring words per step), per sample a
    per-RBG first-maximum scan, the first best sample applied.  Oracle restated from the cited lines (no reference output
    exists); the device must reproduce it bit for bit, rand() coupling with the error-model draws included."""
    _check_batch(rs, oracle, 11, [5] * 20, 25, 4, n_cells=3, n_ttis=60)
    _check_batch(rs, oracle, 11, [5] * 20, 64, 8, n_cells=2, n_ttis=45, phy=1)
    _check_batch(rs, oracle, 11, [10, 20, 30], 25, 4, n_cells=2, n_ttis=45, phy=1)       # the exp-nongreedy slice sizes
    _check_batch(rs, oracle, 11, [3, 7, 0, 1, 12], 33, 3, n_cells=2, n_ttis=45)          # ragged, one empty slice
    _check_batch(rs, oracle, 11, [70, 300], 25, 4, n_cells=1, n_ttis=42, jit=True)       # a batch holds 27 samples of 300 UEs
    _check_batch(rs, oracle, 11, [30] * 4, 25, 4, n_cells=2, n_ttis=45, jit=True, threads=256)
    # round 4: slices of up to 64 users compare 16-bit keys (ranked metrics) instead of doubles -- 64 users exactly, a 200-user slice
    # on doubles in the same cell, every user starting from one average (whole classes of equal metrics share a rank)
    _check_batch(rs, oracle, 11, [64, 40, 200], 12, 2, n_cells=2, n_ttis=60, jit=True)
    _check_batch(rs, oracle, 11, [64, 40, 200], 12, 2, n_cells=1, n_ttis=40)


def test_nvs_nongreedy_drop_in(rs, oracle):
    """ORACLE UNPINNED (sched 11): device == oracle through the drop-in entry point."""
    ues, R, G = [6] * 5, 25, 4
    sc = rs.SliceConfig(ues)
    ts = rs.TtiScheduler(sc, R, G, sched=11)
    cell = oracle.Cell(ues, R, G, 11)
    rng = np.random.default_rng(21)
    for it in range(8):
        cqi = synth_cqi(500 + it, (sc.n_users, R), HIST)
        avg = rng.uniform(1e3, 5e6, sc.n_users)
        if it % 3 == 0:
            avg[:] = 98000.0
        sl = it % 5
        ids = np.arange(sl * 6, sl * 6 + 6)
        draws = rng.integers(0, 2**31 - 1, 300 * len(ids)).astype(np.int32)
        cell.set_cqi(cqi)
        out = cell.new_out()
        assert cell.allocate_nongreedy(avg, sl, draws, out) == 0
        res = ts.schedule_tti(cqi[ids], avg[ids], user_id=ids, rand_draws=draws)
        np.testing.assert_array_equal(res.rbg_to_user, out.rbg_to_user)
        np.testing.assert_array_equal(res.user_nprb, out.user_nprb[ids])
        np.testing.assert_array_equal(res.user_final_cqi, out.user_final_cqi[ids])
        np.testing.assert_array_equal(res.user_tbs_bits, out.user_tbs_bits[ids])
    with pytest.raises(rs.RadioSaberError, match="rand_draws"):
        ts.schedule_tti(cqi[ids], avg[ids], user_id=ids)
    ts.close()


def _experiment_cases():
    cfgs = json.loads((GOLDEN / "experiment_configs.json").read_text())
    cases = []
    for name, c in sorted(cfgs.items()):
        if any(c["algo_alpha"]):
            # customised slices read queue state: the batch's queue model runs this configuration with its own traffic and all
            # three schedulers of its run script (tests/test_gpu_queues.py::test_customize_20slices_experiment_runs_as_a_batch);
            # this test keeps every flow backlogged
            continue
        for sched in c["schedulers_in_run_scripts"]:
            cases.append((name, sched))
    return cfgs, cases


def test_every_shipped_experiment_configuration(rs, oracle):
    """The slice configurations of the reference's NSDI23-radiosaber-experiments directories (ues_per_slice, weights,
    algo parameters; tests/golden/experiment_configs.json, distilled by tools/make_config_fixture.py) x the scheduler
    numbers their run scripts pass, on the as-shipped 64-RBG grid: device == oracle, bit for bit."""
    cfgs, cases = _experiment_cases()
    assert len(cases) > 100
    for k, (name, sched) in enumerate(cases):
        c = cfgs[name]
        S = len(c["ues_per_slice"])
        sc = rs.SliceConfig(c["ues_per_slice"], weight=c["weight"], algo_epsilon=c["algo_epsilon"], algo_psi=c["algo_psi"])
        U, R, G, n_ttis = sc.n_users, 64, 8, 42
        grids = synth_cqi(900 + k, (1, 2, U, R), HIST)
        seeds = np.array([805290992 + k], np.uint32)
        b = rs.BatchScheduler(sc, R, G, 1, sched=sched, phy_error_draws=True)
        b.seed(seeds)
        b.upload_cqi_epochs(grids)
        b.run(n_ttis)
        st = b.state()
        cell = oracle.Cell(c["ues_per_slice"], R, G, sched, weights=c["weight"], epsilon=c["algo_epsilon"], psi=c["algo_psi"])
        cell.run_synth(grids[0], int(seeds[0]), n_ttis, phy_error_draws=1, log=False)
        ost = cell.state()
        assert (st["cum_bytes"][0] == ost["cum_bytes"]).all(), (name, sched)
        assert (st["cum_rbs"][0] == ost["cum_rbs"]).all(), (name, sched)
        assert st["avg_rate"][0].tobytes() == ost["avg_rate"].tobytes(), (name, sched)
        assert st["slice_state"][0].tobytes() == ost["slice_state"].tobytes(), (name, sched)
        b.close()


def test_drop_in_nvs_big_slice(rs, oracle):
    """sched 7 with slices of more than 32 users on average takes the split scan (runs of the served slice reduced per
    RBG); here through the drop-in entry point, against the first-maximum rule written out in numpy."""
    ues, R, G = [45, 40], 25, 4
    sc = rs.SliceConfig(ues)
    ts = rs.TtiScheduler(sc, R, G, sched=7)
    rng = np.random.default_rng(31)
    kb = rs.link_tables()["kbps"]
    for it in range(6):
        cqi = synth_cqi(700 + it, (sc.n_users, R), HIST)
        avg = rng.uniform(1e3, 5e6, sc.n_users)
        if it % 2 == 0:
            avg[:] = 98000.0  # exact ties: the FIRST maximum must win across run boundaries
        ids = np.arange(0, 45) if it % 2 else np.arange(45, 85)
        res = ts.schedule_tti(cqi[ids], avg[ids], user_id=ids)
        met = kb[cqi[ids]] / ((1 + avg[ids]) / 1000.0)[:, None]
        np.testing.assert_array_equal(res.rbg_to_user, ids[np.argmax(met, axis=0)])
    ts.close()


def test_upper_bound_drop_in_lists(rs, oracle):
    """ORACLE UNPINNED (sched 10): device == oracle.  RS_SCHED_UPPERBOUND through the drop-in entry point: besides the per-user results, the per-slice lists the
    reference's apply step walks (RBGs in push order = the slice's std::sort order, and the user each one goes to)."""
    ues, R, G = [5] * 20, 64, 8
    sc = rs.SliceConfig(ues, weight=[0.05] * 20)
    ts = rs.TtiScheduler(sc, R, G, sched=10)
    cell = oracle.Cell(ues, R, G, 10, weights=[0.05] * 20)
    rng = np.random.default_rng(41)
    for it in range(8):
        cqi = synth_cqi(800 + it, (sc.n_users, R), HIST)
        avg = rng.uniform(1e3, 5e6, sc.n_users)
        if it % 3 == 0:
            avg[:] = 98000.0
        r0, r1 = int(rng.integers(0, 2**31 - 1)), int(rng.integers(0, 2**31 - 1))
        cell.set_cqi(cqi)
        out = cell.new_out()
        assert cell.allocate(avg, r0, r1, out) == 0
        res = ts.schedule_tti(cqi, avg, r0, r1)
        np.testing.assert_array_equal(res.upper_rbg, out.upper_rbg)
        np.testing.assert_array_equal(res.upper_user, out.upper_user)
        np.testing.assert_array_equal(res.user_nprb, out.user_nprb)
        np.testing.assert_array_equal(res.user_tbs_bits, out.user_tbs_bits)
        np.testing.assert_array_equal(res.quota_rbgs, out.quota_rbgs)
        # a slice's list is as long as its positive quota, and several slices may hold the same RBG
        assert ((res.upper_rbg >= 0).sum(axis=1) == np.clip(res.quota_rbgs, 0, R)).all()
    ts.close()


@pytest.mark.parametrize("sched", [9, 7, 11, 10, 1])
def test_state_carries_across_launches(rs, oracle, sched):
    """A run cut into launches of uneven length (not aligned with the 40-TTI CQI refresh) ends in the same state as the
    oracle's single run: PF averages, pending transmitted bytes, the rand() ring (incl. the error-model draws owed for the
    last TTI of a launch), clock, CQI epoch position, slice offsets / NVS EWMA."""
    ues, R, G, n_cells = [6] * 7, 25, 4, 3
    cuts = [17, 40, 1, 33, 12]
    total = sum(cuts)
    sc = rs.SliceConfig(ues)
    grids = synth_cqi(77, (n_cells, (total + 39) // 40, sc.n_users, R), HIST)
    seeds = np.arange(n_cells, dtype=np.uint32) + 1234
    for jit in (False, True):
        b = rs.BatchScheduler(sc, R, G, n_cells, sched=sched, phy_error_draws=True, jit=jit)
        b.seed(seeds)
        b.upload_cqi_epochs(grids)
        for n in cuts:
            b.run(n)
        assert b.ttis_done == total
        st = b.state()
        for c in range(n_cells):
            cell = oracle.Cell(ues, R, G, sched)
            cell.run_synth(grids[c], int(seeds[c]), total, phy_error_draws=1, log=False)
            ost = cell.state()
            np.testing.assert_array_equal(st["cum_bytes"][c], ost["cum_bytes"])
            np.testing.assert_array_equal(st["cum_rbs"][c], ost["cum_rbs"])
            assert st["avg_rate"][c].tobytes() == ost["avg_rate"].tobytes()
            assert st["slice_state"][c].tobytes() == ost["slice_state"].tobytes()
        b.close()


def test_trace_replay_across_launches(rs, oracle, traces):
    """Trace mode, launches that start between two CQI reports: the last reported trace row is reloaded (LDS does not
    survive a launch) and the report clock keeps running."""
    ues = [5] * 20
    sc = rs.SliceConfig(ues, weight=[0.05] * 20)
    U = sc.n_users
    b = rs.BatchScheduler(sc, 64, 8, 1, sched=9, phy_error_draws=True)
    b.seed(np.array([805290992], np.uint32), np.array([4321], np.int64))
    b.set_trace(traces["cqi"], traces["mapping"][2][np.arange(U) % 474][None, :])
    for n in (13, 40, 27, 1, 59):
        b.run(n)
    st = b.state()
    cell = oracle.Cell(ues, 64, 8, 9, weights=[0.05] * 20)
    cell.run_trace(traces["cqi"], traces["mapping"][2], 805290992, 4321, 140)
    ost = cell.state()
    np.testing.assert_array_equal(st["cum_bytes"][0], ost["cum_bytes"])
    np.testing.assert_array_equal(st["cum_rbs"][0], ost["cum_rbs"])
    assert st["avg_rate"][0].tobytes() == ost["avg_rate"].tobytes()
    assert st["slice_state"][0].tobytes() == ost["slice_state"].tobytes()
    b.close()


def test_two_batches_interleaved_and_other_refresh_period(rs, oracle):
    """Two live batches on one device (different schedulers, one shape-specialised, one with a 7-TTI CQI refresh) run
    turn by turn; each ends where the oracle ends."""
    ues, R, G = [4] * 9, 25, 4
    sc = rs.SliceConfig(ues)
    U = sc.n_users
    total = 63
    g1 = synth_cqi(91, (2, (total + 39) // 40, U, R), HIST)
    g2 = synth_cqi(92, (2, (total + 6) // 7, U, R), HIST)
    seeds = np.array([5, 6], np.uint32)
    b1 = rs.BatchScheduler(sc, R, G, 2, sched=9, jit=True)
    b2 = rs.BatchScheduler(sc, R, G, 2, sched=8, cqi_refresh=7, phy_error_draws=True)
    for b, g in ((b1, g1), (b2, g2)):
        b.seed(seeds)
        b.upload_cqi_epochs(g)
    for n in (10, 20, 5, 28):
        b1.run_async(n)
        b2.run_async(n)
    b1.sync()
    b2.sync()
    for b, g, sched, refresh, phy in ((b1, g1, 9, 40, 0), (b2, g2, 8, 7, 1)):
        st = b.state()
        for c in range(2):
            cell = oracle.Cell(ues, R, G, sched)
            cell.run_synth(g[c], int(seeds[c]), total, refresh=refresh, phy_error_draws=phy, log=False)
            ost = cell.state()
            np.testing.assert_array_equal(st["cum_bytes"][c], ost["cum_bytes"])
            assert st["avg_rate"][c].tobytes() == ost["avg_rate"].tobytes()
            assert st["slice_state"][c].tobytes() == ost["slice_state"].tobytes()
        b.close()


@pytest.mark.parametrize("sched", [9, 8, 7, 1, 10, 11, 103, 101])
def test_degenerate_shapes(rs, oracle, sched):
    """One RBG, one UE, one slice, an empty slice between two others, one UE on 64 RBGs, 64 UEs on one RBG: no sort level,
    no run, no batch is ever full -- every scheduler, both kernel flavours."""
    for ues, R, G in (([1], 1, 1), ([1, 1], 1, 1), ([2], 2, 1), ([1, 2, 1], 3, 2), ([3], 1, 4), ([1] * 5, 2, 2),
                      ([2, 0, 2], 4, 1), ([1], 64, 8), ([64], 1, 1)):
        for jit in (False, True):
            _check_batch(rs, oracle, sched, ues, R, G, n_cells=2, n_ttis=45, jit=jit, phy=1)


def test_experiment_runner_tool(rs, oracle, tmp_path):
    """tools/run_experiment.py = one line of the reference's run scripts: reads ue<id>.log / mapping.config / the slice
    JSON, runs the cell on the GPU, writes the reference's stderr lines.  Checked on files written here: the last
    cumu_bytes / cumu_rbs of every flow equal the oracle's trace run, and the reducer of plot_throughput.py parses it."""
    import subprocess
    import sys as _sys
    from pathlib import Path
    from radiosaber_amd import logfmt
    root = Path(__file__).resolve().parents[1]
    rng = np.random.default_rng(55)
    n_traces, rows = 6, 475
    grid = rng.integers(1, 16, (n_traces, rows, 64)).astype(np.uint8)
    for t in range(n_traces):
        (tmp_path / f"ue{t}.log").write_text("".join(" ".join(str(v) for v in np.repeat(r, 8)) + " \n" for r in grid[t]))
    mapping = np.array([3, 0, 5, 1, 2, 4, 1, 0], np.int32)
    (tmp_path / "mapping.config").write_text("".join(f"{i} {m}\n" for i, m in enumerate(mapping)))
    cfg = {"ues_per_slice": [3, 2, 4], "slices": [{"n_slices": 3, "weight": 1 / 3, "algo_alpha": 0, "algo_beta": 0,
                                                    "algo_epsilon": 1, "algo_psi": 1}]}
    (tmp_path / "config.json").write_text(json.dumps(cfg))
    log = tmp_path / "run.log"
    for sched in (9, 1):
        r = subprocess.run([_sys.executable, str(root / "tools" / "run_experiment.py"), "--sched", str(sched), "--seed", "1",
                            "--duration", "0.15", "--config", str(tmp_path / "config.json"), "--traces", str(tmp_path),
                            "--rand-skip", "77", "--log", str(log)], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        lines = log.read_text().strip().split("\n")
        cell = oracle.Cell([3, 2, 4], 64, 8, sched)
        cell.run_trace(grid, mapping, 749913912, 77, 150)
        st = cell.state()
        last = {}
        for ln in lines:
            w = ln.split(" ")
            last[int(w[2])] = (int(w[4]), int(w[6]))
        for u in range(9):
            assert last[u] == (int(st["cum_bytes"][u]), int(st["cum_rbs"][u])), (sched, u)
        if sched == 9:
            assert lines[0].startswith("100 app: ") and " user: " in lines[0] and " slice: " in lines[0]
            mbps, _ = logfmt.slice_throughput_from_log(lines, 9, 3, begin_ts=0, end_ts=250)
            assert len(mbps) == 3 and all(x > 0 for x in mbps)
        else:
            assert lines[0].startswith("100 flow: ") and "user:" not in lines[0]
