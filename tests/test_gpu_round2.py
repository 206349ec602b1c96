"""GPU tests added in round 2 (through the C ABI): reference-compiled pins exercised on the device's own values, the
multi-GPU sharding on real device results, the register-resident cumulative counters, RCCL loaded once."""
import ctypes as C
import json
import os

import numpy as np
import pytest
import torch  # noqa: F401  (before the product library first touches HIP: torch ships its own HIP runtime and must initialise first)

from conftest import GOLDEN, synth_cqi
from test_oracle_pins import cqi_keys_of_eff, ref_maximize_cell

pytestmark = pytest.mark.gpu

HIST = (152600, 56656, 270880, 2088792, 3509504, 1595568, 4145392, 5295816, 1903424,
        6890232, 4770864, 2842552, 3579624, 96000, 1227696)


@pytest.mark.parametrize("jit", [False, True])
def test_device_clock_matches_reference_event_core(rs, jit):
    """t_k and m_lastUpdate on the device against the time stamps the reference's own Simulator/Calendar produced
    (tests/golden/ref_clock.json, recorded from oracle/_ref/libref_clock.so), over launches of uneven length."""
    ref = np.array([float.fromhex(x) for x in json.loads((GOLDEN / "ref_clock.json").read_text())["subframe_start"]])
    sc = rs.SliceConfig([5] * 4)
    b = rs.BatchScheduler(sc, 12, 2, 3, sched=9, jit=jit)
    b.seed(np.arange(3, dtype=np.uint32) + 1)
    b.synthesize_cqi(5, 16)
    t, lu = b.clock()
    assert (t == ref[100]).all() and (lu == 0.1).all()
    done = 0
    for n in (1, 39, 2, 158, 77, 300):
        b.run(n)
        done += n
        t, lu = b.clock()
        assert (t == ref[100 + done]).all(), done
        assert (lu == ref[100 + done - 1]).all(), done
    b.close()


@pytest.mark.parametrize("shape", [([5] * 20, 25, 4), ([25] * 20, 25, 4), ([5] * 20, 64, 8)])
def test_maximize_cell_on_device_tti_inputs_matches_reference_unit_code(rs, oracle, shape):
    """The (flow_spectraleff, quota) the DEVICE handed to its inter-slice step, TTI by TTI, go through the reference's own
    MaximizeCell (unittest/test_tp_algos.cpp compiled in place, oracle/_ref/libref_tp_algos.so): the RBG -> slice map it
    returns must be the one the device applied.  The same dump is compared with the oracle's flow_spectraleff /
    user_index, which pins the metric + arg-max stage (a5, a6) at the intermediate level too."""
    L = oracle.ref_lib("libref_tp_algos.so")
    if L is None:
        pytest.skip("oracle/_ref/libref_tp_algos.so did not travel")
    ues, R, G = shape
    n_cells, n_ttis = 2, 90
    sc = rs.SliceConfig(ues)
    grids = synth_cqi(21, (n_cells, 3, sc.n_users, R), HIST)
    seeds = np.arange(n_cells, dtype=np.uint32) + 4242
    for jit in (False, True):
        b = rs.BatchScheduler(sc, R, G, n_cells, sched=9, jit=jit)
        b.seed(seeds)
        b.upload_cqi_epochs(grids)
        got = b.run_logged(n_ttis, slice_keys=True)
        b.close()
        u2s = sc.user_to_slice
        for c in range(n_cells):
            cell = oracle.Cell(ues, R, G, oracle.SCHED_MAXCELL)
            g = oracle.Rng(int(seeds[c]))
            ticks = oracle.clock_ticks(100, n_ttis)
            cell.set_last_update(0.1)
            out = cell.new_out()
            for n in range(n_ttis):
                if n % 40 == 0:
                    cell.set_cqi(grids[c, n // 40])
                assert cell.step(float(ticks[n]), g.rand(), g.rand(), out) == 0
                keys = np.ascontiguousarray(got["slice_cqi"][c, n], np.int32)
                assert (keys == cqi_keys_of_eff(out.slice_eff)).all(), (c, n)
                assert (got["slice_user"][c, n] == out.slice_user).all(), (c, n)
                want = ref_maximize_cell(L, keys, got["quota"][c, n].astype(np.int32))
                m = got["rbg_to_user"][c, n].astype(np.int64)
                dev = np.where(m >= 0, u2s[np.maximum(m, 0)], -1)
                assert (dev == want).all(), (c, n)


def _run_shard(rs, sharding, sc, R, G, rank, world, cells_per_rank, n_ttis, jit):
    b = rs.BatchScheduler(sc, R, G, cells_per_rank, sched=9, jit=jit)
    b.seed(sharding.seeds_for_cells(sharding.cell_ids_for_rank(rank, world, cells_per_rank)))
    b.synthesize_cqi(0x5AB3, (n_ttis + 39) // 40, first_cell=sharding.first_cell_for_rank(rank, world, cells_per_rank))
    grids = [b.download_cqi_epochs(c) for c in range(cells_per_rank)]
    b.run(n_ttis)
    st, sb = b.state(), b.slice_bytes()
    b.close()
    return st, sb, grids


@pytest.mark.parametrize("jit", [False, True])
def test_cell_trajectories_do_not_depend_on_the_sharding(rs, jit):
    """Global cells 0..7 as ONE batch and as W = 2 shards of 4 (what bench.py does per rank): identical CQI grids,
    per-cell counters and PF state, and the per-slice byte totals add up exactly (the vector RCCL all-reduces)."""
    from radiosaber_amd import sharding
    sc = rs.SliceConfig([5] * 20)
    R, G, n_ttis = 25, 4, 130
    one, sb_one, g_one = _run_shard(rs, sharding, sc, R, G, 0, 1, 8, n_ttis, jit)
    total = np.zeros(20, np.uint64)
    for rank in range(2):
        st, sb, g = _run_shard(rs, sharding, sc, R, G, rank, 2, 4, n_ttis, jit)
        sl = slice(4 * rank, 4 * rank + 4)
        for c in range(4):
            assert (g[c] == g_one[4 * rank + c]).all()
        assert (st["cum_bytes"] == one["cum_bytes"][sl]).all() and (st["cum_rbs"] == one["cum_rbs"][sl]).all()
        assert st["avg_rate"].tobytes() == one["avg_rate"][sl].tobytes()
        total += sb
    assert (total == sb_one).all() and int(sb_one.sum()) == int(one["cum_bytes"].sum())


def test_cumulative_counters_in_registers_across_launches(rs, oracle):
    """The shape-specialised kernel keeps cumu_bytes / cumu_rbs in registers and flushes once per launch; the service of a
    launch's last TTI is counted at the flush and must not be counted again by the next launch's first EWMA update.
    1-TTI launches, uneven launches and one long launch must all equal the oracle, for 1 and 2 users per thread."""
    ues, R, G = [25] * 20, 25, 4
    sc = rs.SliceConfig(ues)
    grids = synth_cqi(8, (1, 4, sc.n_users, R), HIST)
    cell = oracle.Cell(ues, R, G, oracle.SCHED_MAXCELL)
    cell.run_synth(grids[0], 99, 150, log=False)
    want = cell.state()
    for threads, plan in ((512, [150]), (512, [1] * 7 + [143]), (256, [40, 1, 1, 68, 40]), (64, [75, 75])):
        b = rs.BatchScheduler(sc, R, G, 1, sched=9, jit=True, threads_per_cell=threads)
        assert b.jit_status()[0] == 1, b.jit_status()
        b.seed(np.array([99], np.uint32))
        b.upload_cqi_epochs(grids)
        done = 0
        for n in plan:
            b.run(n)
            done += n
            st = b.state()
            assert int(st["cum_rbs"].sum()) == done * R * G  # every RBG granted every TTI, visible after every launch
        assert (st["cum_bytes"][0] == want["cum_bytes"]).all() and (st["cum_rbs"][0] == want["cum_rbs"]).all()
        assert st["avg_rate"][0].tobytes() == want["avg_rate"].tobytes()
        b.close()


def test_jit_status_is_reported(rs):
    sc = rs.SliceConfig([5] * 4)
    b = rs.BatchScheduler(sc, 12, 2, 1, sched=9, jit=True)
    assert b.jit_status() == (1, "") and b.kernel_name == "rs_cell_kernel_jit"
    b.close()
    b = rs.BatchScheduler(sc, 12, 2, 1, sched=9, jit=False)
    assert b.jit_status()[0] == 0 and b.kernel_name != "rs_cell_kernel_jit"
    b.close()


def test_rccl_single_rank_all_reduce_of_slice_bytes(rs):
    """RCCL loaded and used once on this box: a one-rank `nccl` process group all-reduces the device-resident uint64[S]
    per-slice byte vector exactly as bench.py's N > 1 path does."""
    import torch
    import torch.distributed as dist
    from radiosaber_amd import sharding
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        sc = rs.SliceConfig([5] * 20)
        b = rs.BatchScheduler(sc, 25, 4, 4, sched=9, jit=True)
        b.seed(sharding.seeds_for_cells(sharding.cell_ids_for_rank(0, 1, 4)))
        b.synthesize_cqi(0x5AB3, 2)
        b.run(60)
        t = torch.zeros(20, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        b.slice_bytes_into(t.data_ptr())
        b.sync()
        before = t.clone()
        dist.all_reduce(t, op=dist.ReduceOp.SUM)  # world size 1: the collective itself runs (RCCL), the sum is the vector
        torch.cuda.synchronize()
        assert torch.equal(t, before) and int(t.sum().item()) == int(b.state()["cum_bytes"].sum())
        assert (t.cpu().numpy().astype(np.uint64) == b.slice_bytes()).all()
        b.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("sched", [9, 8, 7, 1, 10])
def test_per_prb_cqi_sources_in_batches(rs, oracle, sched):
    """Reports that differ inside an RBG (enb-mac-entity.cc:173-186 stores all 512 PRBs): the batch keeps the per-PRB grids / trace
    rows in HBM, the metric reads each RBG's first PRB (downlink-transport-scheduler.cpp:536), link adaptation every allocated
    PRB (:643-646).  Epoch source and trace source, built-in and shape-specialised kernels."""
    ues, R, G, n_cells, n_ttis = [5] * 6, 16, 4, 2, 90
    sc = rs.SliceConfig(ues)
    U = sc.n_users
    grids = synth_cqi(50 + sched, (n_cells, 3, U, R * G), HIST)
    assert (grids.reshape(n_cells, 3, U, R, G).std(axis=4) > 0).any()
    seeds = np.arange(n_cells, dtype=np.uint32) + 900
    for jit in (False, True):
        b = rs.BatchScheduler(sc, R, G, n_cells, sched=sched, jit=jit)
        b.seed(seeds)
        b.upload_cqi_epochs_prb(grids)
        got = b.run_logged(n_ttis)
        st = b.state()
        b.close()
        for c in range(n_cells):
            cell = oracle.Cell(ues, R, G, sched)
            logs = cell.run_synth(grids[c], int(seeds[c]), n_ttis, per_prb=True)
            np.testing.assert_array_equal(got["rbg_to_user"][c], logs["rbg_to_user"])
            np.testing.assert_array_equal(got["tbs_bits"][c], logs["tbs_bits"])
            assert (st["cum_bytes"][c] == cell.state()["cum_bytes"]).all()
            assert st["avg_rate"][c].tobytes() == cell.state()["avg_rate"].tobytes()
    # the same grids as uniform RBGs give different transport blocks: the per-PRB path is really taken
    flat = np.repeat(grids.reshape(n_cells, 3, U, R, G)[..., :1], G, axis=4).reshape(grids.shape)
    b = rs.BatchScheduler(sc, R, G, n_cells, sched=sched)
    b.seed(seeds)
    b.upload_cqi_epochs_prb(flat)
    other = b.run_logged(n_ttis)
    b.close()
    assert (other["tbs_bits"] != got["tbs_bits"]).any()
    # trace source: 12 traces x 8 rows of per-PRB reports
    tr = synth_cqi(70 + sched, (12, 8, R * G), HIST)
    ut = (np.arange(n_cells * U).reshape(n_cells, U) * 5 % 12).astype(np.int32)
    b = rs.BatchScheduler(sc, R, G, n_cells, sched=sched, phy_error_draws=True, jit=True)
    b.seed(seeds)
    b.set_trace_prb(tr, ut, row_modulus=8)
    got = b.run_logged(n_ttis)
    b.close()
    for c in range(n_cells):
        cell = oracle.Cell(ues, R, G, sched)
        logs = cell.run_trace(tr, ut[c], int(seeds[c]), 0, n_ttis, row_modulus=8, per_prb=True)
        np.testing.assert_array_equal(got["rbg_to_user"][c], logs["rbg_to_user"])
        np.testing.assert_array_equal(got["tbs_bits"][c], logs["tbs_bits"])


def test_drop_in_gates_of_schedulers_7_and_1(rs, oracle):
    """rs_tti_in.required_rbs (sched 7: a user competes only while its PRBs are below m_requiredRBs, downlink-nvs-scheduler.cpp:
    299-300) and rs_tti_in.data_to_transmit (sched 1: a flow leaves once the transport block of its PRBs so far carries its queue,
    downlink-packet-scheduler.cpp:253-265) against plain sequential restatements of the two loops (numpy + the oracle's EESM / TBS
    helpers).  ORACLE-SIDE UNPINNED like the rest of schedulers 1 and 7."""
    R, G = 25, 4
    lt = rs.link_tables()
    tabs = oracle.tables()
    rng = np.random.default_rng(77)
    # ---- sched 7
    ues = [9, 14, 6]
    sc = rs.SliceConfig(ues, algo_alpha=[0, 1, 0], algo_psi=[1, 1, 0])
    ts = rs.TtiScheduler(sc, R, G, sched=7)
    starts = np.cumsum([0] + ues)
    bound = 0
    for it in range(12):
        sl = it % 3
        ids = np.arange(starts[sl], starts[sl + 1])
        n = len(ids)
        cqi = synth_cqi(600 + it, (n, R), HIST)
        avg = rng.uniform(1e3, 5e6, n)
        hol = rng.uniform(1e-5, 0.4, n)
        prio = (rng.random(n) < 0.8).astype(np.uint8)
        req = rng.integers(0, 40, n).astype(np.int32)
        if it % 4 == 3:
            req[:] = 10 ** 6  # backlogged: the gate never binds
        res = ts.schedule_tti(cqi, avg, user_id=ids, hol_delay=hol, prio_has_data=prio, required_rbs=req)
        num = lt["kbps"][cqi] if sc.algo_epsilon[sl] else np.ones((n, R))
        den = ((1 + avg) / 1000.0)[:, None] if sc.algo_psi[sl] else np.ones((n, 1))
        met = np.where(prio[:, None] != 0, hol[:, None] * num / den, 0.0) if sc.algo_alpha[sl] else num / den
        alloc = np.zeros(n, np.int64)
        want = np.full(R, -1)
        for r in range(R):
            ok = alloc < req
            if ok.any():
                k = int(np.flatnonzero(ok)[np.argmax(met[ok, r])])
                want[r] = ids[k]
                alloc[k] += G
        np.testing.assert_array_equal(res.rbg_to_user, want, err_msg=f"sched 7 it {it}")
        np.testing.assert_array_equal(res.user_nprb, alloc)
        bound += int((want < 0).sum())
    assert bound > 0, "the m_requiredRBs gate never left an RBG unallocated"
    ts.close()
    # ---- sched 1: the passed "users" are flows
    n = 12
    sc1 = rs.SliceConfig([n])
    ts = rs.TtiScheduler(sc1, R, G, sched=1)
    left = 0
    for it in range(10):
        cqi = synth_cqi(700 + it, (n, R), HIST)
        avg = rng.uniform(1e3, 5e6, n)
        data = rng.integers(20, 150, n).astype(np.int32)
        if it % 3 == 2:
            data[: n // 2] = 100000000  # backlogged flows beside the finite ones: they take what the others leave
        res = ts.schedule_tti(cqi, avg, data_to_transmit=data)
        met = (lt["eff"][cqi] * 180000.) / avg[:, None]
        done = np.zeros(n, bool)
        prbs = [[] for _ in range(n)]
        want = np.full(R, -1)
        for r in range(R):
            ok = ~done & (met[:, r] > 0)
            if not ok.any():
                continue
            k = int(np.flatnonzero(ok)[np.argmax(met[ok, r])])
            want[r] = k
            prbs[k] += [cqi[k, r]] * G
            fc = oracle.final_cqi(np.array(prbs[k], np.uint8))
            tbs = oracle.lib().rso_tbs_bits(int(tabs["cqi_to_mcs"][fc - 1]), len(prbs[k]))
            if tbs >= int(data[k]) * 8:
                done[k] = True
        np.testing.assert_array_equal(res.rbg_to_user, want, err_msg=f"sched 1 it {it}")
        left += int((want < 0).sum())
    assert left > 0, "every RBG was taken: the satisfied-flow break never emptied the race"
    ts.close()


@pytest.mark.gpu
@pytest.mark.parametrize("extra", ["-DRS_NO_HOLD", "-DRS_NO_HOLD -DRS_NO_SPEC", "-DRS_HOLD_MAX_AGE=3", "-DRS_HOLD_ALWAYS", "-DRS_NO_SPEC"])
def test_opt_in_kernel_variants_stay_bit_exact(rs, oracle, extra, monkeypatch):
    """The four mechanisms that prepare or skip work -- held winners (RS_NO_HOLD / RS_HOLD_ALWAYS / a 3-TTI age cap), the
    speculative next-TTI scan (RS_NO_SPEC) -- each switched the other way from what the shape picks by default (RS_JIT_EXTRA is
    part of the kernel cache key): the code they switch to is the default of OTHER shapes, so it stays under test on these too.
    Against the oracle on the headline shape, the 64-RBG grid and ragged / tiny cells.  (The losing variants of rounds 2-3 --
    cooperative scan, forced greedy forms, ballots per owner bit, ... -- were deleted in round 4; git keeps them.)"""
    from test_gpu_parity import _check_batch
    monkeypatch.setenv("RS_JIT_EXTRA", extra)
    _check_batch(rs, oracle, 9, [25] * 20, 25, 4, n_cells=2, n_ttis=90, jit=True)
    _check_batch(rs, oracle, 9, [25] * 20, 64, 8, n_cells=2, n_ttis=60, jit=True)
    _check_batch(rs, oracle, 9, [3, 0, 7, 1, 12], 12, 2, n_cells=2, n_ttis=60, jit=True, threads=64)
    _check_batch(rs, oracle, 9, [2] * 64, 64, 8, n_cells=1, n_ttis=45, jit=True)
    _check_batch(rs, oracle, 9, [5] * 33, 33, 3, n_cells=2, n_ttis=60, jit=True, phy=1)
