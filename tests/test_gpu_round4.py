"""Round 4: cells of more than 512 threads (the 64-RBG grid at 640), cycling CQI epochs and the streamed-CQI mode, a
driver-visible long run of the held winners incl. the age cap that ends a hold, the ABI-checked create functions, and the
multi-GPU checks that validate themselves on a box with two or more GPUs (skipped on the one-GPU boxes)."""
import ctypes as C
import json
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import numpy as np
import pytest

from conftest import synth_cqi

ROOT = Path(__file__).resolve().parents[1]
HIST = (152600, 56656, 270880, 2088792, 3509504, 1595568, 4145392, 5295816, 1903424,
        6890232, 4770864, 2842552, 3579624, 96000, 1227696)


def _bench(args, env_extra=None, timeout=1200):
    env = dict(os.environ)
    for k in ("RS_JIT_EXTRA", "RS_JIT", "WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "RS_BENCH_BACKEND"):
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, str(ROOT / "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)


def _oracle_final_states(oracle, ues, R, G, sched, weights, grids_of, seeds, n_ttis, refresh=40, eps=None, psi=None):
    """Final state of every cell from the CPU oracle (log=False path), cells spread over the host cores (ctypes drops the GIL)."""
    def one(c):
        cell = oracle.Cell(ues, R, G, sched, weights=weights, epsilon=eps, psi=psi)
        cell.run_synth(grids_of(c), int(seeds[c]), n_ttis, refresh=refresh, log=False)
        return cell.state()
    with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as ex:
        return list(ex.map(one, range(len(seeds))))


def _assert_state_equal(st, c, ost, what=""):
    np.testing.assert_array_equal(st["cum_bytes"][c], ost["cum_bytes"], err_msg=f"{what} cell {c} cum_bytes")
    np.testing.assert_array_equal(st["cum_rbs"][c], ost["cum_rbs"], err_msg=f"{what} cell {c} cum_rbs")
    assert st["avg_rate"][c].tobytes() == ost["avg_rate"].tobytes(), f"{what} cell {c}: PF averages differ"
    assert st["slice_state"][c].tobytes() == ost["slice_state"].tobytes(), f"{what} cell {c}: slice state differs"


# ---------------------------------------------------------------- CPU side

def test_checked_create_refuses_another_abi_or_struct_size(rs):
    """nothing is inferred from field values -- a caller built against another layout is refused by its version and
    struct size, before the library reads a single field."""
    from radiosaber_amd import api
    L = rs.lib()
    cfg = api._CfgHolder(rs.SliceConfig([2, 2]), 25, 4, 9, 0, None)
    assert not L.rs_create_checked(C.byref(cfg.c), api.RS_ABI_VERSION - 1, C.sizeof(api._Config))
    assert b"ABI mismatch" in L.rs_last_error()
    assert not L.rs_create_checked(C.byref(cfg.c), api.RS_ABI_VERSION, C.sizeof(api._Config) - 8)
    assert b"ABI mismatch" in L.rs_last_error()
    bc = api._BatchConfig(cfg.c, 1, 100, 40, 0, 0, 0, 0, 0)
    assert not L.rs_batch_create_checked(C.byref(bc), api.RS_ABI_VERSION, C.sizeof(api._BatchConfig) - 8)  # an ABI-8 rs_batch_config
    assert b"ABI mismatch" in L.rs_last_error()
    assert L.rs_abi_version() == api.RS_ABI_VERSION
    # the header and the ctypes mirror agree on the version and on the trailing fields
    hdr = (ROOT / "include" / "radiosaber_hip.h").read_text()
    assert f"#define RS_ABI_VERSION {api.RS_ABI_VERSION} " in hdr
    assert hdr.index("int32_t cqi_epoch_wrap;") < hdr.index("int32_t queue_state_lds;") < hdr.index("} rs_batch_config;")
    assert [f[0] for f in api._BatchConfig._fields_][-4:] == ["cqi_epoch_wrap", "queue_state_lds", "autotune", "selfcheck"]


def test_jit_builds_without_the_llvm_tuning_options_and_for_640_threads(rs):
    """if a hiprtc update drops the two internal -mllvm switches the library must still get its shape-specialised
    kernel (the built-in ones hold no winners: 2-3 x slower) -- the untuned option set compiles; and the 64-RBG grid's
    640-thread cell (two sort positions per lane, five waves per SIMD) compiles."""
    assert rs.jit_selfcheck(20, 500, 25, 4, 512, rs.RS_SCHED_MAXCELL, untuned=True) > 0
    assert rs.jit_selfcheck(20, 500, 64, 8, 640, rs.RS_SCHED_MAXCELL) > 0
    src = (ROOT / "radiosaber_amd" / "csrc" / "rs_jit.cpp").read_text()
    assert "rs_jit_options(S, U, R, G, NT, sched, qmode, win, false, " in src  # the retry in rs_jit_get: same arguments, tuned = false
    assert rs.lds_bytes_per_cell(20, 500, 64, rs.RS_SCHED_MAXCELL, 640) <= 80 * 1024  # two cells per CU


def test_bench_does_not_start_ranks_under_a_profiler():
    """under `rocprofv3 ... -- python3 bench.py --gpus N` the profiler has initialised the GPU in the parent; starting
    the ranks from there is the hop this pool forbids.  The parent must refuse (no child, rc != 0)."""
    r = _bench(["--gpus", "2", "--steps", "1"], {"ROCPROFILER_REGISTER_FORCE_LOAD": "1"})
    assert r.returncode == 2 and "refusing to start" in r.stderr and "torch.distributed.run" not in r.stderr.split("refusing")[0]
    r = _bench(["--gpus", "2", "--steps", "1"], {"LD_PRELOAD": "/nonexistent/librocprofiler-sdk-tool.so"})
    assert r.returncode == 2 and "refusing to start" in r.stderr


# ---------------------------------------------------------------- GPU side

@pytest.mark.gpu
@pytest.mark.parametrize("sched,threads", [(9, 640), (9, 1024), (10, 640), (8, 640)])
def test_cells_of_more_than_512_threads(rs, oracle, sched, threads):
    """Shape-specialised kernels take up to 1 024 threads per cell; 640 gives the 1 280 sort records of the 64-RBG grid two
    positions per lane on every wave.  Bit-exact against the oracle like every other workgroup size."""
    ues, R, G, n_cells, n_ttis = [25] * 20, 64, 8, 3, 90
    sc = rs.SliceConfig(ues)
    grids = synth_cqi(41, (n_cells, 3, sc.n_users, R), HIST)
    seeds = np.arange(n_cells, dtype=np.uint32) * 31 + 7
    b = rs.BatchScheduler(sc, R, G, n_cells, sched=sched, threads_per_cell=threads, jit=True)
    assert b.kernel_name == "rs_cell_kernel_jit"
    b.seed(seeds)
    b.upload_cqi_epochs(grids)
    got = b.run_logged(50)
    b.run(n_ttis - 50)
    st = b.state()
    for c in range(n_cells):
        cell = oracle.Cell(ues, R, G, sched)
        logs = cell.run_synth(grids[c], int(seeds[c]), n_ttis)
        if sched != 10:  # (UpperBound's rbg_to_user is a convention of this build; its per-UE outputs are complete)
            np.testing.assert_array_equal(got["rbg_to_user"][c], logs["rbg_to_user"][:50])
        np.testing.assert_array_equal(got["tbs_bits"][c], logs["tbs_bits"][:50])
        _assert_state_equal(st, c, cell.state(), f"sched {sched} x {threads} threads")
    b.close()


@pytest.mark.gpu
def test_more_than_512_threads_needs_the_shape_specialised_kernel(rs):
    with pytest.raises(rs.RadioSaberError, match="threads_per_cell"):
        rs.BatchScheduler(rs.SliceConfig([5] * 4), 25, 4, 2, threads_per_cell=640, jit=False)


@pytest.mark.gpu
@pytest.mark.parametrize("sched,jit,refresh,R,G", [(9, True, 1, 25, 4), (9, False, 1, 25, 4), (8, True, 1, 25, 4), (9, True, 7, 64, 8),
                                                   (7, True, 1, 25, 4), (1, False, 3, 25, 4), (1, True, 1, 25, 4), (101, True, 1, 25, 4),
                                                   (103, True, 1, 25, 4), (103, False, 2, 25, 4), (8, False, 1, 64, 8), (7, False, 1, 25, 4)])
def test_cycling_epochs_and_streamed_cqi(rs, oracle, sched, jit, refresh, R, G):
    """rs_batch_config.cqi_epoch_wrap: a bounded set of grids serves a run of any length (epoch index modulo n_epochs), across
    launches too; cqi_refresh = 1 is SURVEY 8d's streamed-CQI mode (a grid from HBM every TTI).  The oracle gets the same
    grids tiled.  Without the flag the run still ends with RS_ERR_RANGE."""
    ues, n_cells, n_ttis, n_epochs = [5] * 20, 3, 75, 5
    sc = rs.SliceConfig(ues)
    grids = synth_cqi(77 + sched, (n_cells, n_epochs, sc.n_users, R), HIST)
    seeds = np.arange(n_cells, dtype=np.uint32) + 99
    b = rs.BatchScheduler(sc, R, G, n_cells, sched=sched, jit=jit, cqi_refresh=refresh, cqi_epoch_wrap=True)
    b.seed(seeds)
    b.upload_cqi_epochs(grids)
    got = b.run_logged(31)
    b.run(13)  # uneven launches: the wrapped epoch index is re-derived from the TTI count
    b.run(n_ttis - 44)
    st = b.state()
    reps = (n_ttis + refresh * n_epochs - 1) // (refresh * n_epochs)
    for c in range(n_cells):
        cell = oracle.Cell(ues, R, G, sched)
        logs = cell.run_synth(np.tile(grids[c], (reps, 1, 1)), int(seeds[c]), n_ttis, refresh=refresh)
        np.testing.assert_array_equal(got["rbg_to_user"][c], logs["rbg_to_user"][:31])
        _assert_state_equal(st, c, cell.state(), f"sched {sched} refresh {refresh}")
    b.close()
    b = rs.BatchScheduler(sc, R, G, n_cells, sched=sched, jit=jit, cqi_refresh=refresh)
    b.seed(seeds)
    b.upload_cqi_epochs(grids)
    with pytest.raises(rs.RadioSaberError, match="past the last CQI epoch"):
        b.run(n_ttis)
        b.state()
    b.close()


@pytest.mark.gpu
@pytest.mark.parametrize("sched,ues,R,G,refresh,threads", [
    (8, [25] * 20, 25, 4, 1, 0),      # two 16-byte words per fetching thread
    (8, [25] * 20, 64, 8, 1, 0),      # five
    (7, [25] * 20, 25, 4, 1, 0),      # NVS: no early scan into a refresh, the grid arrives beside the link adaptation
    (1, [50] * 20, 25, 4, 1, 0),      # per-flow PF on lanes (kPf1) reading a grid that was fetched ahead
    (1, [25] * 20, 25, 4, 2, 256),    # every other TTI, four waves
    (103, [25] * 20, 25, 4, 1, 0),
    (101, [12, 30, 9], 33, 3, 1, 128),
    (9, [25] * 20, 25, 4, 1, 0),      # MaximizeCell: the straight copy at the top of the TTI (no fetch ahead)
])
def test_next_cqi_grid_fetched_during_the_serial_phase(rs, oracle, sched, ues, R, G, refresh, threads):
    """Round 4: the device-resident epoch grids are the LDS image (RBG-major [R][Upad]); when the next TTI starts a new epoch, the
    waves beside wave 0 fetch its grid during the serial phase and write it over the old one once wave 0 has read what its link
    adaptation needs (kGridAhead, rs_phase_next.inc).  Streamed-CQI mode on full-size cells, uneven launches (a launch's first TTI
    loads its grid at the top), the decisions of the first launch and the final state against the oracle."""
    sc = rs.SliceConfig(ues)
    n_cells, n_ttis = 2, 58
    grids = synth_cqi(4100 + sched + R, (n_cells, (n_ttis + refresh - 1) // refresh, sc.n_users, R), HIST)
    seeds = np.arange(n_cells, dtype=np.uint32) * 31 + 5
    b = rs.BatchScheduler(sc, R, G, n_cells, sched=sched, jit=True, cqi_refresh=refresh, threads_per_cell=threads)
    assert b.kernel_name == "rs_cell_kernel_jit"
    b.seed(seeds)
    b.upload_cqi_epochs(grids)
    np.testing.assert_array_equal(b.download_cqi_epochs(1), grids[1])  # the caller's [U][R] order back from the RBG-major store
    got = b.run_logged(21)
    for n in (1, 2, 34):
        b.run(n)
    st = b.state()
    for c in range(n_cells):
        cell = oracle.Cell(ues, R, G, sched)
        logs = cell.run_synth(grids[c], int(seeds[c]), n_ttis, refresh=refresh)
        np.testing.assert_array_equal(got["rbg_to_user"][c], logs["rbg_to_user"][:21], err_msg=f"cell {c} RBG map")
        np.testing.assert_array_equal(got["tbs_bits"][c], logs["tbs_bits"][:21], err_msg=f"cell {c} TBS")
        _assert_state_equal(st, c, cell.state(), f"sched {sched} refresh {refresh}")
    b.close()


@pytest.mark.gpu
@pytest.mark.parametrize("sched,extra", [(9, "-DRS_GRID_AHEAD_ALL"), (8, "-DRS_NO_GRID_AHEAD"), (103, "-DRS_NO_GRID_AHEAD")])
def test_fetch_ahead_switched_the_other_way(rs, oracle, sched, extra, monkeypatch):
    """The fetch-ahead switch the other way from the default (MaximizeCell with it, the others without) stays bit-exact: both
    forms are some shape's product path."""
    monkeypatch.setenv("RS_JIT_EXTRA", extra)
    monkeypatch.setenv("RS_JIT_LEAN_MIN_TTIS", "1")
    ues, R, G, refresh, n_cells, n_ttis = [25] * 20, 25, 4, 1, 2, 45
    sc = rs.SliceConfig(ues)
    grids = synth_cqi(6100 + sched, (n_cells, n_ttis, sc.n_users, R), HIST)
    seeds = np.arange(n_cells, dtype=np.uint32) + 71
    b = rs.BatchScheduler(sc, R, G, n_cells, sched=sched, jit=True, cqi_refresh=refresh)
    b.seed(seeds)
    b.upload_cqi_epochs(grids)
    got = b.run_logged(19)
    b.run(26)
    st = b.state()
    for c in range(n_cells):
        cell = oracle.Cell(ues, R, G, sched)
        logs = cell.run_synth(grids[c], int(seeds[c]), n_ttis, refresh=refresh)
        np.testing.assert_array_equal(got["rbg_to_user"][c], logs["rbg_to_user"][:19])
        _assert_state_equal(st, c, cell.state(), f"sched {sched} {extra}")
    b.close()


@pytest.mark.gpu
@pytest.mark.parametrize("sched,ues,R,G,refresh,threads", [
    (9, [25] * 20, 25, 4, 40, 0), (9, [5] * 20, 64, 8, 40, 0), (8, [25] * 20, 25, 4, 40, 0), (8, [50] * 20, 25, 4, 1, 0),
    (7, [25] * 20, 25, 4, 40, 0), (1, [25] * 20, 25, 4, 40, 0), (1, [50] * 20, 25, 4, 2, 0), (103, [12, 30, 9], 33, 3, 40, 128),
    (101, [25] * 20, 25, 4, 40, 0), (10, [5] * 20, 25, 4, 40, 0), (11, [5] * 20, 25, 4, 40, 0),
])
def test_lean_build_of_the_batch_kernel(rs, oracle, sched, ues, R, G, refresh, threads, monkeypatch):
    """Round 4: unlogged launches on epoch grids run the LEAN build of the shape-specialised kernel (the launch's unused run-time
    options -- trace rows, per-PRB twins, the decision log, error-model draws, synthetic-experiment blocks -- as compile-time
    constants; rs_api.cpp's launch(), from RS_JIT_LEAN_MIN_TTIS TTIs per launch on: 1 here).  Logged launches stay on the general
    build: the two alternate on one batch, and the final state must be the oracle's."""
    monkeypatch.setenv("RS_JIT_LEAN_MIN_TTIS", "1")
    sc = rs.SliceConfig(ues)
    n_cells, n_ttis = 2, 131
    grids = synth_cqi(5200 + sched + R, (n_cells, (n_ttis + refresh - 1) // refresh, sc.n_users, R), HIST)
    seeds = np.arange(n_cells, dtype=np.uint32) * 13 + 3
    b = rs.BatchScheduler(sc, R, G, n_cells, sched=sched, jit=True, cqi_refresh=refresh, threads_per_cell=threads)
    assert b.kernel_name == "rs_cell_kernel_jit"
    b.seed(seeds)
    b.upload_cqi_epochs(grids)
    b.prepare_launch(37)      # the lean build is compiled here (rs_batch_prepare_launch), not inside the first launch
    b.run(37)                 # lean
    got = b.run_logged(21)    # general (logs)
    for n in (1, 2, 70):      # lean again
        b.run(n)
    st = b.state()
    for c in range(n_cells):
        cell = oracle.Cell(ues, R, G, sched)
        logs = cell.run_synth(grids[c], int(seeds[c]), n_ttis, refresh=refresh)
        np.testing.assert_array_equal(got["rbg_to_user"][c], logs["rbg_to_user"][37:58], err_msg=f"cell {c} RBG map")
        np.testing.assert_array_equal(got["tbs_bits"][c], logs["tbs_bits"][37:58], err_msg=f"cell {c} TBS")
        _assert_state_equal(st, c, cell.state(), f"sched {sched} lean")
    b.close()


@pytest.mark.gpu
@pytest.mark.parametrize("sched", [9, 8])
def test_held_winners_long_run_uneven_launches(rs, oracle, sched):
    """the optimisation whose exactness rests on a numerical margin gets a driver-visible long run --
    headline shape, shape-specialised kernel, 16 cells x 6 000 TTIs cut into uneven launches; counters, PF averages and slice
    state bitwise against the oracle."""
    ues, R, G, n_cells, n_ttis = [25] * 20, 25, 4, 16, 6000
    sc = rs.SliceConfig(ues)
    seeds = np.arange(n_cells, dtype=np.uint32) * 977 + 805290992
    b = rs.BatchScheduler(sc, R, G, n_cells, sched=sched, jit=True)
    assert b.kernel_name == "rs_cell_kernel_jit"
    b.seed(seeds)
    b.synthesize_cqi(0xC0FFEE + sched, n_ttis // 40)
    done = 0
    for n in (1, 39, 41, 777, 1500, 2, 3000, 640):
        b.run(n)
        done += n
    assert done == n_ttis
    st = b.state()
    grids = [b.download_cqi_epochs(c) for c in range(n_cells)]
    ref = _oracle_final_states(oracle, ues, R, G, sched, None, lambda c: grids[c], seeds, n_ttis)
    for c in range(n_cells):
        _assert_state_equal(st, c, ref[c], f"sched {sched}")
    b.close()


@pytest.mark.gpu
@pytest.mark.parametrize("sched", [9, 8])
def test_held_winners_age_cap_ends_the_holds(rs, oracle, sched):
    """With the CQI grid refreshed only every 200 TTIs it is RS_HOLD_MAX_AGE (40 TTIs), not the refresh, that forces the full
    scans the margin mu = 2^-18 + 2 / (1 + avg_w) is sized for -- a line no other test reaches.  And the held test's other
    condition, avg_w >= 64, is exercised from both sides: two starved slices (tiny weights: served a few times per hundred TTIs)
    start with uploaded averages of 66..120 (rs_batch_write_state), so their winners pass the condition in the launch's first
    full scan and fail it at the age-cap scans of TTI 40 and 80 (0.98^40 = 0.45), while a few users of the other slices start
    right at 64 / just below.  Final state bitwise."""
    ues = [25] * 18 + [30, 20]
    w = [0.0555] * 18 + [0.0006, 0.0004]
    R, G, n_cells, n_ttis, refresh = 25, 4, 6, 2000, 200
    sc = rs.SliceConfig(ues, weight=w)
    U = sc.n_users
    seeds = np.arange(n_cells, dtype=np.uint32) * 13 + 4242
    rng = np.random.default_rng(64)
    avg0 = np.full((n_cells, U), 100000.0)
    avg0[:, 450:] = rng.uniform(66.0, 120.0, (n_cells, 50))
    avg0[:, 0:450:9] = rng.choice([63.5, 64.0, 64.5, 1.0, 200.0], (n_cells, 50))
    b = rs.BatchScheduler(sc, R, G, n_cells, sched=sched, jit=True, cqi_refresh=refresh)
    assert b.kernel_name == "rs_cell_kernel_jit"
    b.seed(seeds)
    b.synthesize_cqi(0xA6E + sched, n_ttis // refresh)
    b.write_state(avg_rate=avg0)
    with pytest.raises(rs.RadioSaberError, match="at or above 1"):
        b.write_state(avg_rate=np.zeros((n_cells, U)))
    grids = [b.download_cqi_epochs(c) for c in range(n_cells)]

    def oracle_states(n):
        def one(c):
            cell = oracle.Cell(ues, R, G, sched, weights=w)
            cell.set_avg_rate(avg0[c])
            cell.run_synth(grids[c], int(seeds[c]), n, refresh=refresh, log=False)
            return cell.state()
        with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as ex:
            return list(ex.map(one, range(n_cells)))

    b.run(85)  # full scans at TTI 0 (launch start), 40 and 80 (age cap)
    st = b.state()
    ref = oracle_states(85)
    below = sum(int((r["avg_rate"][450:] < 64.0).sum()) for r in ref)
    assert below > 20, "the starved slices' averages did not cross the held test's avg_w >= 64 boundary"
    for c in range(n_cells):
        _assert_state_equal(st, c, ref[c], f"sched {sched} after 85 TTIs")
    for n in (248, 1000, 667):
        b.run(n)
    st = b.state()
    ref = oracle_states(n_ttis)
    for c in range(n_cells):
        _assert_state_equal(st, c, ref[c], f"sched {sched}")
    b.close()


@pytest.mark.gpu
@pytest.mark.parametrize("sched", [7, 1])
@pytest.mark.parametrize("ues,weights,R,G,refresh,phy,threads", [
    ([25] * 20, None, 25, 4, 40, 0, 0),            # the sweep's shape: a new slice every TTI, winners found a TTI ahead
    ([50] * 20, None, 25, 4, 40, 1, 0),            # configs[2]: slices scanned in runs; the error model's draws on the stream
    ([5] * 20, None, 64, 8, 7, 0, 256),            # the shipped grid, a refresh every 7 TTIs (no early scan into a refresh)
    ([12, 30, 9], [0.7, 0.2, 0.1], 25, 4, 40, 0, 128),  # skewed NVS weights: the same slice twice in a row (scan at the top)
    ([40], None, 25, 4, 40, 0, 0),                 # one slice: never a different slice
    ([6, 0, 11, 3], None, 12, 2, 3, 1, 64),        # an empty slice, one wave per cell (nothing can be prepared)
])
def test_schedulers_1_and_7_prepare_the_next_tti(rs, oracle, sched, ues, weights, R, G, refresh, phy, threads):
    """Round 4: shape-specialised kernels of schedulers 1 and 7 decay every average, pick the next NVS slice and (NVS, when the
    next slice differs and the grid stays) find the next TTI's winners during the serial end of the current TTI.  Exact by
    construction; checked like everything else: decisions of the first launch and the final state after uneven launches."""
    S = len(ues)
    w = weights or [1.0 / S] * S
    sc = rs.SliceConfig(ues, weight=w)
    n_cells, n_ttis = 3, 130
    grids = synth_cqi(500 + sched + R, (n_cells, (n_ttis + refresh - 1) // refresh, sc.n_users, R), HIST)
    seeds = np.arange(n_cells, dtype=np.uint32) * 101 + 17
    b = rs.BatchScheduler(sc, R, G, n_cells, sched=sched, jit=True, cqi_refresh=refresh, phy_error_draws=bool(phy), threads_per_cell=threads)
    assert b.kernel_name == "rs_cell_kernel_jit"
    b.seed(seeds)
    b.upload_cqi_epochs(grids)
    got = b.run_logged(47)
    for n in (1, 2, 39, 41):
        b.run(n)
    st = b.state()
    for c in range(n_cells):
        cell = oracle.Cell(ues, R, G, sched, weights=w)
        logs = cell.run_synth(grids[c], int(seeds[c]), n_ttis, refresh=refresh, phy_error_draws=phy)
        np.testing.assert_array_equal(got["rbg_to_user"][c], logs["rbg_to_user"][:47], err_msg=f"cell {c} RBG map")
        np.testing.assert_array_equal(got["tbs_bits"][c], logs["tbs_bits"][:47], err_msg=f"cell {c} TBS")
        _assert_state_equal(st, c, cell.state(), f"sched {sched}")
    b.close()


@pytest.mark.gpu
@pytest.mark.parametrize("extra", ["-DRS_PF1_ALWAYS", "-DRS_PF1_ALWAYS -DRS_NO_EARLY17", "-DRS_NO_PF1_LANES"])
def test_per_flow_pf_scan_one_wave_per_rbg(rs, oracle, extra, monkeypatch):
    """Round 4: backlogged per-flow PF (sched 1) in shape-specialised batches settles every RBG on one wave over ALL users (kPf1,
    rs_phase_p3.inc) -- the default from 16 users per lane on (1 000 UEs), forced here on every shape, and switched off again so
    that the item scan stays under test at 1 000 UEs.  The first TTIs are the hard ones: every user starts from one average, so
    whole CQI classes tie and go through the exact comparison (ascending user id, strict '>')."""
    from test_gpu_parity import _check_batch
    monkeypatch.setenv("RS_JIT_EXTRA", extra)
    _check_batch(rs, oracle, 1, [50] * 20, 25, 4, n_cells=2, n_ttis=130, jit=True)            # 16 users per lane
    _check_batch(rs, oracle, 1, [25] * 20, 25, 4, n_cells=2, n_ttis=90, jit=True, phy=1)      # 25 RBGs on 8 waves: shares of 4 and 3
    _check_batch(rs, oracle, 1, [5] * 20, 64, 8, n_cells=2, n_ttis=85, jit=True, threads=256)  # 16 RBGs per wave
    _check_batch(rs, oracle, 1, [3, 0, 7, 1, 12], 12, 2, n_cells=2, n_ttis=60, jit=True, threads=64)  # one wave, ragged, an empty slice
    _check_batch(rs, oracle, 1, [37, 64, 12], 33, 3, n_cells=2, n_ttis=60, jit=True, threads=128)     # 113 users: lanes 15 ... 63 idle
    _check_batch(rs, oracle, 1, [73] * 14, 7, 4, n_cells=1, n_ttis=50, jit=True)              # 1 022 users, fewer RBGs than waves


@pytest.mark.gpu
@pytest.mark.parametrize("sched", [7, 1])
def test_schedulers_1_and_7_trace_replay_with_early_preparation(rs, oracle, traces, sched):
    """The same on the trace source: the report rule ((int)(t * 1000) - lastSent >= 40) decides whether the next TTI reads a new row."""
    ues, R, G, n_cells, n_ttis = [5] * 20, 64, 8, 2, 170
    sc = rs.SliceConfig(ues)
    U = sc.n_users
    tr = traces["cqi"]
    seeds = np.array([749913912, 805290992], np.uint32)
    skips = np.array([5000, 321], np.int64)
    maps = [traces["mapping"][1], traces["mapping"][2]]
    b = rs.BatchScheduler(sc, R, G, n_cells, sched=sched, jit=True, phy_error_draws=True)
    b.seed(seeds, skips)
    b.set_trace(tr, np.stack([m[np.arange(U) % 474] for m in maps]).astype(np.int32))
    for n in (60, 1, 109):
        b.run(n)
    st = b.state()
    for c in range(n_cells):
        cell = oracle.Cell(ues, R, G, sched)
        cell.run_trace(tr, maps[c], int(seeds[c]), int(skips[c]), n_ttis, log=False)
        _assert_state_equal(st, c, cell.state(), f"sched {sched} trace")
    b.close()


@pytest.mark.gpu
@pytest.mark.parametrize("sched", [9, 8, 1, 7, 10, 101, 103, 11])
def test_specialised_drop_in_kernel_matches_the_built_in_one(rs, oracle, sched):
    """rs_ctx_specialize: a context's own hiprtc build of the one-TTI kernel (slices, RBGs, scheduler, user CAPACITY fixed; the
    users of a call a launch argument).  The same calls through a specialised and a plain context (the plain one is what every
    other drop-in test pins against the oracle): every output array and slice_rbs_offset_ identical -- full and partial user
    lists, customised slices with head-of-line delays, per-PRB reports, the gates of schedulers 1 and 7, exact ties."""
    rng = np.random.default_rng(1000 + sched)
    shapes = [([25] * 20, 25, 4), ([5] * 20, 64, 8), ([7, 0, 12, 3], 12, 2)]
    if sched == 10:
        shapes = [([5] * 20, 25, 4), ([7, 0, 12, 3], 12, 2)]
    for ues, R, G in shapes:
        S = len(ues)
        custom = sched in (9, 8, 7)
        alpha = [int(x) for x in rng.integers(0, 2, S)] if custom else [0] * S
        beta = [int(x) for x in rng.integers(0, 2, S)] if custom else [0] * S
        sc = rs.SliceConfig(ues, algo_alpha=alpha, algo_beta=beta)
        U = sc.n_users
        u2s = np.asarray(sc.user_to_slice)
        a = rs.TtiScheduler(sc, R, G, sched=sched, jit=True)
        b = rs.TtiScheduler(sc, R, G, sched=sched)
        for it in range(8):
            if sched in (7, 11):  # NVS: the users of the served slice only
                sl = int(rng.choice([s for s in range(S) if ues[s] > 0]))
                ids = np.flatnonzero(u2s == sl)
            else:
                ids = np.arange(U) if it % 2 == 0 else np.sort(rng.choice(U, max(1, U // 2), replace=False))
            n = len(ids)
            cqi = synth_cqi(7000 + 31 * it + sched, (n, R), HIST)
            avg = rng.choice([1.0, 98000.0, 5e5], n) if it == 3 else rng.uniform(1.0, 1e6, n)
            kw = dict(user_id=ids.astype(np.int32), rand0=int(rng.integers(0, 2**31 - 1)), rand1=int(rng.integers(0, 2**31 - 1)))
            if custom and any(alpha):
                kw["hol_delay"] = rng.uniform(1e-5, 0.3, n)
                kw["prio_has_data"] = (rng.random(n) < 0.8).astype(np.uint8)
            if sched == 11:
                kw["rand_draws"] = rng.integers(0, 2**31 - 1, 300 * n).astype(np.int32)
            if it == 5 and sched == 7:
                kw["required_rbs"] = rng.integers(0, 3 * G, n).astype(np.int32)
            if it == 5 and sched == 1:
                kw["data_to_transmit"] = rng.integers(0, 4000, n).astype(np.int32)
            if it == 6:
                prb = np.repeat(cqi, G, axis=1)
                prb[:, 1::G] = np.maximum(1, prb[:, 1::G] - 1)
                ra, rb = a.schedule_tti(None, avg, cqi_prb=prb, **kw), b.schedule_tti(None, avg, cqi_prb=prb, **kw)
            else:
                ra, rb = a.schedule_tti(cqi, avg, **kw), b.schedule_tti(cqi, avg, **kw)
            for f in ("target_rbs", "quota_rbgs", "rbg_to_user", "user_nprb", "user_final_cqi", "user_mcs", "user_tbs_bits"):
                np.testing.assert_array_equal(getattr(ra, f), getattr(rb, f), err_msg=f"sched {sched} call {it}: {f}")
            if sched == 10:
                np.testing.assert_array_equal(ra.upper_rbg, rb.upper_rbg)
                np.testing.assert_array_equal(ra.upper_user, rb.upper_user)
            assert a.slice_offset.tobytes() == b.slice_offset.tobytes()
        a.close()
        b.close()


@pytest.mark.gpu
def test_bench_streamed_mode_line(rs):
    """`bench.py --cqi-refresh 1` runs the main batch in the streamed-CQI mode (epochs cycle), and the default line carries
    roofline.streamed beside the resident-design figures."""
    r = _bench(["--steps", "2", "--warmup", "1", "--cells", "16", "--ttis", "200", "--no-cpu-baseline", "--no-r64", "--no-cells1024", "--cqi-refresh", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["config"]["cqi_refresh"] == 1 and d["value"] > 0 and "streamed" not in d["roofline"]
    r = _bench(["--steps", "2", "--warmup", "1", "--cells", "16", "--ttis", "200", "--no-cpu-baseline", "--no-r64", "--no-cells1024"])
    assert r.returncode == 0, r.stderr[-2000:]
    d2 = json.loads(r.stdout.strip().splitlines()[-1])
    st = d2["roofline"]["streamed"]
    assert st["value"] > 0 and 0 < st["frac"] < 1 and st["epochs_resident"] >= 2
    assert abs(st["achieved_gbs"] - d2["roofline"]["algorithmic_bytes_per_cell_tti"] * st["value"] / 1e9) < 1e-6 * st["achieved_gbs"]
    assert d2["kernel_ms_mean_per_rank"] and len(d2["kernel_ms_mean_per_rank"]) == 1


# ---------------------------------------------------------------- two or more GPUs: the box validates itself

def _n_gpus():
    import radiosaber_amd
    return radiosaber_amd.device_count()


@pytest.mark.gpu
def test_two_gpus_bench_over_rccl_adds_up():
    """Skipped unless the box has two GPUs.  `python bench.py --gpus 2` with the default backend (nccl = RCCL over xGMI): rc 0, two
    ranks in the group, the RCCL version in the line, one mean launch duration per rank, and the all-reduced per-slice bytes
    equal the sum of two 1-GPU runs over the same GLOBAL cell ids (cells 0..63 and 64..127 as one 128-cell run: the trajectories do
    not depend on the sharding) -- VERDICT r03 next #5, ref run_backlogged.sh:6-14."""
    if _n_gpus() < 2:
        pytest.skip("needs two GPUs")
    common = ["--steps", "2", "--warmup", "1", "--ttis", "400", "--no-cpu-baseline", "--no-r64", "--no-streamed", "--no-cells1024"]
    r = _bench(["--gpus", "2", "--cells", "64"] + common)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["n_gpus"] == 2 and d["backend"] == "nccl" and d["ranks_in_group"] == 2 and d["rccl_version"]
    assert len(d["kernel_ms_mean_per_rank"]) == 2 and min(d["kernel_ms_mean_per_rank"]) > 0
    r1 = _bench(["--cells", "128"] + common)  # the same 128 global cells on one GPU
    assert r1.returncode == 0, r1.stderr[-3000:]
    d1 = json.loads(r1.stdout.strip().splitlines()[-1])
    assert d["total_slice_bytes"] == d1["total_slice_bytes"]


@pytest.mark.gpu
def test_two_gpus_cpp_host_reduces_with_rccl():
    """Skipped unless the box has two GPUs: the C++ host (tools/rs_multi_gpu.cpp: one rs_batch per device, ONE ncclAllReduce of
    uint64[S]) checks its reduced vector against the host-side sums."""
    if _n_gpus() < 2:
        pytest.skip("needs two GPUs")
    exe = ROOT / "tools" / "rs_multi_gpu"
    if not exe.exists():
        subprocess.run(["bash", str(ROOT / "tools" / "build_multi_gpu.sh")], check=True)
    r = subprocess.run([str(exe), "--gpus", "2", "--cells", "64", "--ttis", "400", "--launches", "2", "--check"], capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["n_gpus"] == 2 and d["check"] == "ok" and sum(d["slice_bytes"]) > 0
