"""Round 3: bench.py starting its own ranks, the headline's guards against build switches, arbitrary doubles at the
drop-in entry point, and the CQI-source bookkeeping the advisor flagged."""
import json
import math
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

from conftest import synth_cqi

ROOT = Path(__file__).resolve().parents[1]
HIST = [152600, 56656, 270880, 2088792, 3509504, 1595568, 4145392, 5295816, 1903424, 6890232, 4770864, 2842552, 3579624, 96000, 1227696]


def _bench(args, env_extra=None, timeout=900):
    env = dict(os.environ)
    for k in ("RS_JIT_EXTRA", "RS_JIT", "WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "RS_BENCH_BACKEND"):
        env.pop(k, None)  # (an earlier test of the session may have left a one-rank rendezvous in os.environ)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, str(ROOT / "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)


# ---------------------------------------------------------------- CPU side: argument handling of bench.py

def test_bench_refuses_build_switches_in_the_environment():
    """an inflated headline must not be one `export` away."""
    r = _bench(["--steps", "1"], {"RS_JIT_EXTRA": "-DRS_NO_SPEC"})
    assert r.returncode != 0 and "RS_JIT_EXTRA" in r.stderr and r.stdout.strip() == ""
    r = _bench(["--steps", "1"], {"RS_JIT": "0"})
    assert r.returncode != 0 and "RS_JIT" in r.stderr


def test_product_kernel_source_has_no_wrong_result_switches():
    src = (ROOT / "radiosaber_amd" / "csrc" / "rs_kernels.hip").read_text()
    assert "RS_EXP_" not in src
    jit = (ROOT / "radiosaber_amd" / "csrc" / "rs_jit.cpp").read_text()
    assert "%.60s" not in jit  # the whole RS_JIT_EXTRA string is part of the cache key


def test_device_source_hash_is_stable_and_hex(rs):
    h = rs.device_source_hash()
    assert len(h) == 16 and int(h, 16) >= 0 and h == rs.device_source_hash()


# ---------------------------------------------------------------- GPU side

@pytest.mark.gpu
def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it: two ranks (gloo here: the box has one GPU), rc 0, exactly one JSON
    line, n_gpus 2, and the line says which backend reduced and how many ranks it saw."""
    common = ["--steps", "2", "--warmup", "1", "--ttis", "200", "--no-cpu-baseline", "--no-r64", "--no-streamed", "--no-cells1024"]
    r = _bench(["--gpus", "2", "--cells", "16"] + common, {"RS_BENCH_BACKEND": "gloo"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["backend"] == "gloo" and d["ranks_in_group"] == 2 and d["steps"] == 2
    assert d["value"] > 0 and d["total_slice_bytes"] > 0 and d["jit_extra"] == "" and d["source_hash"]
    # Both shards contributed, and EXACTLY what they should have: rank 0 owns global cells [0, 16), rank 1
    # [16, 32); seeds and CQI grids are functions of the global cell id, so one rank running cells [0, 32) grants the same bytes --
    # the reduced integer must be equal, not merely "more than one and a half shards".
    r1 = _bench(["--cells", "32"] + common)
    assert r1.returncode == 0, r1.stderr[-2000:]
    d1 = json.loads(r1.stdout.strip().splitlines()[-1])
    assert d1["n_gpus"] == 1 and d1["backend"] is None and d1["ranks_in_group"] == 1
    assert d["total_slice_bytes"] == d1["total_slice_bytes"], (d["total_slice_bytes"], d1["total_slice_bytes"])
    # ... and one shard alone is a different (smaller) number
    r16 = _bench(["--cells", "16"] + common)
    assert r16.returncode == 0, r16.stderr[-2000:]
    assert 0 < json.loads(r16.stdout.strip().splitlines()[-1])["total_slice_bytes"] < d["total_slice_bytes"]


@pytest.mark.gpu
def test_bench_refuses_more_rccl_ranks_than_gpus():
    import radiosaber_amd
    if radiosaber_amd.device_count() >= 2:
        pytest.skip("box has several GPUs")
    r = _bench(["--gpus", "2", "--steps", "1", "--cells", "8", "--ttis", "40", "--no-cpu-baseline", "--no-r64", "--no-cells1024"])
    assert r.returncode != 0 and "one GPU per rank" in r.stderr and r.stdout.strip() == ""


@pytest.mark.gpu
def test_bench_variant_runs_only_when_allowed():
    r = _bench(["--steps", "1", "--warmup", "0", "--cells", "8", "--ttis", "80", "--no-cpu-baseline", "--no-r64", "--no-cells1024", "--allow-variant"],
               {"RS_JIT_EXTRA": "-DRS_NO_SPEC"})
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads(r.stdout.strip().splitlines()[-1])["jit_extra"] == "-DRS_NO_SPEC"


@pytest.mark.gpu
@pytest.mark.parametrize("sched", [9, 8, 1, 103, 101, 10])
def test_drop_in_accepts_any_double(rs, oracle, sched):
    """The reference's metric takes whatever double the bearer holds (downlink-transport-scheduler.cpp:677-713): averages of 1,
    1e12, 1e300, infinity, sub-unit and negative values, head-of-line delays of 1e-5, 0 and 1e6 -- outside the FP32 ranking's
    safe range the device compares every user exactly."""
    ues = [6, 5, 7, 4, 8, 6]
    custom = sched in (9, 8)
    alpha = [0, 1, 1, 1, 1, 0] if custom else [0] * 6
    beta = [0, 0, 1, 1, 1, 0] if custom else [0] * 6
    eps = [1, 1, 1, 1, 0, 1]
    psi = [1, 1, 1, 0, 1, 0]
    S, R, G = len(ues), 25, 4
    w = [1.0 / S] * S
    sc = rs.SliceConfig(ues, weight=w, algo_alpha=alpha, algo_beta=beta, algo_epsilon=eps, algo_psi=psi)
    U = sc.n_users
    ts = rs.TtiScheduler(sc, R, G, sched=sched)
    cell = oracle.Cell(ues, R, G, sched, weights=w, epsilon=eps, psi=psi, alpha=alpha, beta=beta)
    rng = np.random.default_rng(5)
    pools = [np.array([1.0, 1e12, 1e300]), np.array([1.0, 3.4e41, 1e38, 1e300]), np.array([1e300]), np.array([np.inf, 1e5]),
             np.array([0.25, 1.0, 1e-300]), np.array([-0.5, -1.0, -3.0, 2.0]), np.array([1.0, 98000.0, 1e12])]
    hols = [np.array([1e-5, 0.0, 1e6]), np.array([1e-300, 1e300, 1.0]), np.array([0.0])]
    for it in range(3 * len(pools)):
        cqi = synth_cqi(900 + it, (U, R), HIST)
        avg = rng.choice(pools[it % len(pools)], U)
        hol = rng.choice(hols[it % len(hols)], U)
        prio = (rng.random(U) < 0.8).astype(np.uint8)
        r0, r1 = int(rng.integers(0, 2**31 - 1)), int(rng.integers(0, 2**31 - 1))
        cell.set_cqi(cqi)
        if custom:
            cell.set_queue_state(hol, prio)
        out = cell.new_out()
        assert cell.allocate(avg, r0, r1, out) == 0
        kw = dict(hol_delay=hol, prio_has_data=prio) if custom else {}
        res = ts.schedule_tti(cqi, avg, r0, r1, **kw)
        np.testing.assert_array_equal(res.rbg_to_user, out.rbg_to_user, err_msg=f"it {it}")
        np.testing.assert_array_equal(res.quota_rbgs, out.quota_rbgs)
        np.testing.assert_array_equal(res.user_nprb, out.user_nprb)
        np.testing.assert_array_equal(res.user_tbs_bits, out.user_tbs_bits)
    ts.close()


@pytest.mark.gpu
def test_drop_in_nvs_accepts_any_double(rs):
    """sched 7 (one slice per call): first maximum of the slice metric from lowest(), whatever the doubles are."""
    ues, R, G = [8] * 4, 25, 4
    sc = rs.SliceConfig(ues, weight=[0.25] * 4)
    ts = rs.TtiScheduler(sc, R, G, sched=7)
    kb = rs.link_tables()["kbps"]
    rng = np.random.default_rng(8)
    for it in range(8):
        sl = it % 4
        ids = np.arange(sl * 8, sl * 8 + 8)
        cqi = synth_cqi(40 + it, (8, R), HIST)
        avg = rng.choice(np.array([1.0, 1e12, 1e300, 5e40]), 8)
        res = ts.schedule_tti(cqi, avg, user_id=ids)
        met = kb[cqi] / ((1 + avg) / 1000.0)[:, None]
        np.testing.assert_array_equal(res.rbg_to_user, ids[np.argmax(met, axis=0)], err_msg=f"it {it}")
    ts.close()


@pytest.mark.gpu
def test_synthesize_after_per_prb_upload_drops_the_stale_twin(rs):
    """upload_cqi_epochs_prb(A epochs) then synthesize_cqi(B > A epochs) left the per-PRB twin of the OLD grids in
    place -- link adaptation read stale CQIs (and past the allocation).  The batch must behave like a fresh one."""
    ues, R, G, n_cells = [4] * 5, 12, 2, 3
    sc = rs.SliceConfig(ues, weight=[0.2] * 5)
    U = sc.n_users
    seeds = np.arange(n_cells, dtype=np.uint32) + 3

    def run(pre_upload):
        b = rs.BatchScheduler(sc, R, G, n_cells, sched=9)
        b.seed(seeds)
        if pre_upload:
            prb = np.full((n_cells, 1, U, R * G), 3, np.uint8)
            prb[..., 1::2] = 15  # differs inside every RBG: a stale twin would change every final CQI
            b.upload_cqi_epochs_prb(prb)
        b.synthesize_cqi(77, 4)
        got = b.run_logged(160)
        st = b.state()
        b.close()
        return got, st

    g0, s0 = run(False)
    g1, s1 = run(True)
    np.testing.assert_array_equal(g0["rbg_to_user"], g1["rbg_to_user"])
    np.testing.assert_array_equal(g0["tbs_bits"], g1["tbs_bits"])
    np.testing.assert_array_equal(s0["cum_bytes"], s1["cum_bytes"])


@pytest.mark.gpu
@pytest.mark.parametrize("sched", [1, 7, 9])
@pytest.mark.parametrize("jit", [False, True])
def test_queue_model_on_per_prb_sources(rs, oracle, sched, jit):
    """with finite queues the gates of schedulers 1 and 7 (the satisfied-flow break: the transport block of the
    flow's PRBs so far, downlink-packet-scheduler.cpp:245-264; m_requiredRBs: the wideband CQI over every PRB of the band,
    packet-scheduler.cpp:319-334) must read the per-PRB reports when the batch has them, like the final link adaptation."""
    from test_gpu_queues import _random_bursts
    ues, R, G, n_cells, n_ttis = [4, 5, 3, 6], 25, 4, 2, 120
    kinds_of = ["B-", "Q-", "QQ", "Q-"]
    alpha = [0, 1, 1, 0] if sched != 1 else [0] * 4
    beta = [0, 0, 1, 0] if sched != 1 else [0] * 4
    sc = rs.SliceConfig(ues, algo_alpha=alpha, algo_beta=beta)
    U, u2s = sc.n_users, sc.user_to_slice
    code = {"B": rs.BEARER_BACKLOG, "Q": rs.BEARER_QUEUE, "-": rs.BEARER_NONE}
    kinds = np.array([[code[kinds_of[u2s[u]][0]], code[kinds_of[u2s[u]][1]]] for u in range(U)], np.uint8)
    rng = np.random.default_rng(61 + sched)
    bursts = {(c, u, k): _random_bursts(rng, n_ttis, 8, 1500) for c in range(n_cells) for u in range(U) for k in range(2)
              if kinds[u, k] == rs.BEARER_QUEUE}
    base = synth_cqi(70 + sched, (n_cells, (n_ttis + 39) // 40, U, R), HIST).astype(np.int16)
    prb = np.clip(np.repeat(base, G, axis=3) + rng.integers(-3, 4, base.shape[:3] + (R * G,)), 1, 15).astype(np.uint8)
    seeds = np.array([9, 10], np.uint32)
    b = rs.BatchScheduler(sc, R, G, n_cells, sched=sched, jit=jit)
    b.set_bearers(kinds)
    b.set_arrivals(bursts)
    b.seed(seeds)
    b.upload_cqi_epochs_prb(prb)
    got = b.run_logged(n_ttis)
    bst = b.bearer_state()
    b.close()
    differs = 0
    for c in range(n_cells):
        cell = oracle.Cell(ues, R, G, sched, alpha=alpha, beta=beta)
        cell.enable_queues(kinds)
        for (cc, u, k), (t, nf, la) in bursts.items():
            if cc == c:
                cell.set_arrivals(u, k, t, nf, la)
        logs = cell.run_synth_queues(prb[c], int(seeds[c]), n_ttis, per_prb=True)
        ob = cell.bearer_state()
        np.testing.assert_array_equal(got["rbg_to_user"][c], logs["rbg_to_user"], err_msg=f"cell {c} RBG map")
        np.testing.assert_array_equal(got["tbs_bits"][c], logs["tbs_bits"], err_msg=f"cell {c} TBS")
        for key in ("cum_bytes", "cum_rbs", "queue_bytes", "queue_packets"):
            np.testing.assert_array_equal(bst[key][c], ob[key], err_msg=f"cell {c} {key}")
        # the per-PRB reports matter: the same run on each RBG's first PRB alone ends elsewhere
        cell2 = oracle.Cell(ues, R, G, sched, alpha=alpha, beta=beta)
        cell2.enable_queues(kinds)
        for (cc, u, k), (t, nf, la) in bursts.items():
            if cc == c:
                cell2.set_arrivals(u, k, t, nf, la)
        logs2 = cell2.run_synth_queues(prb[c][..., ::G], int(seeds[c]), n_ttis)
        differs += int((logs2["tbs_bits"] != logs["tbs_bits"]).any())
    assert differs > 0, "per-PRB and per-RBG runs agree everywhere: the test does not exercise the per-PRB reads"


# ---------------------------------------------------------------- the C++ multi-GPU host (RCCL called directly)

def _build_multi_gpu():
    r = subprocess.run([str(ROOT / "tools" / "build_multi_gpu.sh")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    exe = ROOT / "tools" / "rs_multi_gpu"
    assert exe.exists()
    return exe


def test_cpp_multi_gpu_host_builds_and_fails_loudly_without_a_gpu(rs):
    """tools/rs_multi_gpu.cpp (one process, one rs_batch per GPU, ncclAllReduce over ncclCommInitAll) compiles and links against
    the C ABI library and RCCL; without a HIP device it must refuse, not fall back."""
    exe = _build_multi_gpu()
    if rs.device_count() > 0:
        pytest.skip("a GPU is visible: the run itself is the gpu test")
    r = subprocess.run([str(exe), "--gpus", "1"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and "no HIP device" in r.stderr


@pytest.mark.gpu
def test_cpp_multi_gpu_host_reduces_with_rccl_and_matches_the_python_path(rs):
    """The C++ host on every visible GPU: the RCCL all-reduce of the device-resident per-slice vectors equals the host-side sums
    (--check), and on one GPU the per-slice bytes equal those of the Python path bench.py uses (same sharding rule: seeds and CQI
    grids keyed on the global cell id)."""
    from radiosaber_amd import sharding
    exe = _build_multi_gpu()
    cells, ttis, launches = 8, 200, 2
    r = subprocess.run([str(exe), "--gpus", "1", "--cells", str(cells), "--ttis", str(ttis), "--launches", str(launches), "--check"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["check"] == "ok" and d["n_gpus"] == 1 and d["rccl_version"] > 0
    sc = rs.SliceConfig([25] * 20, weight=[0.05] * 20)
    b = rs.BatchScheduler(sc, 25, 4, cells, sched=rs.RS_SCHED_MAXCELL, jit=True)
    b.seed(sharding.seeds_for_cells(sharding.cell_ids_for_rank(0, 1, cells)))
    b.synthesize_cqi(0x5AB3, (launches * ttis + 40 + 39) // 40, first_cell=0)
    b.run(40)
    for _ in range(launches):
        b.run(ttis)
    want = b.slice_bytes().astype(np.uint64)
    b.close()
    np.testing.assert_array_equal(np.array(d["slice_bytes"], np.uint64), want)
    # every visible GPU (the driver's 8-GPU node; one here)
    r = subprocess.run([str(exe), "--cells", "4", "--ttis", "120", "--launches", "1", "--check"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert json.loads(r.stdout.strip().splitlines()[-1])["n_gpus"] == rs.device_count()


# ---------------------------------------------------------------- held winners

@pytest.mark.gpu
@pytest.mark.parametrize("sched", [9, 8, 103])
@pytest.mark.parametrize("case", ["headline", "windows-of-48-and-64", "too-long-for-holding", "psi-0-and-ties", "tiny-and-empty"])
def test_held_winners_stay_bit_exact(rs, oracle, sched, case):
    """Shape-specialised kernels of up to 32 RBGs scan only the (slice, RBG) items whose winner can have changed: runs long enough
    for winners to be held, served, listed and scanned again (CQI epochs of 40 TTIs, several of them), over slice windows of 32,
    48 and 64 users (one or two groups of 8 users per lane), slices too long for the scheme (it switches itself off), psi = 0
    slices (winner independent of the averages, exact ties between equal CQIs) and cells with empty / one-user slices."""
    from test_gpu_parity import _check_batch
    if case == "headline":
        _check_batch(rs, oracle, sched, [25] * 20, 25, 4, n_cells=2, n_ttis=250, jit=True, seed=21)
    elif case == "windows-of-48-and-64":
        _check_batch(rs, oracle, sched, [40, 33, 56, 7, 49, 50], 25, 4, n_cells=2, n_ttis=170, jit=True, seed=22)
    elif case == "too-long-for-holding":
        _check_batch(rs, oracle, sched, [70, 12, 25], 12, 2, n_cells=2, n_ttis=130, jit=True, seed=23)
    elif case == "psi-0-and-ties":
        _check_batch(rs, oracle, sched, [25, 25, 9, 30], 25, 4, n_cells=2, n_ttis=170, jit=True, psi=[0, 1, 0, 1], eps=[1, 0, 0, 1], seed=24)
    else:
        _check_batch(rs, oracle, sched, [1, 0, 3, 1, 0, 12, 2], 17, 3, n_cells=3, n_ttis=130, jit=True, threads=64, seed=25)


@pytest.mark.gpu
def test_held_winners_across_uneven_launches(rs, oracle):
    """The held bits live in LDS: every launch starts with a full scan, whatever its length (1-TTI launches, launches that end
    inside an epoch, launches that span several)."""
    sc_ues = [25] * 20
    sc = rs.SliceConfig(sc_ues, weight=[0.05] * 20)
    R, G, n_cells = 25, 4, 2
    launches = [1, 2, 37, 41, 80, 3, 60]
    n_ttis = sum(launches)
    grids = synth_cqi(31, (n_cells, (n_ttis + 39) // 40, sc.n_users, R), HIST)
    seeds = np.array([77, 78], np.uint32)
    b = rs.BatchScheduler(sc, R, G, n_cells, sched=9, jit=True)
    b.seed(seeds)
    b.upload_cqi_epochs(grids)
    maps = np.concatenate([b.run_logged(n)["rbg_to_user"] for n in launches], axis=1)
    st = b.state()
    b.close()
    for c in range(n_cells):
        cell = oracle.Cell(sc_ues, R, G, oracle.SCHED_MAXCELL, weights=[0.05] * 20)
        logs = cell.run_synth(grids[c], int(seeds[c]), n_ttis)
        np.testing.assert_array_equal(maps[c], logs["rbg_to_user"])
        assert st["avg_rate"][c].tobytes() == cell.state()["avg_rate"].tobytes()


# ---------------------------------------------------------------- pow() with any integer exponents (drop-in mode)

@pytest.mark.gpu
@pytest.mark.parametrize("sched", [9, 8, 10, 101, 103])
@pytest.mark.parametrize("custom", [False, True])
def test_drop_in_general_integer_exponents(rs, oracle, sched, custom):
    """ref: downlink-transport-scheduler.cpp:690-693 evaluates pow(se_kbps, epsilon) / pow(avg_kbps, psi) for ANY integers
    (packet-scheduler.h:38-49).  A drop-in context takes them: the host's libm (the one the reference and the oracle call) raises
    the 16 numerators per slice at rs_create and every user's denominator per call, the device divides and compares exactly.
    Batches (PF averages updated on the device) keep exponents 0 / 1 and say so."""
    ues = [7, 9, 4, 8, 6, 5]
    alpha = [0, 1, 1, 1, 1, 0] if custom else [0] * 6
    beta = [0, 0, 1, 1, 1, 0] if custom else [0] * 6
    eps = [2, 1, 3, -1, 0, 5]
    psi = [1, 2, -2, 3, 2, 0]
    S, R, G = len(ues), 25, 4
    w = [1.0 / S] * S
    sc = rs.SliceConfig(ues, weight=w, algo_alpha=alpha, algo_beta=beta, algo_epsilon=eps, algo_psi=psi)
    U = sc.n_users
    ts = rs.TtiScheduler(sc, R, G, sched=sched)
    cell = oracle.Cell(ues, R, G, sched, weights=w, epsilon=eps, psi=psi, alpha=alpha, beta=beta)
    rng = np.random.default_rng(50 + sched)
    differs = 0
    for it in range(12):
        cqi = synth_cqi(1500 + it, (U, R), HIST)
        avg = np.exp(rng.uniform(np.log(1.0), np.log(5e7), U)) if it % 3 else rng.choice([1.0, 98000.0, 1e5], U)  # ties too
        hol = rng.uniform(1e-5, 0.2, U)
        prio = (rng.random(U) < 0.8).astype(np.uint8)
        r0, r1 = int(rng.integers(0, 2**31 - 1)), int(rng.integers(0, 2**31 - 1))
        cell.set_cqi(cqi)
        if custom:
            cell.set_queue_state(hol, prio)
        out = cell.new_out()
        assert cell.allocate(avg, r0, r1, out) == 0
        kw = dict(hol_delay=hol, prio_has_data=prio) if custom else {}
        res = ts.schedule_tti(cqi, avg, r0, r1, **kw)
        np.testing.assert_array_equal(res.rbg_to_user, out.rbg_to_user, err_msg=f"it {it}")
        np.testing.assert_array_equal(res.quota_rbgs, out.quota_rbgs)
        np.testing.assert_array_equal(res.user_nprb, out.user_nprb)
        np.testing.assert_array_equal(res.user_tbs_bits, out.user_tbs_bits)
        # the exponents matter: the plain PF metric would have chosen differently
        if sched == 9 and it == 1 and not custom:
            sc1 = rs.SliceConfig(ues, weight=w)
            t1 = rs.TtiScheduler(sc1, R, G, sched=sched)
            differs += int((t1.schedule_tti(cqi, avg, r0, r1).rbg_to_user != res.rbg_to_user).any())
            t1.close()
    ts.close()
    if sched == 9 and not custom:
        assert differs, "exponents (2, 1, 3, -1, 0, 5) / (1, 2, -2, 3, 2, 0) chose like (1, 1): the test does not exercise them"
    # a batch keeps the restriction and says why
    with pytest.raises(rs.RadioSaberError, match="0 or 1 in a batch"):
        rs.BatchScheduler(sc, R, G, 1, sched=sched)


@pytest.mark.gpu
def test_drop_in_nvs_general_exponents(rs, oracle):
    """sched 7 with exponents outside {0, 1} (downlink-nvs-scheduler.cpp:360-390: the same pow expression), with and without the
    m_requiredRBs gate."""
    ues, R, G = [8, 6, 9], 25, 4
    eps, psi = [3, -1, 2], [2, 2, -1]
    sc = rs.SliceConfig(ues, weight=[0.3, 0.3, 0.4], algo_epsilon=eps, algo_psi=psi)
    ts = rs.TtiScheduler(sc, R, G, sched=7)
    kb = np.asarray(rs.link_tables()["kbps"])
    rng = np.random.default_rng(8)
    first = np.concatenate([[0], np.cumsum(ues)])
    for it in range(9):
        s = it % 3
        n = ues[s]
        ids = np.arange(first[s], first[s] + n, dtype=np.int32)
        cqi = synth_cqi(1700 + it, (n, R), HIST)
        avg = np.exp(rng.uniform(np.log(1.0), np.log(5e7), n))
        need = rng.integers(0, 40, n).astype(np.int32) if it % 2 else None
        res = ts.schedule_tti(cqi, avg, user_id=ids, **({"required_rbs": need} if need is not None else {}))
        # libm's pow, element by element: numpy's vectorised power is a different implementation (last-bit differences)
        num = np.array([[math.pow(float(kb[c]), eps[s]) for c in row] for row in cqi])
        den = np.array([math.pow((1.0 + float(a)) / 1000.0, psi[s]) for a in avg])
        met = num / den[:, None]
        left = need.copy() if need is not None else np.full(n, 1 << 30)
        want = np.full(R, -1)
        for r in range(R):
            ok = left > 0
            if ok.any():
                k = int(np.flatnonzero(ok)[np.argmax(met[ok, r])])
                want[r] = ids[k]
                left[k] -= G
        np.testing.assert_array_equal(res.rbg_to_user, want, err_msg=f"it {it}")
    ts.close()


# ---------------------------------------------------------------- the reference's synthetic-experiment transport block

@pytest.mark.gpu
@pytest.mark.parametrize("sched", [9, 8, 7, 10, 103, 1])
@pytest.mark.parametrize("jit", [False, True])
def test_synthetic_exp_transport_block(rs, oracle, sched, jit):
    """FIRST_SYNTHETIC_EXP / SECOND_SYNTHETIC_EXP (CONFIG/global_config:57-58, off as shipped): DownlinkTransportScheduler and the
    NVS scheduler size the transport block PRB by PRB, each with the MCS of its own CQI (downlink-transport-scheduler.cpp:653-659,
    downlink-nvs-scheduler.cpp:336-342); the per-flow PF scheduler has no such branch (the flag changes nothing there).  Whole TTI
    loops (the grant feeds the PF average, so the trajectories differ from the as-shipped build's) on per-RBG and per-PRB sources."""
    ues, R, G, n_cells, n_ttis = [6] * 8, 25, 4, 2, 90
    sc = rs.SliceConfig(ues, weight=[0.125] * 8)
    U = sc.n_users
    grids = synth_cqi(60 + sched, (n_cells, (n_ttis + 39) // 40, U, R), HIST)
    seeds = np.array([5, 9], np.uint32)
    tbs = {}
    for syn in (True, False):
        b = rs.BatchScheduler(sc, R, G, n_cells, sched=sched, jit=jit, synthetic_exp=syn)
        b.seed(seeds)
        b.upload_cqi_epochs(grids)
        got = b.run_logged(n_ttis)
        st = b.state()
        b.close()
        tbs[syn] = got["tbs_bits"]
        for c in range(n_cells):
            cell = oracle.Cell(ues, R, G, sched, weights=[0.125] * 8)
            cell.set_synthetic_exp(syn)
            logs = cell.run_synth(grids[c], int(seeds[c]), n_ttis)
            np.testing.assert_array_equal(got["rbg_to_user"][c], logs["rbg_to_user"], err_msg=f"synthetic {syn} cell {c}")
            np.testing.assert_array_equal(got["tbs_bits"][c], logs["tbs_bits"])
            assert st["avg_rate"][c].tobytes() == cell.state()["avg_rate"].tobytes()
    if sched == 1:
        np.testing.assert_array_equal(tbs[True], tbs[False])  # no synthetic branch in DownlinkPacketScheduler::RBsAllocation
    else:
        assert (tbs[True] != tbs[False]).any()
    # per-PRB reports: every PRB at its own CQI
    prb = np.repeat(grids, G, axis=3).astype(np.int16)
    prb[..., 1::G] = np.clip(prb[..., 1::G] + synth_cqi(3, prb[..., 1::G].shape, HIST).astype(np.int16) % 3 - 1, 1, 15)
    prb = prb.astype(np.uint8)
    b = rs.BatchScheduler(sc, R, G, n_cells, sched=sched, jit=jit, synthetic_exp=True)
    b.seed(seeds)
    b.upload_cqi_epochs_prb(prb)
    got = b.run_logged(n_ttis)
    b.close()
    for c in range(n_cells):
        cell = oracle.Cell(ues, R, G, sched, weights=[0.125] * 8)
        cell.set_synthetic_exp(True)
        logs = cell.run_synth(prb[c], int(seeds[c]), n_ttis, per_prb=True)
        np.testing.assert_array_equal(got["rbg_to_user"][c], logs["rbg_to_user"])
        np.testing.assert_array_equal(got["tbs_bits"][c], logs["tbs_bits"])


@pytest.mark.gpu
def test_synthetic_exp_drop_in(rs, oracle):
    """the same through rs_schedule_tti (what integration/downlink-gpu-scheduler.cpp passes when the reference tree defines
    FIRST_SYNTHETIC_EXP / SECOND_SYNTHETIC_EXP)"""
    ues, R, G = [7, 5, 6], 25, 4
    sc = rs.SliceConfig(ues, weight=[0.3, 0.3, 0.4])
    U = sc.n_users
    rng = np.random.default_rng(4)
    for sched in (9, 10):
        ts = rs.TtiScheduler(sc, R, G, sched=sched, synthetic_exp=True)
        cell = oracle.Cell(ues, R, G, sched, weights=[0.3, 0.3, 0.4])
        cell.set_synthetic_exp(True)
        for it in range(6):
            cqi = synth_cqi(300 + it, (U, R), HIST)
            avg = np.exp(rng.uniform(np.log(1e3), np.log(1e7), U))
            r0, r1 = int(rng.integers(0, 2**31 - 1)), int(rng.integers(0, 2**31 - 1))
            cell.set_cqi(cqi)
            out = cell.new_out()
            assert cell.allocate(avg, r0, r1, out) == 0
            res = ts.schedule_tti(cqi, avg, r0, r1)
            np.testing.assert_array_equal(res.rbg_to_user, out.rbg_to_user)
            np.testing.assert_array_equal(res.user_tbs_bits, out.user_tbs_bits)
            np.testing.assert_array_equal(res.user_mcs, out.user_mcs)  # the PDCCH record keeps the EESM MCS
        ts.close()
