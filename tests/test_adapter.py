"""The C++ adapter with the reference's method names (include/radiosaber_scheduler.hpp)."""
import json
import subprocess
from pathlib import Path

import numpy as np
import pytest

from conftest import GOLDEN

ROOT = Path(__file__).resolve().parents[1]


def _build(tmp_path, rs):
    exe = tmp_path / "adapter_check"
    lib_dir = ROOT / "radiosaber_amd"
    subprocess.run(["g++", "-O1", "-std=c++17", "-ffp-contract=off", "-o", str(exe),
                    str(ROOT / "tests" / "csrc" / "adapter_check.cpp"), f"-L{lib_dir}", "-lradiosaber_hip",
                    f"-Wl,-rpath,{lib_dir}", "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    return exe


def test_adapter_compiles_and_links_against_the_c_abi(tmp_path, rs):
    assert _build(tmp_path, rs).exists()


@pytest.mark.gpu
def test_adapter_replays_the_reference_run(tmp_path, rs, traces):
    """Drop-in mode, one rs_schedule_tti per TTI, real libc rand(): SURVEY.md Appendix A values."""
    ka = json.loads((GOLDEN / "appendix_a.json").read_text())
    exe = _build(tmp_path, rs)
    per_user = traces["cqi"][traces["mapping"][0][np.arange(100) % 474]]  # [100][40][64]
    tf = tmp_path / "trace.bin"
    tf.write_bytes(np.ascontiguousarray(per_user, np.uint8).tobytes())
    out = subprocess.run([str(exe), str(tf), "40", str(ka["config"]["rand_skip"]), "200"], capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = out.stdout.strip().split("\n")
    assert "first 1 final_cqi 15 rbgs 57" in lines
    assert "first 4 final_cqi 15 rbgs 14 32" in lines
    assert "first 47 final_cqi 8 rbgs 42 45" in lines
    assert "served 34 quota16 37 4" in lines
    for u, (cb, cr) in ka["after_200_ttis"]["cumu"].items():
        assert f"cumu {u} {cb} {cr}" in lines
    assert "ts 300" in lines  # 100 idle TTIs + 200 scheduled ones
    # the reference's own output lines, written by the adapter's log streams
    assert any(l.startswith("OUT slice_id, target_rbs, quota_rbgs: (0, 25, 3) ") and "(9, 25, 6) " in l and "(16, 37, 4) " in l
               and l.endswith("(19, 25, 3) ") for l in lines)
    assert "OUT 100" in lines
    assert "OUT User(1) allocated RBGS: 57(15) final_cqi: 15" in lines
    assert "OUT User(7) allocated RBGS: 17(15) 18(15) 29(15) final_cqi: 15" in lines
    assert "OUT User(45) allocated RBGS: 46(7) 51(10) 60(7) final_cqi: 7" in lines
    assert "OUT User(47) allocated RBGS: 42(11) 45(8) final_cqi: 8" in lines
    assert "ERR 100 app: 1 cumu_bytes: 749 cumu_rbs: 8 hol_delay: 0 user: 1 slice: 0" in lines
    assert "ERR 100 app: 7 cumu_bytes: 2196 cumu_rbs: 24 hol_delay: 0 user: 7 slice: 1" in lines
    assert "ERR 299 app: 5 cumu_bytes: 110838 cumu_rbs: 1336 hol_delay: 0 user: 5 slice: 1" in lines


@pytest.mark.gpu
@pytest.mark.parametrize("sched", [7, 8, 1, 10, 11])
def test_adapter_runs_every_cli_scheduler(tmp_path, rs, oracle, traces, sched):
    """The same simulator-style loop (host-side slice pick and EWMA, libc rand() shared with the error model, the 300 x n
    draws of scheduler 11 taken from libc by the adapter) for the other CLI schedulers: the adapter's cumulative counters
    after 120 TTIs equal the oracle's run on the same traces."""
    ka = json.loads((GOLDEN / "appendix_a.json").read_text())
    exe = _build(tmp_path, rs)
    per_user = traces["cqi"][traces["mapping"][0][np.arange(100) % 474]]
    tf = tmp_path / "trace.bin"
    tf.write_bytes(np.ascontiguousarray(per_user, np.uint8).tobytes())
    skip = ka["config"]["rand_skip"]
    out = subprocess.run([str(exe), str(tf), "40", str(skip), "120", str(sched)], capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    got = np.array([[int(x) for x in l.split()[1:]] for l in out.stdout.split("\n") if l.startswith("ALL ")])
    assert got.shape == (100, 3)
    cell = oracle.Cell(ka["config"]["ues_per_slice"], 64, 8, sched, weights=[ka["config"]["weight"]] * 20)
    cell.run_trace(traces["cqi"], traces["mapping"][0], ka["config"]["seed"], skip, 120)
    st = cell.state()
    np.testing.assert_array_equal(got[:, 1], st["cum_bytes"])
    np.testing.assert_array_equal(got[:, 2], st["cum_rbs"])


def _build_two(tmp_path):
    exe = tmp_path / "adapter_two_bearers"
    lib_dir = ROOT / "radiosaber_amd"
    subprocess.run(["g++", "-O1", "-std=c++17", "-ffp-contract=off", "-o", str(exe),
                    str(ROOT / "tests" / "csrc" / "adapter_two_bearers.cpp"), f"-L{lib_dir}", "-lradiosaber_hip",
                    f"-Wl,-rpath,{lib_dir}", "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    return exe


def test_two_bearer_adapter_compiles(tmp_path, rs):
    assert _build_two(tmp_path).exists()


@pytest.mark.gpu
@pytest.mark.parametrize("sched", [9, 8])
def test_adapter_customised_slices_with_two_bearers(tmp_path, rs, oracle, sched):
    """N3 in the drop-in mode: two bearers per user (MAX_BEARERS = 2), customised slices (alpha = 1, beta = 0 / 1).
    The adapter's SelectFlowsToSchedule / metric inputs / DoStopSchedule split against a replay of the same script with
    the oracle's RBsAllocation (which sums the bearers' averages in the reference's order, (1 + a0) + a1) and plain
    Python bookkeeping of the reference's lines (downlink-transport-scheduler.cpp:105-150, :170-221, :677-713;
    packet-scheduler.cpp:305-335; radio-bearer.cpp:139-164)."""
    n_ttis = 60
    exe = _build_two(tmp_path)
    out = subprocess.run([str(exe), str(sched), str(n_ttis)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = out.stdout.strip().split("\n")
    ues, alpha, beta, psi = [3, 4, 3, 2], [0, 1, 1, 0], [0, 0, 1, 0], [1, 1, 1, 0]
    U, R, G = 12, 25, 4
    u2s = np.repeat(np.arange(4), ues)
    cell = oracle.Cell(ues, R, G, sched, weights=[0.3, 0.3, 0.2, 0.2], psi=psi, alpha=alpha, beta=beta)
    rng = oracle.Rng(1234)
    second = [0, 3, 4, 5, 6, 7, 8, 9]
    B = {}  # (user, prio) -> bearer record
    for u in range(U):
        B[u, 0] = dict(avg=100000.0, tx=0, last=0.1, cb=0, cr=0, has=True, hol=0.0, q=-1, app=u)
    for u in second:
        B[u, 1] = dict(avg=100000.0, tx=0, last=0.1, cb=0, cr=0, has=True, hol=0.0, q=-1, app=100 + u)
    t = 0.1
    err = []
    ts = 0
    cqi = np.zeros((U, R), np.uint8)
    for n in range(n_ttis):
        if n % 10 == 0:
            for u in range(U):
                for r in range(R):
                    x = (u * 131 + r * 37 + (n // 10) * 101) & 0xffffffff
                    cqi[u, r] = 1 + (((x * 2654435761) & 0xffffffff) >> 7) % 15
            cell.set_cqi(cqi)
        for u in second:
            b = B[u, 1]
            b["q"] = 0 if (u * 7 + n * 13) % 5 == 0 else 200 + (u * 31 + n * 17) % 1500
            b["has"] = b["q"] > 0
            b["hol"] = 0.001 * (1 + (u * 5 + n * 3) % 40)
        B[4, 0]["has"] = (n % 3 != 0) or not B[4, 1]["has"]
        # UpdateAverageTransmissionRate
        for b in B.values():
            if t == b["last"]:
                continue
            rate = (b["tx"] * 8) / (t - b["last"])
            b["avg"] = ((1 - 0.02) * b["avg"]) + (0.02 * rate)
            if b["avg"] < 1:
                b["avg"] = 1
            b["tx"] = 0
            b["last"] = t
        # SelectFlowsToSchedule + InsertFlowToUser
        sp = [0] * 4
        data = {}
        for u in range(U):
            for pr in (0, 1):
                b = B.get((u, pr))
                if b is None or not b["has"]:
                    continue
                sp[u2s[u]] = max(sp[u2s[u]], pr)
                data.setdefault(u, [-1, -1])[pr] = 100000000 if b["q"] < 0 else b["q"]
        assert sorted(data) == list(range(U))  # the script keeps every user in the record
        avg1, avg2, hol, prio = np.zeros(U), np.full(U, -1.0), np.zeros(U), np.zeros(U, np.uint8)
        for u in range(U):
            present = [pr for pr in (0, 1) if data[u][pr] >= 0]
            avg1[u] = B[u, present[0]]["avg"]
            if len(present) == 2:
                avg2[u] = B[u, 1]["avg"]
            pb = B.get((u, sp[u2s[u]]))
            hol[u] = pb["hol"] if pb else 0.0
            prio[u] = 1 if data[u][sp[u2s[u]]] > 0 else 0
        cell.set_second_bearer_avg(avg2)
        cell.set_queue_state(hol, prio)
        o = cell.new_out()
        r0, r1 = rng.rand(), rng.rand()
        assert cell.allocate(avg1, r0, r1, o) == 0
        want = " ".join(f"{u}:{o.user_nprb[u]}:{o.user_final_cqi[u]}:{o.user_tbs_bits[u]}" for u in range(U) if o.user_nprb[u])
        assert lines[n] == (f"T {n} " + want).rstrip(), n
        # DoStopSchedule
        for u in range(U):
            if not o.user_nprb[u]:
                continue
            avail = int(o.user_tbs_bits[u]) // 8
            for pr in (1, 0):
                if avail <= 0:
                    break
                if data[u][pr] > 0:
                    sent = min(avail, data[u][pr])
                    avail -= sent
                    b = B[u, pr]
                    b["tx"] += sent
                    b["cb"] += sent
                    b["cr"] += int(o.user_nprb[u])
                    err.append(f"ERR {ts} app: {b['app']} cumu_bytes: {b['cb']} cumu_rbs: {b['cr']} hol_delay: {b['hol']:g} "
                               f"user: {u} slice: {u2s[u]}")
        ts += 1
        t += 0.001
    got_b = [l for l in lines if l.startswith("B ")]
    want_b = [f"B {u} {pr} {b['cb']} {b['cr']} {float(b['avg']).hex()}" for (u, pr), b in sorted(B.items())]
    assert [l.split()[:5] for l in got_b] == [l.split()[:5] for l in want_b]
    assert [float.fromhex(l.split()[5]) for l in got_b] == [b["avg"] for _, b in sorted(B.items())]
    got_err = [l for l in lines if l.startswith("ERR ")]
    assert got_err == err
    # the script did exercise what it is for: a bearer pair that shares a grant, and a record with only the priority-1 bearer
    assert any(B[u, 1]["cb"] > 0 and B[u, 0]["cb"] > 0 for u in second)


def test_two_bearer_average_goes_through_the_abi_exactly():
    """rs_tti_in.avg_rate is one number per user and the kernel forms 1 + avg_rate; the reference forms (1 + a0) + a1 for a
    user with two bearers.  The adapter passes K - 1 with K = (1 + a0) + a1: for averages >= 1 (the EWMA clamps there) K - 1
    is exact and 1 + (K - 1) == K, whereas the plain sum a0 + a1 is off by an ulp now and then."""
    rng = np.random.default_rng(5)
    a0 = np.concatenate([rng.uniform(1, 4, 200000), 10 ** rng.uniform(0, 9, 200000), [1.0, 1.0, 2.0 ** 52]])
    a1 = np.concatenate([rng.uniform(1, 4, 200000), 10 ** rng.uniform(0, 9, 200000), [1.0, 2.0 ** 52, 1.0]])
    K = (1 + a0) + a1
    assert ((1 + (K - 1)) == K).all()
    assert ((1 + (a0 + a1)) != K).any()  # why the sum of the averages is not what goes in
