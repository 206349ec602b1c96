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
