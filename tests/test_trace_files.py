"""N1: the reference's CQI trace files -> the arrays rs_batch_set_trace takes.

The parser lives in the C ABI (rs_trace_*; host code, no GPU needed).  Checked against
  * small files written here in the reference's format (ue<id>.log: one line per report, space-separated
    CQI per PRB; mapping<i>.config: "<user> <trace>" lines), incl. the extraction loop's corner cases
    (short lines repeat the last value, ref: enb-mac-entity.cc:175-186);
  * when the reference tree is present (build container only): the shipped corpus against the committed
    fixture tests/golden/cqi_traces_rbg64.npz (which tools/make_trace_fixture.py made with numpy).
"""
from pathlib import Path

import numpy as np
import pytest

import radiosaber_amd as rs

REF = Path("/root/reference/cqi-traces-noise0")
GOLD = Path(__file__).parent / "golden" / "cqi_traces_rbg64.npz"


def write_log(path, rows):
    path.write_text("".join(" ".join(str(v) for v in r) + " \n" for r in rows))


def test_ue_log_rbg_and_prb_views(tmp_path):
    rng = np.random.default_rng(1)
    rbg = rng.integers(1, 16, (475, 64))
    prb = np.repeat(rbg, 8, axis=1)
    write_log(tmp_path / "ue0.log", prb)
    got, mixed = rs.read_ue_trace(tmp_path / "ue0.log")
    assert mixed == 0 and got.shape == (475, 64) and (got == rbg).all()
    got_prb, _ = rs.read_ue_trace(tmp_path / "ue0.log", per_prb=True)
    assert (got_prb == prb).all()
    # 100 PRBs / RBGs of 4 (the 25-RBG grid): the first 100 values of every line
    got25, mixed25 = rs.read_ue_trace(tmp_path / "ue0.log", nb_rbs=100, rbg_size=4)
    assert mixed25 == 0 and (got25 == prb[:, 0:100:4]).all()
    # only the first n_rows lines are read
    got10, _ = rs.read_ue_trace(tmp_path / "ue0.log", n_rows=10)
    assert (got10 == rbg[:10]).all()


def test_mixed_rbgs_are_counted(tmp_path):
    prb = np.full((3, 16), 7)
    prb[1, 5] = 9      # RBG 0 of row 1 (rbg_size 8) is not uniform
    prb[2, 8] = 3      # RBG 1 of row 2: its FIRST PRB differs -> the RBG view takes 3
    write_log(tmp_path / "ue0.log", prb)
    got, mixed = rs.read_ue_trace(tmp_path / "ue0.log", nb_rbs=16, rbg_size=8, n_rows=3)
    assert mixed == 2
    assert got.tolist() == [[7, 7], [7, 7], [7, 3]]


def test_short_and_missing_lines_repeat_the_last_value(tmp_path):
    # the reference's `int cqi` lives outside both loops and a failed `iss >> cqi` leaves it alone
    (tmp_path / "ue0.log").write_text("5 6 7\n\n9\n")
    got, _ = rs.read_ue_trace(tmp_path / "ue0.log", nb_rbs=4, rbg_size=1, n_rows=5)
    assert got.tolist() == [[5, 6, 7, 7], [7, 7, 7, 7], [9, 9, 9, 9], [9, 9, 9, 9], [9, 9, 9, 9]]
    # a token that is not a number stores 0 and fails the rest of the line (C++11 num_get)
    (tmp_path / "ue1.log").write_text("5 x 7 8\n4 4 4 4\n")
    got, _ = rs.read_ue_trace(tmp_path / "ue1.log", nb_rbs=4, rbg_size=1, n_rows=2)
    assert got.tolist() == [[5, 0, 0, 0], [4, 4, 4, 4]]


def test_errors_are_loud(tmp_path):
    with pytest.raises(rs.RadioSaberError):
        rs.read_ue_trace(tmp_path / "nope.log")
    (tmp_path / "ue0.log").write_text("5 300 7\n")
    with pytest.raises(rs.RadioSaberError, match="does not fit"):
        rs.read_ue_trace(tmp_path / "ue0.log", nb_rbs=3, rbg_size=1, n_rows=1)
    with pytest.raises(rs.RadioSaberError):
        rs.read_ue_trace(tmp_path / "ue0.log", nb_rbs=10, rbg_size=4, n_rows=1)  # 10 % 4 != 0
    (tmp_path / "empty.config").write_text("")
    with pytest.raises(rs.RadioSaberError):
        rs.read_trace_mapping(tmp_path / "empty.config")


def test_mapping_and_directory(tmp_path):
    (tmp_path / "mapping0.config").write_text("0 2\n1 0\n2 1\n3 2\n")
    m = rs.read_trace_mapping(tmp_path / "mapping0.config")
    assert m.tolist() == [2, 0, 1, 2]
    rng = np.random.default_rng(2)
    grids = rng.integers(1, 16, (3, 6, 4))
    for t in range(3):
        write_log(tmp_path / f"ue{t}.log", np.repeat(grids[t], 2, axis=1))
    got, mixed = rs.load_trace_dir(tmp_path, n_traces=3, nb_rbs=8, rbg_size=2, n_rows=6)
    assert mixed == 0 and (got == grids).all()
    # user u replays trace m[u % len(m)] (enb-mac-entity.cc:171)
    users = np.arange(10)
    assert (m[users % len(m)] == [2, 0, 1, 2, 2, 0, 1, 2, 2, 0]).all()


@pytest.mark.skipif(not REF.exists() or not GOLD.exists(), reason="reference corpus only exists in the build container")
def test_shipped_corpus_matches_the_committed_fixture():
    g = np.load(GOLD)  # cqi [158 traces][first 40 rows][64 RBGs], mapping [4 files][474 entries]
    for t in (0, 76, 157):
        got, mixed = rs.read_ue_trace(REF / f"ue{t}.log")
        assert mixed == 0
        assert (got[: g["cqi"].shape[1]] == g["cqi"][t]).all()
    all_traces, mixed = rs.load_trace_dir(REF, n_rows=g["cqi"].shape[1])
    assert mixed == 0 and (all_traces == g["cqi"]).all()
    for i in range(4):
        m = rs.read_trace_mapping(REF / f"mapping{i}.config")
        assert (m == g["mapping"][i]).all()
    m = rs.read_trace_mapping(REF / "mapping0.config")
    assert m[:3].tolist() == [76, 157, 81]  # "user 0 uses trace 76", 1 -> 157, 2 -> 81
