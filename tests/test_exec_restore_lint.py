"""tools/lint_exec_restore.py over what ships and what hiprtc builds for the shapes the benchmark and the tests use: no per-lane
instruction (spill store, split copy, ...) between a join block's entry and the `s_or_b64 exec, exec, sN` that re-enables its
lanes -- the compiler defect behind round 4's wrong SubOpt kernel (profiles/r05_onelane.md)."""
import importlib.util
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
spec = importlib.util.spec_from_file_location("lint_exec_restore", ROOT / "tools" / "lint_exec_restore.py")
lint = importlib.util.module_from_spec(spec)
spec.loader.exec_module(lint)

BAD = """
_Z6kernelv:
	s_and_saveexec_b64 s[0:1], vcc
	s_cbranch_execz .LBB0_2
; %bb.1:
	v_add_u32_e32 v1, 1, v1
.LBB0_2:
	s_waitcnt vmcnt(0)
	v_mov_b64_e32 v[4:5], v[28:29]
	scratch_store_dwordx2 off, v[30:31], off ; 8-byte Folded Spill
	v_readlane_b32 s4, v125, 3
	s_or_b64 exec, exec, s[0:1]
	s_endpgm
"""
GOOD = """
_Z6kernelv:
	s_and_saveexec_b64 s[0:1], vcc
	s_cbranch_execz .LBB0_2
; %bb.1:
	v_add_u32_e32 v1, 1, v1
	scratch_store_dwordx2 off, v[30:31], off ; 8-byte Folded Spill
.LBB0_2:
	v_readlane_b32 s4, v125, 3
	s_or_b64 exec, exec, s[0:1]
	scratch_store_dwordx2 off, v[32:33], off offset:8 ; 8-byte Folded Spill
.LBB0_3:
	v_add_u32_e32 v1, 1, v1
	s_or_b64 exec, exec, s[2:3]
	s_endpgm
"""


def test_the_lint_flags_the_pattern_and_nothing_else():
    found = lint.lint_text(BAD, "bad")
    assert len(found) == 2 and "v_mov_b64_e32 v[4:5]" in found[0] and "Folded Spill" in found[1], found
    # code inside the IF, lane-independent reads before the restore, code after it, and a block no execz jumps to: all fine
    assert lint.lint_text(GOOD, "good") == []


def test_library_kernels_are_clean(rs):
    from radiosaber_amd import build
    text = lint.disassemble(lint.code_of_library(build.LIB))
    assert text.count("rs_cell_kernel") > 25  # every instantiation is in there
    assert lint.lint_text(text, "libradiosaber_hip.so") == []


SHAPES = [  # S, U, R, G, threads, sched, lean, streamed: the bench's kernels and one per scheduler family
    (20, 500, 25, 4, 512, 9, False, False), (20, 500, 25, 4, 512, 9, True, False), (20, 500, 25, 4, 512, 9, True, True),
    (20, 500, 64, 8, 512, 9, True, False), (20, 1000, 25, 4, 512, 9, False, False),
    (20, 500, 25, 4, 512, 8, True, False), (20, 500, 25, 4, 512, 7, True, False), (20, 500, 25, 4, 512, 1, True, False),
    (20, 500, 25, 4, 512, 101, False, False), (20, 500, 25, 4, 512, 103, False, False), (20, 500, 25, 4, 512, 10, False, False),
    (20, 500, 25, 4, 512, 11, False, False),
]


def test_run_time_compiled_kernels_are_clean(rs, tmp_path, monkeypatch):
    monkeypatch.setenv("RS_JIT_CACHE_DIR", str(tmp_path))
    monkeypatch.delenv("RS_JIT_CACHE", raising=False)
    for (S, U, R, G, nt, sched, lean, streamed) in SHAPES:
        rs.jit_cache_warm(S, U, R, G, nt, sched, lean=lean, streamed=streamed)
    files = sorted(tmp_path.glob("*.rsco"))
    assert len(files) == len(SHAPES)
    for f in files:
        key, code = lint.code_of_cache_file(f)
        assert "-DRS_JIT_S=20" in key
        found = lint.lint_text(lint.disassemble(code), f.name)
        assert found == [], (key, found)
