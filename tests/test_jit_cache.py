"""The run-time compiled kernels' cache on disk (rs_jit.cpp, ABI 10): key stability, a second process loads instead of compiling,
corrupt or foreign files are rejected and replaced, RS_JIT_CACHE=0 switches it off.  hiprtc needs no GPU."""
import json
import os
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]

SHAPE = (4, 12, 12, 2, 128, 9)  # small: ~1.5 s of hiprtc

CHILD = r"""
import json, sys, time
sys.path.insert(0, %r)
import radiosaber_amd as rs
shape = %r
t = time.time()
n = rs.jit_cache_warm(*shape, lean=%r)
print(json.dumps({"size": n, "seconds": time.time() - t, "stats": rs.jit_cache_stats(), "file": rs.jit_cache_file(*shape, lean=%r)}))
"""


def _child(cache_dir, lean=False, env_extra=None):
    # (comgr keeps a cache of its own under ~/.cache/comgr since ROCm 7: off here, so that a miss of OUR cache is a real hiprtc run)
    env = dict(os.environ, RS_JIT_CACHE_DIR=str(cache_dir), AMD_COMGR_CACHE="0")
    env.pop("RS_JIT_CACHE", None)
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, "-c", CHILD % (str(ROOT), SHAPE, lean, lean)], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr
    return json.loads(r.stdout.strip().split("\n")[-1])


def test_second_process_loads_the_code_object_instead_of_compiling(rs, tmp_path):
    first = _child(tmp_path)
    assert first["stats"] == {"hits": 0, "misses": 1, "stores": 1, "rejected": 0}
    f = Path(first["file"])
    assert f.parent == tmp_path and f.exists() and f.stat().st_size > first["size"]
    second = _child(tmp_path)
    assert second["stats"] == {"hits": 1, "misses": 0, "stores": 0, "rejected": 0}
    assert second["size"] == first["size"] and second["file"] == first["file"]
    assert second["seconds"] < 0.25 * first["seconds"], (first["seconds"], second["seconds"])
    assert not list(tmp_path.glob("*.tmp"))


def test_key_follows_the_options_and_the_variant(rs, tmp_path):
    a = _child(tmp_path)["file"]
    b = _child(tmp_path, lean=True)["file"]
    c = _child(tmp_path, env_extra={"RS_JIT_EXTRA": "-DRS_NO_HOLD"})["file"]
    d = _child(tmp_path, env_extra={"RS_JIT_SCHED_STRATEGY": "default"})["file"]
    assert len({a, b, c, d}) == 4
    assert len(list(tmp_path.glob("*.rsco"))) == 4
    # the key text (source hash, hiprtc version, every option) is stored in the file
    head = Path(a).read_bytes()[:4096]
    assert head.startswith(b"RSJC2\n") and rs.device_source_hash().encode() in head and b"-DRS_JIT_S=4\n" in head


def test_corrupt_truncated_and_foreign_files_are_compiled_again(rs, tmp_path):
    first = _child(tmp_path)
    f = Path(first["file"])
    good = f.read_bytes()
    # one flipped byte in the code
    bad = bytearray(good)
    bad[-100] ^= 0x40
    f.write_bytes(bytes(bad))
    r = _child(tmp_path)
    assert r["stats"] == {"hits": 0, "misses": 1, "stores": 1, "rejected": 1} and f.read_bytes() == good
    # truncated (a writer that died would have left a .tmp, never this -- but a full disk might)
    f.write_bytes(good[:len(good) // 2])
    assert _child(tmp_path)["stats"]["rejected"] == 1 and f.read_bytes() == good
    # another kernel's file under this name (a hash collision): the stored key text does not match
    other = Path(_child(tmp_path, lean=True)["file"]).read_bytes()
    f.write_bytes(other)
    assert _child(tmp_path)["stats"]["rejected"] == 1 and f.read_bytes() == good
    # garbage
    f.write_bytes(b"not a code object")
    assert _child(tmp_path)["stats"]["rejected"] == 1 and f.read_bytes() == good


def test_switch_and_unwritable_directory(rs, tmp_path):
    off = _child(tmp_path, env_extra={"RS_JIT_CACHE": "0"})
    assert off["stats"] == {"hits": 0, "misses": 0, "stores": 0, "rejected": 0} and not list(tmp_path.iterdir())
    blocked = tmp_path / "file_in_the_way"
    blocked.write_text("x")
    r = _child(blocked / "sub")  # mkdir fails: the kernel still compiles, nothing is stored
    assert r["size"] > 0 and r["stats"]["stores"] == 0 and r["stats"]["misses"] == 1
