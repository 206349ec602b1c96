"""The run-time compiled kernels' cache on disk (rs_jit.cpp, ABI 10): key stability, a second process loads instead of compiling,
corrupt or foreign files are rejected and replaced, RS_JIT_CACHE=0 switches it off.  hiprtc needs no GPU."""
import json
import os
import subprocess
import sys
import time
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]

SHAPE = (4, 12, 12, 2, 128, 9)  # small: ~1.5 s of hiprtc

CHILD = r"""
import json, sys, time, os
sys.path.insert(0, %r)
if os.environ.get("RS_TEST_PREFER_SYSTEM_COMPILER") == "1":
    from radiosaber_amd import toolchain
    toolchain.prefer_system_compiler()
if os.environ.get("RS_TEST_IMPORT_TORCH_FIRST") == "1":
    import torch  # noqa: F401 -- the torch wheel bundles its own ROCm user space: whichever libhiprtc a process loads first serves it
sys.path.insert(0, %r)
import radiosaber_amd as rs
shape = %r
t = time.time()
n = rs.jit_cache_warm(*shape, lean=%r)
print(json.dumps({"size": n, "seconds": time.time() - t, "stats": rs.jit_cache_stats(), "file": rs.jit_cache_file(*shape, lean=%r),
                  "identity": rs.jit_compiler_identity()}))
"""


def _child(cache_dir, lean=False, env_extra=None):
    # (comgr keeps a cache of its own under ~/.cache/comgr since ROCm 7: off here, so that a miss of OUR cache is a real hiprtc run)
    env = dict(os.environ, RS_JIT_CACHE_DIR=str(cache_dir), AMD_COMGR_CACHE="0")
    env.pop("RS_JIT_CACHE", None)
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, "-c", CHILD % (str(ROOT), str(ROOT), SHAPE, lean, lean)], capture_output=True, text=True, env=env, timeout=600)
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


# ---- round 6: the compiler's full identity in the key, the self-check mark, whose files are trusted ----

def test_compiler_identity_names_the_compiler_down_to_its_commit(rs, tmp_path):
    r = _child(tmp_path)   # (a bare process: this pytest process may have torch -- and with it another ROCm -- loaded, see below)
    ident = r["identity"]
    # hiprtc major.minor alone was the key until round 5; a patch-level update must change the text
    assert "hiprtc " in ident and "hip-runtime " in ident and "clang " in ident and "comgr " in ident, ident
    rt = int(ident.split("hip-runtime ")[1].split()[0])
    assert rt > 10_000_000 and rt % 100_000 != 0, ident          # major * 10^7 + minor * 10^5 + PATCH
    clang = ident.split("clang ")[1]
    assert any(len(tok.strip("()")) == 40 and all(ch in "0123456789abcdef" for ch in tok.strip("()")) for tok in clang.split()), \
        f"no 40-digit LLVM commit in the clang version string: {clang!r}"
    assert ident.encode() in Path(r["file"]).read_bytes()[:4096], "the identity is not part of the stored key text"


def test_a_process_that_loaded_torch_first_has_another_compiler_and_another_file(rs, tmp_path):
    """Found in round 6: this image holds TWO ROCm user spaces -- /opt/rocm (what hipcc and the library link against) and the one inside
    the torch wheel.  A process that imports torch first (bench.py does) gets torch's libhiprtc, i.e. ANOTHER clang, for its run-time
    builds.  hiprtc's major.minor is the same for both (9.0), so round 5's key served one compiler's code objects to the other; the
    identity tells them apart."""
    bare = _child(tmp_path)
    with_torch = _child(tmp_path, env_extra={"RS_TEST_IMPORT_TORCH_FIRST": "1"})
    if bare["identity"] == with_torch["identity"]:
        pytest.skip("torch runs on the system's ROCm here: one compiler")
    assert bare["identity"].split(" clang ")[0].split()[:2] == with_torch["identity"].split(" clang ")[0].split()[:2], "hiprtc major.minor differ too"
    assert bare["file"] != with_torch["file"] and with_torch["stats"]["misses"] == 1, (bare, with_torch)
    assert len(list(tmp_path.glob("*.rsco"))) == 2
    # radiosaber_amd.toolchain.prefer_system_compiler() before `import torch` (what bench.py does): the system's comgr is mapped first and
    # the process compiles with the toolchain the library was built with -- the bare process's compiler, the bare process's cache file
    fixed = _child(tmp_path, env_extra={"RS_TEST_IMPORT_TORCH_FIRST": "1", "RS_TEST_PREFER_SYSTEM_COMPILER": "1"})
    assert fixed["identity"].split(" clang ")[1] == bare["identity"].split(" clang ")[1], (fixed["identity"], bare["identity"])


def test_two_compiler_identities_give_two_files(rs, tmp_path):
    """yesterday's code objects must not be served after a compiler update.  RS_JIT_COMPILER_ID stands in for one."""
    a = _child(tmp_path, env_extra={"RS_JIT_COMPILER_ID": "hiprtc 9.0 patch 26015 clang 22.0.0git aaaa"})
    b = _child(tmp_path, env_extra={"RS_JIT_COMPILER_ID": "hiprtc 9.0 patch 26016 clang 22.0.0git bbbb"})
    assert a["file"] != b["file"] and a["stats"]["misses"] == 1 and b["stats"]["misses"] == 1
    assert len(list(tmp_path.glob("*.rsco"))) == 2
    again = _child(tmp_path, env_extra={"RS_JIT_COMPILER_ID": "hiprtc 9.0 patch 26015 clang 22.0.0git aaaa"})
    assert again["file"] == a["file"] and again["stats"]["hits"] == 1


def test_new_files_are_unchecked_private_and_foreign_or_writable_files_are_not_trusted(rs, tmp_path):
    cache = tmp_path / "deep" / "cache"
    first = _child(cache)
    f = Path(first["file"])
    assert f.read_bytes()[-8:] == b"UNCHECKD"                     # no process has self-checked it yet (that needs a GPU)
    assert (cache.stat().st_mode & 0o077) == 0 and (f.stat().st_mode & 0o077) == 0, "the cache is the caller's alone"
    # a file somebody else could have written is compiled again, not run
    f.chmod(0o666)
    r = _child(cache)
    assert r["stats"] == {"hits": 0, "misses": 1, "stores": 1, "rejected": 1}
    assert (f.stat().st_mode & 0o077) == 0
    # ... unless the deployment says the cache is shared
    f.chmod(0o664)
    assert _child(cache, env_extra={"RS_JIT_CACHE_SHARED": "1"})["stats"]["hits"] == 1
    # the mark is 8 bytes at the end, rewritten in place: a file that carries it is a hit like any other
    b = bytearray(f.read_bytes())
    b[-8:] = b"VERIFIED"
    f.chmod(0o600)
    f.write_bytes(bytes(b))
    assert _child(cache)["stats"] == {"hits": 1, "misses": 0, "stores": 0, "rejected": 0}
    # anything else in those 8 bytes just means "unchecked"
    b[-8:] = b"whatever"
    f.write_bytes(bytes(b))
    assert _child(cache)["stats"]["hits"] == 1
