"""Build the gfx950 shared library radiosaber_amd/libradiosaber_hip.so in-tree with hipcc.

    python -m radiosaber_amd.build [--force]

The library is the product: hand-written HIP kernels (csrc/rs_kernels.hip) + the C ABI host side
(csrc/rs_api.cpp).  -ffp-contract=off is mandatory: results must round like the reference's
x86-64 SSE2 build (no FMA contraction), on the device and in the host table code alike.
"""
import os
import shutil
import subprocess
import sys
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
LIB = PKG / "libradiosaber_hip.so"
SOURCES = [CSRC / "rs_kernels.hip", CSRC / "rs_api.cpp"]
DEPS = SOURCES + [CSRC / "rs_device.h", CSRC / "rs_sort_emul.h", CSRC / "rs_amc_tables.inc",
                  PKG.parent / "include" / "radiosaber_hip.h"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
         "-fPIC", "-shared", "-Wall", "-Wno-unused-function", "-Wno-missing-braces"]


def hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and Path(c).exists():
            return c
    raise RuntimeError("hipcc not found")


def needs_build():
    if not LIB.exists():
        return True
    t = LIB.stat().st_mtime
    return any(d.stat().st_mtime > t for d in DEPS)


def build_variant(name, defines):
    """Diagnostic variants (e.g. name='stamps', defines=['-DRS_STAMPS']); never the product library."""
    out = PKG / f"libradiosaber_hip_{name}.so"
    cmd = [hipcc()] + FLAGS + list(defines) + [str(s) for s in SOURCES] + ["-o", str(out)]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + r.stdout)
    return out


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    cmd = [hipcc()] + FLAGS + [str(s) for s in SOURCES] + ["-o", str(LIB)]
    if verbose:
        print(" ".join(cmd))
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + r.stdout)
    if verbose and r.stdout.strip():
        print(r.stdout)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
