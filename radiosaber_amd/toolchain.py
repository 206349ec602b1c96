"""Which compiler builds the run-time kernels of a Python process.

This image (and any box with a ROCm-enabled torch wheel) holds TWO ROCm user spaces: the system's (`/opt/rocm`, the one `hipcc` and
`libradiosaber_hip.so` are built with) and the one bundled inside the torch wheel.  The compiler behind hiprtc lives in
`libamd_comgr.so.3`, and a process uses whichever copy it loaded first: a process that imports torch first compiles the library's
run-time kernels with the wheel's clang (ROCm 7.0.2: clang 20 here), every other process -- the C++ simulator, the test suite, anything
under rocprofv3, whose tool library pulls the system's comgr in -- with the system's (ROCm 7.2: clang 22).  Round 6 measured the
difference on the headline kernel: 114.2 ms per 8 000-TTI launch built by clang 22 against 116.4 ms built by clang 20 on the same lease
(profiles/r06_notes.md section 7) -- which is also what rounds 4 and 5 chased as "rocprofv3 sees the kernel 2 % faster".

`prefer_system_compiler()` loads the system's comgr into the process BEFORE torch is imported, so that hiprtc -- whoever's copy --
binds to it: the run-time builds then come from the toolchain the library itself was built with, in every process alike.  It does
nothing when the file is missing or a comgr is already loaded; `radiosaber_amd.jit_compiler_identity()` says what a process ended up
with (it is part of every cache key, and bench.py prints it as `compiler`)."""
import ctypes
import os
from pathlib import Path


def comgr_loaded():
    """Path of the libamd_comgr this process has mapped, or None."""
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                if "libamd_comgr" in line:
                    return line.split()[-1]
    except OSError:
        pass
    return None


def prefer_system_compiler():
    """Load <ROCM_PATH or /opt/rocm>/lib/libamd_comgr.so.3 globally unless a comgr is already mapped.  Call before `import torch`.
    Returns the path now in use (None: no comgr found)."""
    have = comgr_loaded()
    if have:
        return have
    if os.environ.get("RS_SYSTEM_COMGR", "1") == "0":  # A/B switch: leave the choice to the import order
        return None
    for root in (os.environ.get("ROCM_PATH"), "/opt/rocm"):
        if not root:
            continue
        for name in ("libamd_comgr.so.3", "libamd_comgr.so"):
            p = Path(root) / "lib" / name
            if p.exists():
                try:
                    ctypes.CDLL(str(p), mode=ctypes.RTLD_GLOBAL)
                    return str(p)
                except OSError:
                    continue
    return None
