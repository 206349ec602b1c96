"""Lint gfx950 assembly (hipcc -S, or llvm-objdump -d of a code object) for register-allocator code that LLVM placed at the head of a
join block BEFORE the instruction that re-enables the lanes (`s_or_b64 exec, exec, sN`, the lowered SI_END_CF).

Round 4 met a kernel (SubOpt's drop-in kernel, git 0adb0e5 with one line reverted) whose results were wrong on every wave but
one; profiles/r05_onelane.md tracks it down to

    .LBB19_220:                         <- target of `s_cbranch_execz` (waves whose mask is empty arrive with EXEC = 0)
        v_mov_b64_e32 v[4:5], v[28:29]              <- live-range split copy
        scratch_store_dwordx2 off, v[30:31], off    <- 8-byte Folded Spill   (1.797e308, SubOpt's `least`)
        s_or_b64 exec, exec, s[0:1]                 <- lanes come back only here

the spill store runs under the narrowed EXEC (nothing is stored on a wave that skipped the region), the reload later runs with all
lanes on and reads whatever the scratch slot held.  An instruction that moves per-lane data (VALU move, scratch / LDS / global
access) between a block's entry and its leading exec restore is never what the source meant; v_readlane / v_writelane /
v_readfirstlane and every s_* instruction ignore EXEC and are fine.

    python tools/lint_exec_restore.py FILE [...]        exit 1 and one line per finding when anything is flagged
FILE: hipcc -S output (*.s), a HIP shared library (*.so: its gfx950 code object is taken out of the fat binary), a code object
(*.co / *.hsaco) or an entry of radiosaber_amd's disk cache of run-time compiled kernels (*.rsco).  tests/test_exec_restore_lint.py
runs it over the library and over the kernels hiprtc builds for the shapes the benchmark and the tests use.
"""
import re
import sys

LANE_FREE = ("v_readlane_b32", "v_writelane_b32", "v_readfirstlane_b32")
# hipcc -S:  "name:" / ".LBB1_2:"        llvm-objdump -d --symbolize-operands:  "0000000000010e18 <L2>:" / "... <name>:"
LABEL = re.compile(r"^(?:[0-9a-f]+ <([\w.$]+)>|(\.LBB\w+|[A-Za-z_][\w.$]*)):")
EXEC_RESTORE = re.compile(r"^\s*s_or_b64\s+exec,\s*exec,")
EXEC_WRITE = re.compile(r"^\s*s_\w+\s+exec\b|^\s*s_\w*saveexec")


def per_lane(op):
    if op in LANE_FREE:
        return False
    return op.startswith(("v_", "ds_", "scratch_", "global_", "flat_", "buffer_"))


def is_local(label):
    return label.startswith(".L") or re.fullmatch(r"L\d+", label) is not None


def lint_function(func, lines, name):
    """lines: (line number, raw text) of one function"""
    # join blocks: labels some `s_cbranch_execz` jumps to -- the skip edge of an IF, taken with the narrowed (possibly empty) EXEC
    skip_targets = set()
    for _, raw in lines:
        m = re.match(r"^\s*s_cbranch_execz\s+(\S+)", raw)
        if m:
            skip_targets.add(m.group(1))
    findings = []
    block, head, open_head = None, [], False  # head: per-lane instructions since the block's label, while no exec write was seen
    for ln, raw in lines:
        line = raw.split(";")[0].split("//")[0].rstrip()
        m = LABEL.match(line.strip())
        if m:
            block = m.group(1) or m.group(2)
            head, open_head = [], block in skip_targets
            continue
        s = line.strip()
        if not s or s.startswith(".") or not open_head:
            continue
        op = s.split()[0]
        if EXEC_RESTORE.match(line):
            for (l2, ins, raw2) in head:
                findings.append(f"{name}:{l2}: {func} {block}: `{ins}` runs before the block's `{s}`" +
                                ("  [" + raw2.split(";")[1].strip() + "]" if ";" in raw2 else ""))
            open_head = False
            continue
        if EXEC_WRITE.match(line) or op.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_barrier", "s_setpc")):
            open_head = False  # some other exec change or the block's end: not a join-block head
            continue
        if per_lane(op):
            head.append((ln, s, raw))
    return findings


def lint_text(text, name="<asm>"):
    findings, func, cur = [], "?", []
    for ln, raw in enumerate(text.split("\n"), 1):
        m = LABEL.match(raw.split(";")[0].strip())
        if m and not is_local(m.group(1) or m.group(2)):
            findings += lint_function(func, cur, name)
            func, cur = (m.group(1) or m.group(2)), []
        cur.append((ln, raw))
    findings += lint_function(func, cur, name)
    return findings


def disassemble(code_object_bytes):
    """gfx950 ELF -> llvm-objdump text with symbolized branch targets"""
    import shutil
    import subprocess
    import tempfile
    objdump = shutil.which("llvm-objdump") or "/opt/rocm/lib/llvm/bin/llvm-objdump"
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(code_object_bytes)
        f.flush()
        return subprocess.run([objdump, "-d", "--mcpu=gfx950", "--symbolize-operands", f.name], capture_output=True, text=True,
                              check=True).stdout


def code_of_cache_file(path):
    """radiosaber_amd's disk cache entry (rs_jit.cpp): "RSJC2\\n" | u64 key length | key | u64 code length | u64 checksum | code | 8-byte
    self-check mark ("VERIFIED" / "UNCHECKD")"""
    import struct
    b = open(path, "rb").read()
    assert b[:6] == b"RSJC2\n", path
    klen = struct.unpack_from("<Q", b, 6)[0]
    key = b[14:14 + klen].decode()
    clen = struct.unpack_from("<Q", b, 14 + klen)[0]
    return key, b[30 + klen:30 + klen + clen]


def code_of_library(path):
    """the gfx950 code object inside a HIP shared library (.hip_fatbin bundle)"""
    import shutil
    import subprocess
    import tempfile
    llvm = "/opt/rocm/lib/llvm/bin/"
    objcopy = shutil.which("llvm-objcopy") or llvm + "llvm-objcopy"
    bundler = shutil.which("clang-offload-bundler") or llvm + "clang-offload-bundler"
    with tempfile.TemporaryDirectory() as d:
        subprocess.run([objcopy, "--dump-section", f".hip_fatbin={d}/fat.bin", str(path)], check=True)
        subprocess.run([bundler, "--unbundle", "--type=o", f"--input={d}/fat.bin", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                        f"--output={d}/gfx950.co"], check=True)
        return open(f"{d}/gfx950.co", "rb").read()


def main(argv):
    bad, n = [], 0
    for f in argv:
        n += 1
        if f.endswith(".rsco"):
            key, code = code_of_cache_file(f)
            bad += lint_text(disassemble(code), f)
        elif f.endswith(".so"):
            bad += lint_text(disassemble(code_of_library(f)), f)
        elif f.endswith((".co", ".hsaco")):
            bad += lint_text(disassemble(open(f, "rb").read()), f)
        else:
            bad += lint_text(open(f).read(), f)
    for b in bad:
        print(b)
    print(f"{len(bad)} finding(s) in {n} file(s)")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
