"""Rules over the device sources that the compiler cannot check.

One-lane regions: `if (lane == 0)` / `if (tid == 0)` / "the lane on a sub-range's first position" regions that contain a loop (or call
one of the serial walks) must be on the list below, with the GPU test that executes the region.  Round 4 suspected such a region of a
miscompilation; round 5 found the defect elsewhere (profiles/r05_onelane.md: spill code in front of a join block's exec restore,
tools/lint_exec_restore.py) -- what stays is the discipline that no long serial walk under a one-lane mask goes unexecuted."""
import re
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
CSRC = ROOT / "radiosaber_amd" / "csrc"
FILES = ["rs_kernels.hip", "rs_wave.h", "rs_sort_device.h", "rs_interslice.h"] + sorted(p.name for p in CSRC.glob("rs_phase_*.inc"))
ONE_LANE = re.compile(r"\b(lane|tid)\s*==\s*0\b|\bF\[i\]\s*==\s*x\b|\bx\s*==\s*F(\[i\])?\b|segF\[x\]\s*==\s*x")
SERIAL = re.compile(r"\b(for|while|do)\b|\b(heap_sort|rs_umap_order|introsort_loop|adjust_heap)\s*\(")

# (file, first words of the condition) -> the GPU test that runs the region
ALLOWED = {
    ("rs_kernels.hip", "tid == 0"): "every GPU test: the cell's scalars, the device error word and the heap-sort counters (a three-entry "
                                    "loop) are written back by thread 0 at the end of a launch; tests/test_sort_killers.py reads the counters",
}


def _region(text, i):
    depth, j = 0, i
    while True:
        depth += {"(": 1, ")": -1}.get(text[j], 0)
        if depth == 0:
            break
        j += 1
    cond = " ".join(text[i + 1:j].split())
    k = j + 1
    while text[k] in " \t\n":
        k += 1
    if text[k] != "{":
        return cond, text[k:text.index(";", k) + 1]
    depth, e = 0, k
    while True:
        depth += {"{": 1, "}": -1}.get(text[e], 0)
        if depth == 0:
            break
        e += 1
    return cond, text[k:e + 1]


def one_lane_regions_with_loops():
    out = []
    for f in FILES:
        text = (CSRC / f).read_text()
        text = re.sub(r"/\*.*?\*/", lambda m: " " * len(m.group(0)), text, flags=re.S)  # comments out, offsets kept
        for m in re.finditer(r"\bif\s*(?:constexpr\s*)?\(", text):
            cond, body = _region(text, m.end() - 1)
            if ONE_LANE.search(cond) and SERIAL.search(body):
                out.append((f, text.count("\n", 0, m.start()) + 1, cond))
    return out


def test_one_lane_regions_with_loops_are_listed_with_the_gpu_test_that_runs_them():
    found = one_lane_regions_with_loops()
    unlisted = [(f, line, cond) for (f, line, cond) in found if not any(f == af and cond.startswith(ac) for (af, ac) in ALLOWED)]
    assert not unlisted, ("one-lane regions around a loop that no GPU test is named for (run the walk on every lane with scalar operands, "
                          "as heap_sort_on_wave does, or list the region with its test): " + repr(unlisted))
    # the list carries no dead entries
    for (af, ac) in ALLOWED:
        assert any(f == af and cond.startswith(ac) for (f, _, cond) in found), (af, ac)


def test_the_heap_sort_fallback_runs_on_whole_waves():
    """the three std::__partial_sort sites go through heap_sort_on_wave (scalar operands, every lane), never a one-lane call"""
    text = (CSRC / "rs_sort_device.h").read_text()
    assert text.count("heap_sort_on_wave(v,") == 3
    assert len(re.findall(r"rs_sort::heap_sort\(", text)) == 1  # inside heap_sort_on_wave only
    assert "atomicAdd(&m->heap_sorts[site], 1)" in text


def test_the_detector_sees_a_region_it_should():
    bad = "void f() { if (lane == 0 && more) n = rs_umap_order(a, b); if (tid == 0) { for (int i = 0; i < 9; ++i) g(i); } if (lane == 0) x = 1; }"
    hits = [m for m in re.finditer(r"\bif\s*(?:constexpr\s*)?\(", bad)]
    flagged = []
    for m in hits:
        cond, body = _region(bad, m.end() - 1)
        flagged.append(bool(ONE_LANE.search(cond) and SERIAL.search(body)))
    assert flagged == [True, True, False]
