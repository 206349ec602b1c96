"""The C ABI library (not gpu): builds for gfx950, loads, exports every symbol the header declares,
computes the host-libm tables, and refuses to compute without a device (no CPU fallback)."""
import json
import re
from pathlib import Path

import numpy as np
import pytest

from conftest import GOLDEN

ROOT = Path(__file__).resolve().parents[1]
KA = json.loads((GOLDEN / "appendix_a.json").read_text())


def _header_functions():
    txt = (ROOT / "include" / "radiosaber_hip.h").read_text()
    txt = re.sub(r"/\*.*?\*/", " ", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(rs_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(rs):
    from radiosaber_amd import api
    L = rs.lib()
    declared = _header_functions()
    assert declared, "no prototypes found"
    for name in declared:
        assert hasattr(L, name), f"{name} declared in include/radiosaber_hip.h but not exported"
    assert sorted(api.ABI_SYMBOLS) == declared
    assert L.rs_abi_version() == 11


def test_code_object_is_gfx950(rs):
    data = (ROOT / "radiosaber_amd" / "libradiosaber_hip.so").read_bytes()
    assert b"gfx950" in data and b"rs_cell_kernel" in data


def _check_link_tables(rs):
    """This machine's libm against the pinned glibc-2.35 set and against the values the unmodified reference printed."""
    hex_of = lambda a: [float.hex(float(x)) for x in a]  # noqa: E731
    want_e = [float.hex(float.fromhex(h)) for h in KA["eesm_E_hex"]]
    want_x = [float.hex(float.fromhex(h)) for h in KA["eesm_X_hex"]]
    pinned = rs.link_tables(pinned=True)
    assert pinned["eff"][1:].tolist() == KA["eff_of_cqi"] and pinned["kbps"][1:].tolist() == KA["kbps_of_cqi"]
    assert hex_of(pinned["eesm_e"][1:]) == want_e and hex_of(pinned["eesm_x"][1:14]) == want_x, "the pinned set is not Appendix A's"
    t = rs.link_tables()
    assert t["eff"][1:].tolist() == KA["eff_of_cqi"]
    assert t["kbps"][1:].tolist() == KA["kbps_of_cqi"]
    n, text = rs.link_tables_compare()
    import platform
    libc = " ".join(platform.libc_ver())
    assert n == 0 and hex_of(t["eesm_e"][1:]) == want_e and hex_of(t["eesm_x"][1:14]) == want_x, \
        f"this machine's libm ({libc}) gives {n} EESM constant(s) that differ from glibc 2.35's: {text}"
    return libc


def test_link_tables_match_reference_known_answers(rs):
    _check_link_tables(rs)


@pytest.mark.gpu
def test_link_tables_of_the_gpu_box_match_reference_known_answers(rs):
    """the same check where the GPU runs -- there the device and the oracle share the box's libm, so they would agree
    with each other whatever it returned; this is the test that says which libm the box has and that it is the fixtures'."""
    libc = _check_link_tables(rs)
    print(f"GPU box libm: {libc}; host-evaluated EESM constants == pinned glibc-2.35 set == SURVEY Appendix A")
    # a drop-in context created with the default follows the host's libm and has nothing to warn about here
    ts = rs.TtiScheduler(rs.SliceConfig([2, 2]), 12, 2)
    assert ts.create_warning == "", ts.create_warning
    ts.close()


def test_threshold_classification_equals_libm_formula(rs, oracle):
    """final CQI by thresholds on x == the reference's dB formula, on random allocations."""
    t = rs.link_tables()
    rng = np.random.default_rng(9)
    for _ in range(20000):
        n_rbg = int(rng.integers(1, 65))
        G = int(rng.choice([2, 4, 8]))
        cq = rng.integers(1, 16, n_rbg).astype(np.uint8)
        if rng.random() < 0.5:
            cq[:] = cq[0]
        prb = np.repeat(cq, G)
        s = 0.0
        for c in prb:
            s += t["eesm_e"][c]
        x = s / len(prb)
        mine = 15 if x == 0 else 1 + int((x <= t["eesm_x"][1:14]).sum())
        assert mine == oracle.final_cqi(prb)


def test_no_cpu_fallback(rs):
    if rs.device_count() > 0:
        pytest.skip("a GPU is present")
    sc = rs.SliceConfig([2, 2])
    with pytest.raises(rs.RadioSaberError) as e:
        rs.BatchScheduler(sc, 12, 2, 1)
    assert "no HIP device" in str(e.value)
    with pytest.raises(rs.RadioSaberError):
        rs.TtiScheduler(sc, 12, 2)


def test_config_validation_messages(rs):
    L = rs.lib()
    for kw, frag in ((dict(algo_alpha=[2, 0]), "algo_alpha"), (dict(algo_epsilon=[2, 1]), "algo_epsilon")):
        sc = rs.SliceConfig([2, 2], **kw)
        with pytest.raises(rs.RadioSaberError) as e:
            rs.BatchScheduler(sc, 12, 2, 1)
        assert frag in str(e.value)
    with pytest.raises(rs.RadioSaberError) as e:
        rs.BatchScheduler(rs.SliceConfig([2, 2]), 65, 8, 1)
    assert "n_rbgs" in str(e.value)


def test_slice_config_from_reference_json(rs):
    cfg = {"slices": [{"n_slices": 20, "weight": 0.05, "algo_alpha": 0, "algo_beta": 0, "algo_epsilon": 1, "algo_psi": 1}],
           "ues_per_slice": [5] * 20}
    sc = rs.SliceConfig.from_json(cfg)
    assert sc.n_slices == 20 and sc.n_users == 100 and sc.weight == [0.05] * 20
    assert sc.user_to_slice.tolist() == [i // 5 for i in range(100)]


def test_product_never_touches_the_oracle():
    """oracle/ is test infrastructure: nothing under radiosaber_amd/ or include/ may import, include,
    link or execute it (the oracle includes the product's table data, not the other way round)."""
    bad = []
    for f in list((ROOT / "radiosaber_amd").rglob("*")) + list((ROOT / "include").rglob("*")):
        if f.is_file() and f.suffix in (".py", ".h", ".hpp", ".hip", ".cpp", ".inc"):
            for n, line in enumerate(f.read_text(errors="ignore").splitlines(), 1):
                code = line.split("//")[0]
                if re.search(r"(import|from|include|CDLL|dlopen).*\boracle", code):
                    bad.append(f"{f.relative_to(ROOT)}:{n}: {line.strip()}")
    assert not bad, bad
    so = (ROOT / "radiosaber_amd" / "libradiosaber_hip.so").read_bytes()
    assert b"librs_oracle" not in so and b"rso_" not in so


@pytest.mark.parametrize("shape", [(20, 500, 25, 4, 512, 9), (20, 100, 64, 8, 512, 9), (5, 23, 25, 4, 128, 9),
                                   (20, 500, 25, 4, 512, 8), (20, 1000, 25, 4, 512, 1), (20, 500, 64, 8, 256, 7)])
def test_shape_specialised_source_compiles_for_gfx950(rs, shape):
    """hiprtc build of the embedded kernel source with the cell shape as compile-time constants (no GPU needed)."""
    assert rs.jit_selfcheck(*shape) > 10000


@pytest.mark.parametrize("shape", [(20, 194, 64, 8, 512, 9), (20, 194, 64, 8, 512, 7), (20, 194, 64, 8, 512, 1),
                                   (20, 500, 25, 4, 512, 1), (20, 1000, 25, 4, 512, 1), (4, 18, 25, 4, 64, 8), (3, 30, 12, 2, 128, 103)])
def test_queue_model_kernels_compile_for_gfx950(rs, shape):
    """The queue-model code object of a batch (rs_batch_set_bearers switches to it): bearer words in LDS, sched 7's metric table,
    sched 1's flows in registers (U = 194: 7 per lane, U = 500: 16; U = 1 000 falls back to the chunked loop)."""
    assert rs.jit_selfcheck(*shape[:5], sched=shape[5], queues=True) > 10000


def test_headline_shapes_keep_four_cells_per_cu(rs):
    """160 KB of LDS per CU: a cell of the headline shape (20 slices x 500 UEs x 25 RBGs) must stay at or under 40 960 B so
    that large batches place four cells on a CU (measured: 38 M instead of 23-29 M TTIs/s at 1 024+ cells), and the
    as-shipped 64-RBG grid under 81 920 B (two per CU)."""
    for sched in (1, 7, 8, 9):
        for threads in (256, 512):
            assert rs.lds_bytes_per_cell(20, 500, 25, sched, threads) <= 40960, (sched, threads)
            if (sched, threads) != (9, 256):  # 1 280 records on 256 threads = the state-in-LDS form of the sort: not a tuned shape
                assert rs.lds_bytes_per_cell(20, 500, 64, sched, threads) <= 81920, (sched, threads)
    assert rs.lds_bytes_per_cell(20, 500, 25, 10, 512) <= 40960
    with pytest.raises(rs.RadioSaberError):
        rs.lds_bytes_per_cell(65, 500, 25)


def test_sampler_cells_keep_two_per_cu(rs):
    """Round 5: the NVS non-greedy sampler (sched 11) keeps 16-bit winner indices instead of metric doubles and no record array; a cell
    of 20 slices x 25 UEs x 64 RBGs is exactly 81 920 B (two per CU; 128 000 B before: one), the reference's exp-nongreedy shapes
    of 10 / 20 UEs per slice fit too (30 per slice: the 38 KB CQI grid keeps it at one).  The drop-in query knows no slice window
    (the metric array is sized for U users)."""
    assert rs.lds_bytes_per_cell(20, 500, 64, 11, 512) == 81920
    for users in (200, 400):
        assert rs.lds_bytes_per_cell(20, users, 64, 11, 512) <= 81920
    assert rs.lds_bytes_per_cell(20, 600, 64, 11, 512) > 81920
    assert rs.lds_bytes_per_cell(20, 500, 25, 11, 512) <= 81920


def test_division_by_1000_in_three_operations_is_correctly_rounded():
    """The kernel replaces `averageRate /= 1000.0` by q = x * 0.001; r = fma(-q, 1000, x); fma(r, 0.001, q) (rs_div_1000, Markstein's
    theorem).  Checked here in exact rational arithmetic -- float(Fraction) rounds to nearest even, so an fma is one exact
    expression rounded once -- on random inputs over the PF averages' range, on inputs whose quotient lies next to a midpoint of two
    doubles, and on small integers."""
    from fractions import Fraction
    import math
    import random

    y = 0.001

    def fma(a, b, c):
        return float(Fraction(a) * Fraction(b) + Fraction(c))

    def div3(x):
        q = x * y
        return fma(fma(-q, 1000.0, x), y, q)

    rnd = random.Random(7)
    xs = [float(i) for i in range(1, 3000)]
    xs += [1.0 + rnd.random() * 10 ** rnd.uniform(0, 9) for _ in range(60000)]
    xs += [math.ldexp(1.0 + rnd.random(), rnd.randrange(0, 40)) for _ in range(20000)]
    # quotients half an ulp away from a double: x = RN(1000 * (q + ulp(q)/2)) and its neighbours
    for _ in range(20000):
        q = math.ldexp(1.0 + rnd.random(), rnd.randrange(-9, 30))
        mid = Fraction(q) + Fraction(math.ulp(q)) / 2
        x = float(mid * 1000)
        xs += [x, math.nextafter(x, math.inf), math.nextafter(x, 0.0)]
    for x in xs:
        if x >= 1.0:
            assert div3(x) == x / 1000.0, x
