"""bench.py's roofline arithmetic (no GPU): SURVEY 8(d)'s algorithmic bytes per TTI, per scheduler since round 6 (the
sched-9 formula applied to scheduler 7 printed nominal fractions above 1)."""
import importlib.util
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
spec = importlib.util.spec_from_file_location("bench_module", ROOT / "bench.py")
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)


def test_survey_8d_figures_for_the_transport_schedulers():
    # 22 920 B at U=500, R=25, S=20; 42 576 B at R=64; 45 420 B at U=1 000, R=25
    for sched in (8, 9, 10, 101, 103):
        assert bench.algorithmic_bytes_per_tti(500, 25, 20, sched) == 22920
        assert bench.algorithmic_bytes_per_tti(500, 64, 20, sched) == 42576
        assert bench.algorithmic_bytes_per_tti(1000, 25, 20, sched) == 45420
    assert bench.algorithmic_bytes_per_tti(500, 25, 20) == 22920  # the default is the headline scheduler


def test_per_scheduler_bytes():
    # per-flow PF keeps no slice state
    assert bench.algorithmic_bytes_per_tti(500, 25, 20, 1) == 22920 - 2 * 20 * 8
    # NVS serves one slice per TTI: its rows, its averages, the per-slice EWMA times
    assert bench.algorithmic_bytes_per_tti(500, 25, 20, 7) == 25 * 25 + 20 * 25 + 4 * 25 + 2 * 20 * 8
    assert bench.algorithmic_bytes_per_tti(500, 25, 20, 11) == bench.algorithmic_bytes_per_tti(500, 25, 20, 7)


def test_no_recorded_rate_gives_a_fraction_above_one():
    # the fastest rates ever recorded per (scheduler, shape) (profiles/r05_notes.md 5, r05_sched_sweep.md), M TTIs/s
    best = {(7, 500, 25): 267.8, (7, 1000, 25): 215.1, (7, 500, 64): 218.7, (1, 500, 25): 169.6, (1, 500, 64): 107.4, (8, 500, 25): 103.9,
            (9, 500, 25): 36.1, (10, 500, 25): 57.1, (11, 500, 25): 19.6}
    for (sched, U, R), m in best.items():
        frac = bench.algorithmic_bytes_per_tti(U, R, 20, sched) * m * 1e6 / 1e9 / bench.HBM_PEAK_GBS
        assert 0 < frac < 1, (sched, U, R, frac)
