"""The margin behind "held winners" (DESIGN.md 2.12), attacked on the CPU with the device's own arithmetic.

A (slice, RBG) item's winner w is held when its stage-1 value leads every other user's by mu = 2^-18 + 2 / (1 + avg_w) (and
avg_w >= 64); the claim is that then, for up to RS_HOLD_MAX_AGE = 40 TTIs in which w is not served, the reference's rounded FP64
metric of w stays STRICTLY above every other user's -- whatever the others do (being served only lowers a metric, so the worst
opponent is one that is never served either).  This test builds opponents that pass the device's held test by the smallest
possible amount, with the stage-1 reciprocal pushed one ulp in the unfavourable direction on both sides, lets both averages decay
for 40 TTIs exactly as the kernel's EWMA does (separate multiply, add of beta * 0, clamp, 1 + avg, division by 1000, the metric's
division), and requires q_w > q_v in every TTI."""
import re
from pathlib import Path

import numpy as np

# the age cap comes from the kernel source itself: the bound proved here and the constant the device uses cannot drift apart
_SRC = (Path(__file__).resolve().parents[1] / "radiosaber_amd" / "csrc" / "rs_kernels.hip").read_text()
MAX_AGE = int(re.search(r"#define RS_HOLD_MAX_AGE (\d+)", _SRC).group(1))
C = 1.0 - 0.02  # the kernel's (1 - beta) in double
KBPS = np.array([16, 32, 56, 88, 120, 136, 176, 224, 280, 328, 376, 440.00000000000006, 520, 584, 712])


def stage1(num, avg, ulp):
    """fl32(num) * rcp32(fl32(den)) with the reciprocal off by `ulp` ulps (v_rcp_f32 is within 1)."""
    den = (1.0 + avg) / 1000.0
    r = (np.float32(1.0) / den.astype(np.float32)).astype(np.float32)
    r = np.nextafter(r, np.float32(np.inf) if ulp > 0 else np.float32(-np.inf)) if ulp else r
    return (num.astype(np.float32) * r).astype(np.float32)


def held(la, t2, aw):
    """the device's test (rs_kernels.hip, scan_item / the listed items' scan), in float32"""
    mu = np.float32(2.0 ** -18) + np.float32(2.0) / (np.float32(1.0) + aw.astype(np.float32))
    lhs = (t2 * (np.float32(1.0) + mu)).astype(np.float32) * np.float32(1.000001)
    return (aw >= 64.0) & (lhs.astype(np.float32) <= la)


def metric(num, avg):
    return num / ((1.0 + avg) / 1000.0)


def test_a_held_winner_is_never_overtaken_within_40_unserved_ttis():
    rng = np.random.default_rng(12)
    n = 400_000
    avg_w = np.exp(rng.uniform(np.log(64.0), np.log(5e7), n))
    num_w = KBPS[rng.integers(0, 15, n)]
    num_v = KBPS[rng.integers(0, 15, n)]
    # the opponent's average that makes the exact metrics equal, then pushed away until the held test just passes
    avg_v = (1.0 + avg_w) * num_v / num_w - 1.0
    keep = avg_v >= 1.0
    avg_w, num_w, num_v, avg_v = avg_w[keep], num_w[keep], num_v[keep], avg_v[keep]
    la = stage1(num_w, avg_w, -1)  # the winner's reciprocal one ulp low ...
    lo, hi = avg_v.copy(), avg_v * 1.5 + 10.0  # bisection on the opponent's average: the smallest one that is held against
    for _ in range(60):
        mid = 0.5 * (lo + hi)
        ok = held(la, stage1(num_v, mid, +1), avg_w)  # ... the opponent's one ulp high
        hi = np.where(ok, mid, hi)
        lo = np.where(ok, lo, mid)
    avg_v = hi
    assert held(la, stage1(num_v, avg_v, +1), avg_w).all()
    worst = np.inf
    aw, av = avg_w.copy(), avg_v.copy()
    assert MAX_AGE <= 40, "the margin mu = 2^-18 + 2 / (1 + avg_w) is sized for at most 40 unserved TTIs (DESIGN.md, held winners)"
    for _ in range(MAX_AGE + 1):
        qw, qv = metric(num_w, aw), metric(num_v, av)
        assert (qw > qv).all(), "a held winner was caught"
        worst = min(worst, float(((qw - qv) / qw).min()))
        aw = np.maximum(C * aw + 0.02 * 0.0, 1.0)
        av = np.maximum(C * av + 0.02 * 0.0, 1.0)
    assert worst > 1e-7  # the margin that is left after 40 TTIs (2^-18 - 2^-20.4 = 3e-6 at the start)


def test_the_loss_per_tti_is_bounded_by_the_winners_own_average():
    """the inequality the margin is built on: one unserved TTI takes at most (1 - c) / (c (1 + avg_w)) (+ rounding) from q_w / q_v"""
    rng = np.random.default_rng(5)
    n = 200_000
    aw = np.exp(rng.uniform(np.log(2.0), np.log(1e8), n))
    av = np.exp(rng.uniform(np.log(1.0), np.log(1e9), n))
    num = KBPS[rng.integers(0, 15, n)]
    before = metric(num, aw) / metric(num, av)
    after = metric(num, np.maximum(C * aw, 1.0)) / metric(num, np.maximum(C * av, 1.0))
    bound = (1.0 - (1.0 - C) / (C * (1.0 + aw))) * (1.0 - 2.0 ** -48)
    assert (after / before >= bound).all()
