"""CPU models of two exactness arguments of round 4's second half (no GPU needed), with the device's own arithmetic in numpy:

1. per-flow PF on lanes (kPf1, rs_phase_p3.inc): the stage-1 value of a user is the FP32 product fl32(num) * rcp32(fl32(avg)) with the
   user's position in the lane written into its five low mantissa bits; the filter keeps the users whose word is at least
   (1 - 2^-16) of the largest word.  Claim: a user whose rounded FP64 metric num / avg is >= the best one's is never filtered out
   (whatever the reciprocal's 1-ulp error does), so settling the survivors exactly gives the reference's first maximum.
2. the NVS non-greedy sampler on 16-bit keys (rs_phase_nvs_sampler.inc): ranking the slice's metrics (rank = 1 + number of strictly
   smaller ones, 0 for an ineligible draw's metric 0.0) and taking the largest rank << 6 | 63 - i is the scan `if (hm < metric)`
   from -1 over the users in order: same winner, same winning metric, ties and all-zero rows included.
"""
import numpy as np

PFNUM = np.array([16, 32, 56, 88, 120, 136, 176, 224, 280, 328, 376, 440.00000000000006, 520, 584, 712]) * 1000.0  # eff * 180000.
TOL16 = np.float32(1.0 - 2.0 ** -16)


def _word(num, avg, ulp, k):
    """(bits of fl32(num) * rcp32(fl32(avg)), reciprocal off by `ulp` ulps) with the position 31 - k in the five low bits"""
    r = (np.float32(1.0) / avg.astype(np.float32)).astype(np.float32)
    if ulp:
        r = np.nextafter(r, np.float32(np.inf) if ulp > 0 else np.float32(-np.inf))
    a = (num.astype(np.float32) * r).astype(np.float32)
    return (a.view(np.int32) & np.int32(~31)) | np.int32(31 - k)


def test_per_flow_pf_filter_never_drops_a_user_that_can_win_or_tie():
    rng = np.random.default_rng(21)
    n = 600_000
    avg_b = np.exp(rng.uniform(0.0, np.log(1e9), n))  # the best user's average, 1 ... 1e9 (the EWMA clamps at 1)
    num_b = PFNUM[rng.integers(0, 15, n)]
    num_v = PFNUM[rng.integers(0, 15, n)]
    # an opponent whose exact metric is equal to the best one's, then nudged a few ulps either way: ties and near-ties are the
    # cases the filter must not lose
    avg_v = avg_b * num_v / num_b
    avg_v = avg_v * (1.0 + rng.integers(-6, 7, n) * 2.0 ** -52)
    keep = avg_v >= 1.0
    avg_b, num_b, num_v, avg_v = avg_b[keep], num_b[keep], num_v[keep], avg_v[keep]
    q_b, q_v = num_b / avg_b, num_v / avg_v  # the reference's expression, rounded FP64 division
    can_win = q_v >= q_b
    assert can_win.sum() > 50_000 and (~can_win).sum() > 50_000
    worst = 0
    for ub in (-1, 0, 1):
        for uv in (-1, 0, 1):
            for kb, kv in ((0, 31), (31, 0), (5, 5)):  # every combination of position bits: they move a word by < 2^-18
                w_b, w_v = _word(num_b, avg_b, ub, kb), _word(num_v, avg_v, uv, kv)
                top = np.maximum(w_b, w_v)
                thr = (top.view(np.float32) * TOL16).astype(np.float32).view(np.int32)
                dropped = can_win & (w_v < thr)
                worst = max(worst, int(dropped.sum()))
                # and the best user itself always survives its own threshold
                assert (w_b >= (w_b.view(np.float32) * TOL16).astype(np.float32).view(np.int32)).all()
    assert worst == 0
    # the margin is real: with the 2^-19 tolerance of the plain stage-1 filter the position bits WOULD drop winners
    w_b, w_v = _word(num_b, avg_b, 1, 0), _word(num_v, avg_v, -1, 31)
    thr19 = (np.maximum(w_b, w_v).view(np.float32) * np.float32(1.0 - 2.0 ** -19)).astype(np.float32).view(np.int32)
    assert (can_win & (w_v < thr19)).any()


def test_ranked_16_bit_keys_pick_the_scan_s_first_maximum():
    rng = np.random.default_rng(22)
    for trial in range(400):
        n = int(rng.integers(1, 65))
        # metrics of the slice's n users for the 4 possible draws: few distinct values (whole classes tie, as on the first TTIs)
        pool = np.concatenate([rng.uniform(1e-3, 1e3, int(rng.integers(1, 9))), [rng.uniform(1e-3, 1e3)] * 3])
        val = rng.choice(pool, size=(n, 4))
        flat = val.reshape(-1)
        rank = 1 + (flat[None, :] < flat[:, None]).sum(1)  # the kernel's count of strictly smaller metrics
        rank = rank.reshape(n, 4)
        assert rank.max() <= 256 and rank.min() >= 1
        for _ in range(40):
            d = rng.integers(0, 4, n)
            elig = rng.random(n) < (0.0 if trial % 7 == 0 else 0.7)  # every seventh trial: nobody eligible (all metrics 0.0)
            metric = np.where(elig, val[np.arange(n), d], 0.0)
            hm, ha = -1.0, None  # the reference's scan (downlink-nvs-scheduler.cpp:508-527)
            for i in range(n):
                if hm < metric[i]:
                    hm, ha = metric[i], i
            key = (np.where(elig, rank[np.arange(n), d], 0) << 6) | (63 - np.arange(n))
            bk = int(key.max())
            iw = 63 - (bk & 63)
            got_hm = val[iw, d[iw]] if (bk >> 6) != 0 else 0.0
            assert iw == ha and got_hm == hm, (trial, n)
