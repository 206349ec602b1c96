"""CPU model of the data-parallel Hoare partition the gfx950 kernel uses to emulate
std::__unguarded_partition (libstdc++ 11 bits/stl_algo.h) one wave-chunk at a time.

Claim checked here (DESIGN.md "MaximizeCell on the device"): with
  L  = positions, left to right, whose key is NOT before the pivot  (key <= pivot: left-scan stops)
  Rr = positions, right to left, whose key the pivot is NOT before  (key >= pivot: right-scan stops)
taken on the array BEFORE the partition, the serial loop performs exactly the swaps (L[j], Rr[j])
for j < k, k = first j with L[j] >= Rr[j], and returns  L[0] if k == 0 else min(L[k], Rr[k-1]).
"""
import numpy as np


def serial_partition(keys, first, last, pivot_key):
    """std::__unguarded_partition(first, last, pivot) with comp(a, b) = key(a) > key(b)."""
    v = keys
    while True:
        while v[first] > pivot_key:
            first += 1
        last -= 1
        while pivot_key > v[last]:
            last -= 1
        if not first < last:
            return first
        v[first], v[last] = v[last], v[first]
        first += 1


def parallel_partition(keys, lo, hi, pivot_key):
    idx = np.arange(lo, hi)
    k = keys[lo:hi]
    L = idx[k <= pivot_key]
    Rr = idx[k >= pivot_key][::-1]
    n = min(len(L), len(Rr))
    sw = L[:n] < Rr[:n]
    kk = int(np.argmin(sw)) if not sw.all() else n
    assert sw[:kk].all() and not sw[kk:].any()  # monotone
    a, b = L[:kk], Rr[:kk]
    keys[a], keys[b] = keys[b].copy(), keys[a].copy()
    if kk == 0:
        return int(L[0])
    lk = int(L[kk]) if kk < len(L) else 1 << 30
    return min(lk, int(Rr[kk - 1]))


def median_to_first(v, result, a, b, c):
    def before(x, y):
        return v[x] > v[y]
    if before(a, b):
        pick = b if before(b, c) else (c if before(a, c) else a)
    elif before(a, c):
        pick = a
    elif before(b, c):
        pick = c
    else:
        pick = b
    v[result], v[pick] = v[pick], v[result]


def test_parallel_partition_equals_serial():
    rng = np.random.default_rng(5)
    for trial in range(4000):
        n = int(rng.integers(17, 700))
        levels = int(rng.integers(1, 17))
        keys = rng.integers(0, levels, n).astype(np.int64)
        if trial % 7 == 0:
            keys.sort()
        if trial % 11 == 0:
            keys = keys[::-1].copy()
        a = keys.copy()
        median_to_first(a, 0, 1, n // 2, n - 1)
        b = a.copy()
        cut_s = serial_partition(a, 1, n, a[0])
        cut_p = parallel_partition(b, 1, n, b[0])
        assert cut_s == cut_p, (trial, n, cut_s, cut_p)
        assert (a == b).all(), trial


def local_rule_partition(keys, lo, hi, pivot_key):
    """The rule the register sort uses: every position decides from two stop counts.
    A(x) = A-stops (key <= pivot) in [lo, x), B(x) = B-stops (key >= pivot) in (x, hi):
    an A-stop is swapped iff B > A (takes B-stop number A from the right), a B-stop iff A > B;
    cut = leftmost unswapped A-stop or swapped B-stop."""
    k = keys[lo:hi].copy()
    n = hi - lo
    isA = k <= pivot_key
    isB = k >= pivot_key
    A = np.concatenate([[0], np.cumsum(isA)[:-1]])
    B = np.concatenate([np.cumsum(isB[::-1])[::-1][1:], [0]])
    swA = isA & (B > A)
    swB = isB & ~swA & (A > B)
    assert not (swA & swB).any()
    xbuf = np.full(n + 1, -1, dtype=np.int64)
    f = -1  # slots relative to f = lo - 1: A-stop a -> f + a (index a), B-stop b -> l - 1 - b
    slot = np.full(n, -1)
    slot[swA] = A[swA]
    slot[swB] = (n - 1) - B[swB]
    assert len(np.unique(slot[slot >= 0])) == (slot >= 0).sum()
    xbuf[slot[slot >= 0]] = k[slot >= 0]
    out = k.copy()
    sw = slot >= 0
    out[sw] = xbuf[(n - 1) - slot[sw]]
    assert (out[sw] >= 0).all()
    keys[lo:hi] = out
    cand = np.where(isA, ~swA, swB)
    return lo + int(np.argmax(cand)) if cand.any() else None


def test_local_rule_equals_serial():
    rng = np.random.default_rng(6)
    for trial in range(4000):
        n = int(rng.integers(17, 700))
        levels = int(rng.integers(1, 17))
        keys = rng.integers(0, levels, n).astype(np.int64)
        if trial % 7 == 0:
            keys.sort()
        if trial % 11 == 0:
            keys = keys[::-1].copy()
        a = keys.copy()
        median_to_first(a, 0, 1, n // 2, n - 1)
        b = a.copy()
        cut_s = serial_partition(a, 1, n, a[0])
        cut_p = local_rule_partition(b, 1, n, b[0])
        assert cut_s == cut_p, (trial, n, cut_s, cut_p)
        assert (a == b).all(), trial


# ---- round 6: the two shortcuts the level body takes (rs_sort_device.h, profiles/r06_sort_staged.md) ----

def _stops_and_swaps(k, pivot_key):
    isA = k <= pivot_key
    isB = k >= pivot_key
    A = np.concatenate([[0], np.cumsum(isA)[:-1]])
    B = np.concatenate([np.cumsum(isB[::-1])[::-1][1:], [0]])
    swA = isA & (B > A)
    swB = isB & ~swA & (A > B)
    return isA, isB, A, B, swA, swB


def test_candidates_are_exactly_the_positions_from_the_cut_on():
    """The cut is reported by the ONE candidate whose left neighbour is not a candidate (a DPP shift instead of a range mask): that
    needs the candidates -- unswapped A-stops and swapped B-stops -- to be a contiguous suffix [cut, l) of the sub-range.  The
    single-wave finish relies on the same fact when it takes the first candidate at or above the piece's first inner lane."""
    rng = np.random.default_rng(61)
    for trial in range(6000):
        n = int(rng.integers(17, 300))
        levels = int(rng.integers(1, 17))
        keys = rng.integers(0, levels, n).astype(np.int64)
        if trial % 7 == 0:
            keys.sort()
        if trial % 11 == 0:
            keys = keys[::-1].copy()
        a = keys.copy()
        median_to_first(a, 0, 1, n // 2, n - 1)
        k = a[1:n]
        isA, isB, A, B, swA, swB = _stops_and_swaps(k, a[0])
        cand = np.where(isA, ~swA, swB)
        ser = a.copy()
        cut = serial_partition(ser, 1, n, ser[0]) - 1  # relative to the first inner position
        assert cand.any() and int(np.argmax(cand)) == cut, trial
        assert cand[cut:].all() and not cand[:cut].any(), trial
        # the reporting rule: a candidate whose left neighbour (the pivot slot for the first inner position) is not one
        prev = np.concatenate([[False], cand[:-1]])
        assert (cand & ~prev).sum() == 1 and int(np.argmax(cand & ~prev)) == cut, trial


def test_stop_counts_from_chunk_ranks_and_chunk_totals():
    """R takes A(x), B(x) from what every position wrote down in F -- A-stops of its 64-position chunk below it, B-stops up to and
    including it -- plus one prefix over the chunk totals:  A(x) = P_A(x) - P_A(lo),  B(x) = P_Bincl(hi - 1) - P_Bincl(x), whatever
    other sub-ranges share the chunks (their stops cancel in the differences)."""
    rng = np.random.default_rng(62)
    for trial in range(300):
        N = int(rng.integers(200, 1400))
        # a level's worth of sub-ranges: random cut points, some retired (no stops at all), pivots per sub-range
        cuts = np.unique(np.concatenate([[0, N], rng.integers(1, N, int(rng.integers(1, 40)))]))
        key = rng.integers(0, int(rng.integers(1, 17)), N)
        isA = np.zeros(N, bool)
        isB = np.zeros(N, bool)
        ranges = []
        for f, l in zip(cuts[:-1], cuts[1:]):
            if l - f <= 16 or rng.random() < 0.2:
                continue  # retired: contributes no stops
            pk = key[f]
            isA[f + 1:l] = key[f + 1:l] <= pk
            isB[f + 1:l] = key[f + 1:l] >= pk
            ranges.append((f, l))
        chunk = np.arange(N) >> 6
        n_chunks = int(chunk[-1]) + 1
        rA = np.array([isA[(c << 6):x].sum() for x, c in enumerate(chunk)])          # below me in my chunk
        rB = np.array([isB[(c << 6):x + 1].sum() for x, c in enumerate(chunk)])      # up to and including me
        totA = np.array([isA[c << 6:(c + 1) << 6].sum() for c in range(n_chunks)])
        totB = np.array([isB[c << 6:(c + 1) << 6].sum() for c in range(n_chunks)])
        preA = np.concatenate([[0], np.cumsum(totA)[:-1]])
        preB = np.concatenate([[0], np.cumsum(totB)[:-1]])
        PA = preA[chunk] + rA
        PB = preB[chunk] + rB
        for f, l in ranges:
            lo, hm = f + 1, l - 1
            for x in range(lo, l):
                a = PA[x] - PA[lo]
                b = PB[hm] - PB[x]
                assert a == isA[lo:x].sum() and b == isB[x + 1:l].sum(), (trial, f, l, x)


def test_greedy_fixed_point_from_any_start():
    """MaximizeCell's vector scan iterates  T -> F(T),  F(T)(i) = live(i) and no T(k<i) has i's RBG and fewer than left[slice(i)]
    T(k<i) have i's slice.  Round 6 starts it from "first live record of its RBG" instead of T = live: F has one fixed point -- the
    serial scan's answer -- and the iteration reaches it from ANY start (position 0 is final after one round, position 1 after two...)."""
    rng = np.random.default_rng(63)
    for trial in range(400):
        R, S = int(rng.integers(2, 33)), int(rng.integers(2, 33))
        n = 64
        rbg = rng.integers(0, R, n)
        sl = rng.integers(0, S, n)
        left = rng.integers(0, 4, S)
        live = rng.random(n) < 0.8
        # the serial scan
        want = np.zeros(n, bool)
        free = np.ones(R, bool)
        q = left.copy()
        for i in range(n):
            if live[i] and free[rbg[i]] and q[sl[i]] > 0:
                want[i] = True
                free[rbg[i]] = False
                q[sl[i]] -= 1

        def F(T):
            out = np.zeros(n, bool)
            for i in range(n):
                before = T[:i]
                out[i] = live[i] and not (before & (rbg[:i] == rbg[i])).any() and (before & (sl[:i] == sl[i])).sum() < left[sl[i]]
            return out

        first_of_rbg = np.array([live[i] and not (live[:i] & (rbg[:i] == rbg[i])).any() for i in range(n)])
        for start in (live.copy(), first_of_rbg, np.zeros(n, bool), rng.random(n) < 0.5):
            T = start
            for rounds in range(n + 2):
                Tn = F(T)
                if (Tn == T).all():
                    break
                T = Tn
            assert (T == want).all(), trial
