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
