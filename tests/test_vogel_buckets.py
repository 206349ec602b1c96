"""CPU model of the device's VogelApproximate searches (rs_interslice.h: vogel_best_second): the walk over key values on four
bit planes gives exactly the reference's sequential best / "second" / arg-best scan (downlink-transport-scheduler.cpp:395-431:
a new best does not demote the old best to second), on random rows incl. exact ties, holes and single elements."""
import random


def sequential(keys, allowed):
    best = second = arg = -1
    for i, key in enumerate(keys):
        if not (allowed >> i) & 1:
            continue
        if best < 0 or key > best:
            best, arg = key, i
        elif second < 0 or key > second:
            second = key
    return best, second, arg


def planes_walk(keys, allowed):
    n = len(keys)
    p = [sum(((k >> b) & 1) << i for i, k in enumerate(keys)) for b in range(4)]
    full = (1 << n) - 1
    cand = full & allowed
    seen = 0
    best = second = arg = -1
    for v in range(15, -1, -1):
        if cand == 0 or (best >= 0 and second >= 0) or seen == cand:
            break
        a = cand
        for b in range(4):
            a &= p[b] if (v >> b) & 1 else ~p[b] & full
        seen |= a
        first = seen & -seen
        if a and best < 0:
            best, arg = v, (a & -a).bit_length() - 1
        if (a & ~first) and second < 0:
            second = v
    return best, second, arg


def test_bit_plane_walk_is_the_sequential_scan():
    rng = random.Random(103)
    for _ in range(60000):
        n = rng.randint(1, 64)
        lo = rng.randint(0, 15)
        hi = rng.randint(lo, 15)
        keys = [rng.randint(lo, hi) for _ in range(n)]
        allowed = rng.getrandbits(n) if rng.random() < 0.8 else (1 << n) - 1
        assert planes_walk(keys, allowed) == sequential(keys, allowed), (keys, bin(allowed))
