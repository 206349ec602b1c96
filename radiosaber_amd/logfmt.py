"""Log-compatible text output and the per-slice throughput reducer.

The reference's experiments are evaluated by parsing the simulator's output streams:

  stderr, one line per (TTI, transmitting bearer), DoStopSchedule
      (downlink-transport-scheduler.cpp:192-199, same in the NVS and PF schedulers):
      "<ts> app: <A> cumu_bytes: <B> cumu_rbs: <K> hol_delay: <H> user: <U> slice: <S>"
  stdout, per TTI, RBsAllocation (downlink-transport-scheduler.cpp:523-527, 631-649):
      "slice_id, target_rbs, quota_rbgs: (0, t, q) (1, t, q) ... "
      "<ts>"
      "User(<id>) allocated RBGS: <rbg>(<cqi>) ... final_cqi: <c>"

These helpers rebuild those lines from the per-TTI decision log of one cell (BatchScheduler.run_logged
or the oracle) so the reference's own scripts (NSDI23-radiosaber-experiments/*/plot_*.py) keep working,
and reimplement the reducer of plot_throughput.py:26-56.  InfiniteBuffer bearers have an empty MAC
queue, so hol_delay prints as 0 (src/flows/radio-bearer.cpp:281-289); app id == user id for the
one-bearer-per-UE backlogged configs.
"""
from typing import Iterable, List, Sequence

import numpy as np


def stderr_lines(tbs_bits, rbg_to_user, user_to_slice: Sequence[int], rbg_size: int, first_ts: int = 100,
                 cum_bytes0=None, cum_rbs0=None, nprb=None, pf_format: bool = False) -> List[str]:
    """tbs_bits [n_ttis][U], rbg_to_user [n_ttis][R] of ONE cell -> the reference's stderr lines.
    nprb [n_ttis][U] (run_logged's "nprb"): the per-user PRB counts, needed for UpperBound where several users hold one
    RBG; derived from rbg_to_user otherwise.  pf_format: the "flow:" line of the PF scheduler
    (downlink-packet-scheduler.cpp:140-145) instead of the "app: .. user: .. slice: .." line."""
    tbs_bits = np.asarray(tbs_bits)
    rbg_to_user = np.asarray(rbg_to_user)
    n_ttis, U = tbs_bits.shape
    cb = np.zeros(U, np.int64) if cum_bytes0 is None else np.array(cum_bytes0, np.int64)
    cr = np.zeros(U, np.int64) if cum_rbs0 is None else np.array(cum_rbs0, np.int64)
    out = []
    for n in range(n_ttis):
        prbs = np.asarray(nprb[n]) if nprb is not None else \
            np.bincount(rbg_to_user[n][rbg_to_user[n] >= 0], minlength=U) * rbg_size
        for u in np.flatnonzero(tbs_bits[n] // 8 > 0):
            cb[u] += min(int(tbs_bits[n, u]) // 8, 100000000)
            cr[u] += int(prbs[u])
            if pf_format:
                out.append(f"{first_ts + n} flow: {u} cumu_bytes: {cb[u]} cumu_rbs: {cr[u]} hol_delay: 0")
            else:
                out.append(f"{first_ts + n} app: {u} cumu_bytes: {cb[u]} cumu_rbs: {cr[u]} hol_delay: 0 "
                           f"user: {u} slice: {user_to_slice[u]}")
    return out


def stdout_lines(rbg_to_user, final_cqi, target, quota, cqi_of, first_ts: int = 100, transport: bool = True) -> List[str]:
    """The reference's allocation map.  cqi_of(n, user, rbg) -> CQI the user reported on that RBG."""
    rbg_to_user = np.asarray(rbg_to_user)
    n_ttis, R = rbg_to_user.shape
    out = []
    for n in range(n_ttis):
        if transport:
            out.append("slice_id, target_rbs, quota_rbgs: " +
                       "".join(f"({i}, {int(target[n][i])}, {int(quota[n][i])}) " for i in range(len(quota[n]))))
        out.append(str(first_ts + n))
        for u in np.unique(rbg_to_user[n][rbg_to_user[n] >= 0]):
            rb = np.flatnonzero(rbg_to_user[n] == u)
            out.append(f"User({u}) allocated RBGS:" + "".join(f" {r}({cqi_of(n, int(u), int(r))})" for r in rb) +
                       f" final_cqi: {int(final_cqi[n][u])}")
    return out


def _parse_counter_lines(lines: Iterable[str]):
    """The numeric columns of every counter line, looked up by their labels: a line is "<ts> app: <A> cumu_bytes: <B> cumu_rbs: <K>
    hol_delay: <H> user: <U> slice: <S>"; anything that does not start with a TTI stamp or lacks a label is not a counter line."""
    want = ("app:", "cumu_bytes:", "cumu_rbs:", "slice:")
    rows = []
    for line in lines:
        tok = line.split()
        if not tok or not tok[0].isdigit():
            continue
        try:
            rows.append([int(tok[0])] + [int(tok[tok.index(k) + 1]) for k in want])
        except (ValueError, IndexError):
            continue
    return np.asarray(rows, np.int64).reshape(-1, 1 + len(want))


def slice_throughput_from_log(lines: Iterable[str], n_users: int, n_slices: int, begin_ts: int = 0,
                              end_ts: int = 10000):
    """Per-slice throughput of one run's stderr, as the reference's evaluation defines it (what plot_throughput.py:26-56 computes):
    for every flow the LAST cumu_bytes / cumu_rbs it printed at a stamp in (begin_ts, end_ts], over end_ts milliseconds in seconds,
    summed over the flows of a slice; bytes as Mbit/s (x 8 / 1e6).  A flow that printed nothing in the window counts for nothing.
    Returns (mbps[n_slices], rbs_per_s[n_slices]).  The stamps of a log ascend, so "the last line" is the row with the largest index."""
    tab = _parse_counter_lines(lines)
    mbps, rbs = np.zeros(n_slices), np.zeros(n_slices)
    if tab.size:
        ts, flow = tab[:, 0], tab[:, 1]
        tab = tab[(ts > begin_ts) & (ts <= end_ts) & (flow >= 0) & (flow < n_users)]
    if tab.size:
        # the last row of each flow: first occurrence in the reversed table
        flows, first_rev = np.unique(tab[::-1, 1], return_index=True)
        last = tab[len(tab) - 1 - first_rev]
        seconds = end_ts / 1000
        np.add.at(mbps, last[:, 4], last[:, 2] / seconds)
        np.add.at(rbs, last[:, 4], last[:, 3] / seconds)
    return list(mbps * 8 / 1e6), list(rbs)
