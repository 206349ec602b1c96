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


def slice_throughput_from_log(lines: Iterable[str], n_users: int, n_slices: int, begin_ts: int = 0,
                              end_ts: int = 10000):
    """plot_throughput.py:26-56 (get_cumubytes + the Mbps conversion): last cumu_bytes of every flow with
    begin_ts < ts <= end_ts, divided by the window in seconds, summed per slice, x 8 / 1e6."""
    cumu_bytes = [0.0] * n_users
    cumu_rbs = [0.0] * n_users
    flow_to_slice = [-1] * n_users
    for line in lines:
        words = line.split(" ")
        if not words[0].isdigit():
            continue
        if int(words[0]) > end_ts:
            break
        if int(words[0]) > begin_ts:
            flow = int(words[2])
            flow_to_slice[flow] = int(words[12])
            cumu_rbs[flow] = int(words[6]) / (end_ts / 1000)
            cumu_bytes[flow] = int(words[4]) / (end_ts / 1000)
    sb = [0.0] * n_slices
    sr = [0.0] * n_slices
    for f in range(n_users):
        if flow_to_slice[f] >= 0:
            sb[flow_to_slice[f]] += cumu_bytes[f]
            sr[flow_to_slice[f]] += cumu_rbs[f]
    return [x * 8 / (1000 * 1000) for x in sb], sr
