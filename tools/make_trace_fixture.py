#!/usr/bin/env python3
"""Distil the reference's measured CQI traces into a small DATA fixture.

Runs only in the build container (needs /root/reference/cqi-traces-noise0).
Input  : ue<k>.log, k = 0..157 -- 475 rows x 512 space-separated per-PRB CQIs (1..15);
         mapping<i>.config -- 474 lines "ue_id trace_id"
         (format per enb-mac-entity.cc:38-55,169-191).
Output : tests/golden/cqi_traces_rbg64.npz
           cqi      u8 [158][ROWS][64]  per-RBG CQI (column rbg*8 of the row; the script asserts
                                         that every 8-PRB group of every row is constant, so the
                                         per-RBG value carries the whole row)
           mapping  i32 [4][474]        trace id of ue_id (mapping0..3.config)
           hist     i64 [16]            CQI histogram of the WHOLE corpus (all 475 rows)
ROWS = 40 rows per trace are kept (enough for 1 600 TTIs at CQI_INTERVAL 40).
"""
import sys
from pathlib import Path

import numpy as np

REF = Path("/root/reference/cqi-traces-noise0")
ROOT = Path(__file__).resolve().parents[1]
ROWS = 40
N_TRACE, N_TTI, N_PRB = 158, 475, 512


def main():
    cqi = np.zeros((N_TRACE, ROWS, 64), np.uint8)
    hist = np.zeros(16, np.int64)
    for k in range(N_TRACE):
        a = np.loadtxt(REF / f"ue{k}.log", dtype=np.int16)
        assert a.shape == (N_TTI, N_PRB), (k, a.shape)
        assert a.min() >= 1 and a.max() <= 15
        g = a.reshape(N_TTI, 64, 8)
        assert (g == g[:, :, :1]).all(), f"trace {k}: CQI not constant inside an 8-PRB group"
        hist += np.bincount(a.ravel(), minlength=16)
        cqi[k] = g[:ROWS, :, 0]
    mapping = np.zeros((4, 474), np.int32)
    for i in range(4):
        m = np.loadtxt(REF / f"mapping{i}.config", dtype=np.int32)
        assert m.shape == (474, 2) and (m[:, 0] == np.arange(474)).all()
        mapping[i] = m[:, 1]
    out = ROOT / "tests" / "golden" / "cqi_traces_rbg64.npz"
    np.savez_compressed(out, cqi=cqi, mapping=mapping, hist=hist)
    print("wrote", out, out.stat().st_size, "bytes; hist[1..15] =", hist[1:].tolist())


if __name__ == "__main__":
    sys.exit(main())
