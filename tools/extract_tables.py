#!/usr/bin/env python3
"""Capture the AMC constant tables of the reference as DATA.

Runs only in the build container (needs /root/reference).  It parses the numeric
initialisers of four arrays in src/protocolStack/mac/AMCModule.cpp
(MapCQIToMCS :36-40, SINRForCQIIndex :96-100, McsToItbs :114-117,
TransportBlockSizeTable :120-231 = 3GPP TS 36.213 Table 7.1.7.2.1-1) and writes

  radiosaber_amd/data/amc_tables.npz      binary arrays; radiosaber_amd/build.py turns them into the generated
                                          (git-ignored) header csrc/rs_amc_tables.inc at build time
  tests/golden/amc_tables.npz             the same numbers for the python tests

No reference source text is copied: only the numeric values (3GPP-standard facts).
"""
import re
import sys
from pathlib import Path

REF = Path("/root/reference/src/protocolStack/mac/AMCModule.cpp")
ROOT = Path(__file__).resolve().parents[1]


def strip_comments(txt: str) -> str:
    txt = re.sub(r"/\*.*?\*/", " ", txt, flags=re.S)
    txt = re.sub(r"//[^\n]*", " ", txt)
    return txt


def grab(txt: str, decl: str):
    """Return the flat list of numeric tokens of the initialiser following `decl`."""
    i = txt.index(decl)
    j = txt.index("=", i)
    k = txt.index(";", j)
    body = txt[j + 1:k]
    return re.findall(r"-?\d+\.\d+|-?\d+", body)


def main():
    raw = REF.read_text()
    # the file contains a large commented-out block with older SINR tables: drop comments first
    txt = strip_comments(raw)
    cqi_to_mcs = [int(x) for x in grab(txt, "int MapCQIToMCS[15]")]
    sinr = [float(x) for x in grab(txt, "double SINRForCQIIndex[15]")]
    mcs_to_itbs = [int(x) for x in grab(txt, "int McsToItbs[29]")]
    tbs = [int(x) for x in grab(txt, "int TransportBlockSizeTable [110][27]")]
    assert len(cqi_to_mcs) == 15 and len(sinr) == 15 and len(mcs_to_itbs) == 29
    assert len(tbs) == 110 * 27, len(tbs)
    sinr_txt = grab(txt, "double SINRForCQIIndex[15]")

    import numpy as np
    payload = dict(tbs=np.array(tbs, np.int32).reshape(110, 27), mcs_to_itbs=np.array(mcs_to_itbs, np.int32),
                   cqi_to_mcs=np.array(cqi_to_mcs, np.int32), sinr_for_cqi=np.array(sinr, np.float64))
    out = ROOT / "radiosaber_amd" / "data" / "amc_tables.npz"
    out.parent.mkdir(exist_ok=True)
    np.savez_compressed(out, **payload)
    gold = ROOT / "tests" / "golden" / "amc_tables.npz"
    np.savez_compressed(gold, **payload)
    print("wrote", out, gold)


if __name__ == "__main__":
    sys.exit(main())
