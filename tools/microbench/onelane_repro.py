"""Run from the root of the tree tools/onelane_repro.sh rebuilds (round 4, git 0adb0e5, with SubOpt's hashtable walk back under
`if (lane == 0)`): 12 drop-in SubOpt calls against the oracle.  RS_HIP_LIB selects the library.  See profiles/r05_onelane.md."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import radiosaber_amd as rs
from oracle import oracle_py as oracle
sys.path.insert(0, "tests")
from conftest import synth_cqi
HIST = rs.TRACE_CQI_HISTOGRAM
ues, R, G = [5] * 20, 64, 8
sc = rs.SliceConfig(ues, weight=[0.05] * 20)
U = sc.n_users
ts = rs.TtiScheduler(sc, R, G, sched=101)
cell = oracle.Cell(ues, R, G, 101, weights=[0.05] * 20)
rng = np.random.default_rng(3)
bad = 0
for it in range(12):
    cqi = synth_cqi(100 + it, (U, R), HIST)
    avg = rng.uniform(1e3, 5e6, U)
    if it % 3 == 0:
        avg[:] = 98000.0
    r0, r1 = int(rng.integers(0, 2**31 - 1)), int(rng.integers(0, 2**31 - 1))
    cell.set_cqi(cqi)
    out = cell.new_out()
    assert cell.allocate(avg, r0, r1, out) == 0
    res = ts.schedule_tti(cqi, avg, r0, r1)
    d = np.nonzero(res.rbg_to_user != out.rbg_to_user)[0]
    print(it, "mismatching RBGs:", len(d), "unassigned:", int((res.rbg_to_user < 0).sum()), "quota equal:", bool((res.quota_rbgs == out.quota_rbgs).all()))
    if it < 2:
        print("  quota dev", res.quota_rbgs, "\n  quota ref", out.quota_rbgs, "\n  target dev", res.target_rbs, "\n  target ref", out.target_rbs)
        print("  map dev", res.rbg_to_user, "\n  map ref", out.rbg_to_user)
        print("  nprb dev", res.user_nprb[:40], "\n  nprb ref", out.user_nprb[:40])
    if len(d):
        bad += 1
        print("  dev", res.rbg_to_user[d][:16], "\n  ref", out.rbg_to_user[d][:16])
print("RESULT", os.environ.get("RS_HIP_LIB"), "calls with mismatches:", bad)
