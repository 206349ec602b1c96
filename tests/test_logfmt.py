"""Log-compatible output (N2): the reference's own lines, rebuilt from a decision log."""
import json

import numpy as np

from conftest import GOLDEN
from radiosaber_amd import logfmt

KA = json.loads((GOLDEN / "appendix_a.json").read_text())


def test_reference_log_lines_from_the_oracle_run(oracle, traces):
    cfg = KA["config"]
    c = oracle.Cell(cfg["ues_per_slice"], 64, 8, 9, weights=[cfg["weight"]] * 20)
    logs = c.run_trace(traces["cqi"], traces["mapping"][0], cfg["seed"], cfg["rand_skip"], 200)
    u2s = np.repeat(np.arange(20), 5)
    err = logfmt.stderr_lines(logs["tbs_bits"], logs["rbg_to_user"], u2s, 8)
    # SURVEY.md Appendix A, stderr of the unmodified reference
    assert "100 app: 1 cumu_bytes: 749 cumu_rbs: 8 hol_delay: 0 user: 1 slice: 0" in err
    assert "100 app: 4 cumu_bytes: 1479 cumu_rbs: 16 hol_delay: 0 user: 4 slice: 0" in err
    assert "100 app: 7 cumu_bytes: 2196 cumu_rbs: 24 hol_delay: 0 user: 7 slice: 1" in err
    assert "299 app: 1 cumu_bytes: 88573 cumu_rbs: 1176 hol_delay: 0 user: 1 slice: 0" in err
    assert "299 app: 2 cumu_bytes: 45300 cumu_rbs: 840 hol_delay: 0 user: 2 slice: 0" in err
    assert "299 app: 5 cumu_bytes: 110838 cumu_rbs: 1336 hol_delay: 0 user: 5 slice: 1" in err

    def cqi_of(n, u, r):
        row = 2 + n // 40
        return int(traces["cqi"][traces["mapping"][0][u], row, r])

    out = logfmt.stdout_lines(logs["rbg_to_user"], logs["final_cqi"], logs["target"], logs["quota"], cqi_of)
    assert out[0].startswith("slice_id, target_rbs, quota_rbgs: (0, 25, 3) ")
    assert "(9, 25, 6) " in out[0] and "(16, 37, 4) " in out[0] and out[0].endswith("(19, 25, 3) ")
    assert out[1] == "100"
    assert "User(1) allocated RBGS: 57(15) final_cqi: 15" in out
    assert "User(45) allocated RBGS: 46(7) 51(10) 60(7) final_cqi: 7" in out
    assert "User(47) allocated RBGS: 42(11) 45(8) final_cqi: 8" in out
    # the reducer of plot_throughput.py on those lines == the counters
    mbps, rbs = logfmt.slice_throughput_from_log(err, 100, 20, end_ts=299)
    st = c.state()
    exp = np.add.reduceat(st["cum_bytes"], np.arange(0, 100, 5)) / (299 / 1000) * 8 / 1e6
    np.testing.assert_allclose(mbps, exp, rtol=1e-12)
