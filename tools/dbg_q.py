import json, sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import radiosaber_amd as rs
from oracle import oracle_py as oracle
from conftest import GOLDEN, synth_cqi
HIST = rs.TRACE_CQI_HISTOGRAM
cfg = json.loads((GOLDEN / "experiment_configs.json").read_text())["exp-customization/exp-customize-20slices/config.json"]
sc = rs.SliceConfig(cfg["ues_per_slice"], cfg["weight"], cfg["algo_alpha"], cfg["algo_beta"], cfg["algo_epsilon"], cfg["algo_psi"], cfg["traffic"])
kinds = sc.bearer_kinds(); U, u2s = sc.n_users, sc.user_to_slice
video = json.loads((GOLDEN / "video_foreman_1280k.json").read_text())
n_cells, n_ttis, R, G = 1, 240, 64, 8
stop = 0.1 + n_ttis / 1000.0 + 0.01
bursts = {}
for c in range(n_cells):
    for u in range(U):
        tr = cfg["traffic"][u2s[u]]
        for j in range(int(tr["internet_flow"])):
            rate = tr["if_bitrate"][j] / cfg["ues_per_slice"][u2s[u]]
            bursts[(c, u, j)] = rs.internet_flow_arrivals(rate, 0.1, stop, 1000 * c + 2 * u + j)
        if int(tr["video_app"]):
            t, ts = 0.1, []
            for k in range(len(video["bytes"])):
                if k: t = (video["time_ms"][k] - video["time_ms"][k - 1]) * 0.001 + t
                if t >= stop: break
                ts.append(t)
            bursts[(c, u, 0)] = rs.frames_to_bursts(ts, video["bytes"][:len(ts)])
grids = synth_cqi(77, (n_cells, (n_ttis + 39) // 40, U, R), HIST)
seeds = np.array([5], np.uint32)
for jit in (False, True):
    b = rs.BatchScheduler(sc, R, G, n_cells, sched=9, jit=jit)
    b.set_bearers(kinds); b.set_arrivals(bursts); b.seed(seeds); b.upload_cqi_epochs(grids)
    got = b.run_logged(n_ttis); bst = b.bearer_state(); b.close()
    cell = oracle.Cell(cfg["ues_per_slice"], R, G, 9, weights=cfg["weight"], alpha=cfg["algo_alpha"], beta=cfg["algo_beta"], epsilon=cfg["algo_epsilon"], psi=cfg["algo_psi"])
    cell.enable_queues(kinds)
    for (cc, u, k), (t, nf, la) in bursts.items(): cell.set_arrivals(u, k, t, nf, la)
    logs = cell.run_synth_queues(grids[0], 5, n_ttis)
    bad = np.flatnonzero((got["rbg_to_user"][0] != logs["rbg_to_user"]).any(1))
    print("jit", jit, "differing TTIs:", bad[:10], len(bad))
    if len(bad):
        n = bad[0]
        d = np.flatnonzero(got["rbg_to_user"][0][n] != logs["rbg_to_user"][n])
        print(" TTI", n, "rbgs", d[:10], "dev", got["rbg_to_user"][0][n][d][:10], "ora", logs["rbg_to_user"][n][d][:10])
        for u in set(got["rbg_to_user"][0][n][d][:4].tolist() + logs["rbg_to_user"][n][d][:4].tolist()):
            if u >= 0: print("  user", u, "slice", u2s[u], "alpha/beta", cfg["algo_alpha"][u2s[u]], cfg["algo_beta"][u2s[u]], "kinds", kinds[u], cfg["traffic"][u2s[u]])
        print(" quota dev", got["quota"][0][n].tolist())
        print(" target dev", got["target"][0][n].tolist())
