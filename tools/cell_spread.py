import sys, numpy as np
sys.path.insert(0, "/root/repo")
import radiosaber_amd as rs
sc = rs.SliceConfig([25] * 20, weight=[0.05] * 20)
cells = 512
b = rs.BatchScheduler(sc, 25, 4, cells, sched=9, jit=True, cqi_epoch_wrap=True)
b.seed((np.arange(cells, dtype=np.uint64) * 2654435761 + 805290992).astype(np.uint32))
b.synthesize_cqi(0x5AB3, 400)
b.prepare_launch(8000)
b.run(8000)
for rep in range(2):
    ms = float(b.run_timed(8000, 1)[0])
    mhz, kms = b.debug_clocks()
    print("launch %.2f ms; cells min %.2f mean %.2f max %.2f" % (ms, kms.min(), kms.mean(), kms.max()))
    print(" first half (cells 0..255) mean %.2f  second half mean %.2f" % (kms[:256].mean(), kms[256:].mean()))
    print(" by cell index mod 8:", np.round([kms[i::8].mean() for i in range(8)], 2))
    print(" histogram:", np.histogram(kms, bins=8)[0], np.round(np.histogram(kms, bins=8)[1], 1))
    pairs = np.stack([kms[:256], kms[256:]])
    print(" pair (c, c+256): mean |diff| %.2f, corr %.2f; mean of pair-max %.2f, pair-min %.2f" % (np.abs(pairs[0]-pairs[1]).mean(), np.corrcoef(pairs)[0,1], pairs.max(0).mean(), pairs.min(0).mean()))
b.close()
