#!/bin/bash
# Round 5: evidence under (or against) round 4's "each launch starts on a drained GPU" explanation of rocprofv3 seeing the cell kernel
# 2-3 % faster than a plain run.  One lease: plain / profiled / plain / plain with a host sleep between launches; every launch reports
# the shader clock it ran at from the kernel's own two clocks; one --pmc pass with GRBM_GUI_ACTIVE GRBM_COUNT.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_gap; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocm-smi --showclocks --json > $O/smi_before.json 2>&1
python3 $R/tools/gap_probe.py --tag plain1 > $O/plain1.json 2> $O/plain1.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_kt -- python3 $R/tools/gap_probe.py --tag profiled > $O/profiled.json 2> $O/profiled.err
python3 $R/tools/gap_probe.py --tag plain2 > $O/plain2.json 2> $O/plain2.err
python3 $R/tools/gap_probe.py --tag sleep1ms --sleep-ms 1 > $O/sleep1.json 2> $O/sleep1.err
python3 $R/tools/gap_probe.py --tag sleep20ms --sleep-ms 20 > $O/sleep20.json 2> $O/sleep20.err
python3 $R/tools/gap_probe.py --tag sleep200ms --sleep-ms 200 --launches 8 > $O/sleep200.json 2> $O/sleep200.err
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $O/prof_pmc -- python3 $R/tools/gap_probe.py --tag pmc --launches 6 > $O/pmc.json 2> $O/pmc.err
rocm-smi --showclocks --json > $O/smi_after.json 2>&1
for f in plain1 profiled plain2 sleep1 sleep20 sleep200 pmc; do python3 -c "
import json,sys
d=json.loads(open('$O/$f.json').read().strip().split('\n')[-1])
print('%-10s event %.3f ms (min %.3f max %.3f)  shader %.1f MHz (min %.1f)  %.2f M cycles per launch' % (d['tag'], d['event_ms_mean'], d['event_ms_min'], d['event_ms_max'], d['shader_mhz_mean'], d['shader_mhz_min'], d['cycles_per_launch_mean_M']))"; done
find $O/prof_kt -name "*kernel_stats.csv" | head -1 | xargs cat | cut -c1-160 | head -5
