#!/bin/bash
# Round 4, GPU run 19: phase stamps of the streamed-CQI mode (cqi_refresh = 1) beside the resident mode, sched 9 and 8
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_run19; mkdir -p $O; cd ..; rm -f $O/stamps.log; export RS_HIP_LIB=radiosaber_amd/libradiosaber_hip_stamps.so
for s in 9 8; do for rf in 40 1; do
echo "=== sched $s refresh $rf" >> $O/stamps.log
RS_JIT_EXTRA="-DRS_STAMPS" timeout 200 python tools/phase_stamps.py --jit --sched $s --cqi-refresh $rf >> $O/stamps.log 2>&1
done; done
cat $O/stamps.log
