#!/bin/bash
# Round 4, GPU run 16: phase stamps of sched 1 with / without kPf1 (wave 0's clock)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_run16; mkdir -p $O; cd ..; rm -f $O/stamps.log; export RS_HIP_LIB=radiosaber_amd/libradiosaber_hip_stamps.so
for v in "" "-DRS_NO_PF1_LANES"; do
for args in "" "--ues-per-slice 50" "--rbgs 64 --rbg-size 8"; do
echo "=== [$v] $args" >> $O/stamps.log
RS_JIT_EXTRA="-DRS_STAMPS $v" timeout 200 python tools/phase_stamps.py --jit --sched 1 $args >> $O/stamps.log 2>&1
done; done
cat $O/stamps.log
