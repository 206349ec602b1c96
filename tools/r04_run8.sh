#!/bin/bash
# Round 4, GPU run 8: where P5 (link adaptation on wave 0) spends its cycles, per scheduler
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_run8; mkdir -p $O; cd $R
export RS_HIP_LIB=radiosaber_amd/libradiosaber_hip_stamps.so
for s in 7 8 1 9; do
  echo "=== sched $s"; RS_JIT_EXTRA="-DRS_STAMPS -DRS_STAMPS_P5" timeout 200 python tools/phase_stamps.py --jit --p5 --sched $s 2>&1 | grep -v "sort \|held winners\|greedy:" | tee $O/p5_s$s.log
done
echo "=== sched 7, 64 RBGs"; RS_JIT_EXTRA="-DRS_STAMPS -DRS_STAMPS_P5" timeout 200 python tools/phase_stamps.py --jit --p5 --sched 7 --rbgs 64 --rbg-size 8 2>&1 | grep -v "sort \|held winners\|greedy:" | tee $O/p5_s7_r64.log
