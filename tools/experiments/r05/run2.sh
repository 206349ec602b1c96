#!/bin/bash
# Round 5, GPU run 2: the rocprofv3-vs-plain gap probe; the shipped shapes again on the changed rules (+ one autotuned run per
# scheduler 8 / 9 shape); the rewritten counting sort at one chunk per wave (A/B, headline shape).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_run2; mkdir -p $O; cd $R
bash tools/experiments/r05/run_gap.sh 2>&1 | tee $O/gap_summary.log
cd $R
python tools/sweep_shipped_shapes.py --autotune --out gpurun_out/r05_shipped_after.json > $O/shipped_after.log 2>&1; tail -4 $O/shipped_after.log
for v in "" "-DRS_COUNTING_SORT_V2_MIN_CPW=1" "" "-DRS_COUNTING_SORT_V2_MIN_CPW=1"; do
  RS_JIT_EXTRA="$v" python bench.py --allow-variant --no-cpu-baseline --no-streamed --no-r64 --steps 8 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().split('\n')[-1])
print('counting sort at one chunk per wave [%s]: %.2f M' % (sys.argv[1], d['value'] / 1e6))" "$v" | tee -a $O/counting_cpw1_ab.log
done
