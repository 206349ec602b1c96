#!/bin/bash
# Round 4, GPU run 2: the 64-RBG A/B matrix (same box): dead-chunk skip, greedy forms, held winners at 64 RBGs, hand-off threshold.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_run2; mkdir -p $O; cd ..
python -m pytest tests/test_gpu_round4.py tests/test_gpu_parity.py -m gpu -x -q -k "round4 or 64 or random or held or cycling or age" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
ab() { # tag, extra
  RS_JIT_EXTRA="$2" timeout 200 python bench.py --allow-variant --no-cpu-baseline --no-r64 --no-streamed --steps 5 --warmup 1 --ttis 4000 --rbgs 64 --rbg-size 8 > $O/ab_$1.log 2>&1
  grep -h '^{' $O/ab_$1.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%-70s %.3f M TTIs/s  %.2f us' % (sys.argv[1], d['value']/1e6, d['us_per_tti_per_cell']))" "[$2]" || tail -3 $O/ab_$1.log
}
for rep in 1 2; do
ab default_$rep ""
ab noskip_$rep "-DRS_SORT_NO_SKIP"
ab vec_$rep "-DRS_GREEDY_VECTOR"
ab veccoop_$rep "-DRS_GREEDY_VECTOR -DRS_COOP_SCAN"
ab hold_$rep "-DRS_HOLD_ALWAYS"
ab holdvec_$rep "-DRS_HOLD_ALWAYS -DRS_GREEDY_VECTOR"
ab holdvecearly_$rep "-DRS_HOLD_ALWAYS -DRS_GREEDY_VECTOR -DRS_HOLD_EARLY_ALL"
ab fin2_$rep "-DRS_WAVE_FINISH_MAX=2"
ab fin8_$rep "-DRS_WAVE_FINISH_MAX=8"
done
tail -3 $O/pytest.log
