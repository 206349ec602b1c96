#!/bin/bash
# Round 4, GPU run 12: the listed items of the next TTI packed during the serial phase for every scheduler that holds winners
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_run12; mkdir -p $O; cd $R
python -m pytest tests -m gpu -x -q -k "held or hold or headline or variants or random or experiment or long_run or full_size or specialised_kernels" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -3 $O/pytest.log
ab() { # tag, extra, bench args
  local tag=$1 extra=$2; shift 2
  RS_JIT_EXTRA="$extra" timeout 300 python bench.py --allow-variant --no-cpu-baseline --no-r64 --no-streamed --steps 5 --warmup 1 "$@" > $O/ab_$tag.log 2>&1
  grep -h '^{' $O/ab_$tag.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%-28s %-34s %.3f M TTIs/s  %.3f us' % (sys.argv[1], sys.argv[2], d['value']/1e6, d['us_per_tti_per_cell']))" "$tag" "[$extra]" || tail -3 $O/ab_$tag.log
}
for rep in 1 2 3; do
ab s9_$rep "" --sched 9
ab s9_old_$rep "-DRS_PRELIST_WITH_EWMA_ONLY" --sched 9
done
ab s103 "" --sched 103 --ttis 2000
ab s103_old "-DRS_PRELIST_WITH_EWMA_ONLY" --sched 103 --ttis 2000
ab s101 "" --sched 101 --ttis 2000
ab s101_old "-DRS_PRELIST_WITH_EWMA_ONLY" --sched 101 --ttis 2000
ab s9_u1000 "" --sched 9 --ues-per-slice 50
ab s9_u1000_old "-DRS_PRELIST_WITH_EWMA_ONLY" --sched 9 --ues-per-slice 50
