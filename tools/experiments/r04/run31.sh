#!/bin/bash
# Round 4, GPU run 31: the fetch-ahead code only in the kernels of streamed batches -- resident numbers back, streamed numbers kept; whole suite
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_run31; mkdir -p $O; cd ..
ab() { local tag=$1 extra=$2; shift 2
  RS_JIT_EXTRA="$extra" timeout 300 python bench.py --allow-variant --no-cpu-baseline --no-r64 --no-streamed --steps 5 --warmup 1 "$@" > $O/ab_$tag.log 2>&1
  grep -h '^{' $O/ab_$tag.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%-28s %-24s %.2f M TTIs/s  %.3f us' % (sys.argv[1], sys.argv[2], d['value']/1e6, d['us_per_tti_per_cell']))" "$tag" "[$extra]" || tail -3 $O/ab_$tag.log
}
ab s8_r64 "" --sched 8 --ttis 4000 --rbgs 64 --rbg-size 8
ab s7_r64 "" --sched 7 --ttis 4000 --rbgs 64 --rbg-size 8
ab s7_r25 "" --sched 7 --ttis 4000
ab s8_r25 "" --sched 8 --ttis 4000
ab s7_stream "" --sched 7 --ttis 2000 --cqi-refresh 1
ab s8_stream "" --sched 8 --ttis 2000 --cqi-refresh 1
ab s1_stream "" --sched 1 --ttis 2000 --cqi-refresh 1
python -m pytest tests -m gpu -q > $O/pytest_all.log 2>&1; echo "pytest rc $?" >> $O/pytest_all.log
grep -n "FAILED\|passed\|failed\|rc " $O/pytest_all.log | tail -8
