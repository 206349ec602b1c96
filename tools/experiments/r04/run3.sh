#!/bin/bash
# Round 4, GPU run 3: schedulers 1 and 7 with the next TTI prepared beside wave 0 -- parity, then same-box A/B against -DRS_NO_EARLY17
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_run3; mkdir -p $O; cd ..
python -m pytest tests/test_gpu_round4.py -m gpu -x -q -k "prepare or early_preparation" > $O/pytest_early.log 2>&1; echo "pytest rc $?" >> $O/pytest_early.log
tail -5 $O/pytest_early.log
ab() { # tag, extra, bench args
  local tag=$1 extra=$2; shift 2
  RS_JIT_EXTRA="$extra" timeout 200 python bench.py --allow-variant --no-cpu-baseline --no-r64 --no-streamed --steps 5 --warmup 1 --ttis 4000 "$@" > $O/ab_$tag.log 2>&1
  grep -h '^{' $O/ab_$tag.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%-28s %-20s %.2f M TTIs/s  %.3f us' % (sys.argv[1], sys.argv[2], d['value']/1e6, d['us_per_tti_per_cell']))" "$tag" "[$extra]" || tail -3 $O/ab_$tag.log
}
for rep in 1 2; do
for s in 7 1; do
ab s${s}_r25_early_$rep "" --sched $s
ab s${s}_r25_base_$rep "-DRS_NO_EARLY17" --sched $s
ab s${s}_u1000_early_$rep "" --sched $s --ues-per-slice 50
ab s${s}_u1000_base_$rep "-DRS_NO_EARLY17" --sched $s --ues-per-slice 50
ab s${s}_r64_early_$rep "" --sched $s --rbgs 64 --rbg-size 8
ab s${s}_r64_base_$rep "-DRS_NO_EARLY17" --sched $s --rbgs 64 --rbg-size 8
done
done
python -m pytest tests -m gpu -x -q > $O/pytest_all.log 2>&1; echo "pytest rc $?" >> $O/pytest_all.log
tail -4 $O/pytest_all.log
