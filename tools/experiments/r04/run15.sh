#!/bin/bash
# Round 4, GPU run 15: per-flow PF (sched 1) scanned one wave per RBG over all users (kPf1) -- parity, then same-box A/B against -DRS_NO_PF1_LANES
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_run15; mkdir -p $O; cd ..
python -m pytest tests -m gpu -x -q -k "sched or sweep or 1000 or random or config or early or prepar" > $O/pytest_s1.log 2>&1; echo "pytest rc $?" >> $O/pytest_s1.log
tail -5 $O/pytest_s1.log
ab() { # tag, extra, bench args
  local tag=$1 extra=$2; shift 2
  RS_JIT_EXTRA="$extra" timeout 200 python bench.py --allow-variant --no-cpu-baseline --no-r64 --no-streamed --steps 5 --warmup 1 --ttis 4000 "$@" > $O/ab_$tag.log 2>&1
  grep -h '^{' $O/ab_$tag.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%-28s %-20s %.2f M TTIs/s  %.3f us' % (sys.argv[1], sys.argv[2], d['value']/1e6, d['us_per_tti_per_cell']))" "$tag" "[$extra]" || tail -3 $O/ab_$tag.log
}
for rep in 1 2; do
ab s1_r25_new_$rep "" --sched 1
ab s1_r25_base_$rep "-DRS_NO_PF1_LANES" --sched 1
ab s1_u1000_new_$rep "" --sched 1 --ues-per-slice 50
ab s1_u1000_base_$rep "-DRS_NO_PF1_LANES" --sched 1 --ues-per-slice 50
ab s1_r64_new_$rep "" --sched 1 --rbgs 64 --rbg-size 8
ab s1_r64_base_$rep "-DRS_NO_PF1_LANES" --sched 1 --rbgs 64 --rbg-size 8
done
ab s1_u1000_new_noearly "-DRS_NO_EARLY17" --sched 1 --ues-per-slice 50
python -m pytest tests -m gpu -x -q > $O/pytest_all.log 2>&1; echo "pytest rc $?" >> $O/pytest_all.log
tail -4 $O/pytest_all.log
