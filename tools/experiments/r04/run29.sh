#!/bin/bash
# Round 4, GPU run 29: users per stage-1 block (RS_P3_BLOCK) for the schedulers that scan every TTI
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_run29; mkdir -p $O; cd ..
ab() { local tag=$1 extra=$2; shift 2
  RS_JIT_EXTRA="$extra" timeout 300 python bench.py --allow-variant --no-cpu-baseline --no-r64 --no-streamed --steps 5 --warmup 1 "$@" > $O/ab_$tag.log 2>&1
  grep -h '^{' $O/ab_$tag.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%-28s %-36s %.2f M TTIs/s  %.3f us' % (sys.argv[1], sys.argv[2], d['value']/1e6, d['us_per_tti_per_cell']))" "$tag" "[$extra]" || tail -3 $O/ab_$tag.log
}
for v in "" "-URS_P3_BLOCK -DRS_P3_BLOCK=16" "-URS_P3_BLOCK -DRS_P3_BLOCK=8"; do
ab s1_r25 "$v" --sched 1 --ttis 4000
ab s1_r64 "$v" --sched 1 --ttis 4000 --rbgs 64 --rbg-size 8
ab s7_u1000 "$v" --sched 7 --ttis 4000 --ues-per-slice 50
ab s8_r64 "$v" --sched 8 --ttis 4000 --rbgs 64 --rbg-size 8
ab s10_r25 "$v" --sched 10 --ttis 2000
done
