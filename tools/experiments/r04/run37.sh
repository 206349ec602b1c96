#!/bin/bash
# Round 4, GPU run 37: earlier shape rules re-measured on the lean build (lower register pressure)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_run37; mkdir -p $O; cd ..
ab() { local tag=$1 extra=$2; shift 2
  RS_JIT_EXTRA="$extra" timeout 300 python bench.py --allow-variant --no-cpu-baseline --no-r64 --no-streamed --steps 5 --warmup 1 "$@" > $O/ab_$tag.log 2>&1
  grep -h '^{' $O/ab_$tag.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%-20s %-44s %.2f M TTIs/s  %.3f us' % (sys.argv[1], sys.argv[2], d['value']/1e6, d['us_per_tti_per_cell']))" "$tag" "[$extra]" || tail -3 $O/ab_$tag.log
}
for v in "" "-DRS_PF1_ALWAYS"; do
ab s1_r25 "$v" --sched 1 --ttis 4000
ab s1_r64 "$v" --sched 1 --ttis 4000 --rbgs 64 --rbg-size 8
done
for v in "" "-URS_P3_BLOCK_TOP -DRS_P3_BLOCK_TOP=16" "-URS_P3_BLOCK -DRS_P3_BLOCK=32" "-URS_P3_BLOCK_TOP -DRS_P3_BLOCK_TOP=0" "-DRS_NO_HOLD"; do
ab s9_r25 "$v" --sched 9 --ttis 8000
done
for v in "" "-DRS_HOLD_ALWAYS" "-URS_P3_BLOCK -DRS_P3_BLOCK=16"; do
ab s9_r64 "$v" --sched 9 --ttis 4000 --rbgs 64 --rbg-size 8
ab s8_r64 "$v" --sched 8 --ttis 4000 --rbgs 64 --rbg-size 8
done
for v in "" "-DRS_P3_BLOCK_TOP=8" "-DRS_NO_HOLD"; do
ab s8_r25 "$v" --sched 8 --ttis 4000
done
