#!/bin/bash
# Round 4, GPU run 38: users per stage-1 block of the 64-RBG MaximizeCell kernel (speculative scan in the serial phase) on the lean build
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_run38; mkdir -p $O; cd ..
ab() { local tag=$1 extra=$2; shift 2
  RS_JIT_EXTRA="$extra" timeout 300 python bench.py --allow-variant --no-cpu-baseline --no-r64 --no-streamed --steps 5 --warmup 1 "$@" > $O/ab_$tag.log 2>&1
  grep -h '^{' $O/ab_$tag.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%-20s %-44s %.2f M TTIs/s  %.3f us' % (sys.argv[1], sys.argv[2], d['value']/1e6, d['us_per_tti_per_cell']))" "$tag" "[$extra]" || tail -3 $O/ab_$tag.log
}
for rep in 1 2; do for v in "" "-URS_P3_BLOCK -DRS_P3_BLOCK=16" "-URS_P3_BLOCK -DRS_P3_BLOCK=24" "-URS_P3_BLOCK -DRS_P3_BLOCK=32"; do
ab s9_r64_$rep "$v" --sched 9 --ttis 4000 --rbgs 64 --rbg-size 8
done; done
for v in "" "-URS_P3_BLOCK -DRS_P3_BLOCK=16"; do
RS_JIT_LEAN=0 ab s9_r64_general "$v" --sched 9 --ttis 4000 --rbgs 64 --rbg-size 8
ab s9_r64_u100 "$v" --sched 9 --ttis 4000 --rbgs 64 --rbg-size 8 --ues-per-slice 5
ab s9_r50 "$v" --sched 9 --ttis 4000 --rbgs 50 --rbg-size 4
done
