#!/bin/bash
# Round 4, GPU run 30: does the fetch-ahead code cost the resident mode anything?  (64-RBG grid and 1 000 UEs, per scheduler)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_run30; mkdir -p $O; cd ..
ab() { local tag=$1 extra=$2; shift 2
  RS_JIT_EXTRA="$extra" timeout 300 python bench.py --allow-variant --no-cpu-baseline --no-r64 --no-streamed --steps 5 --warmup 1 "$@" > $O/ab_$tag.log 2>&1
  grep -h '^{' $O/ab_$tag.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%-28s %-24s %.2f M TTIs/s  %.3f us' % (sys.argv[1], sys.argv[2], d['value']/1e6, d['us_per_tti_per_cell']))" "$tag" "[$extra]" || tail -3 $O/ab_$tag.log
}
for rep in 1 2; do for v in "" "-DRS_NO_GRID_AHEAD"; do
ab s8_r64_$rep "$v" --sched 8 --ttis 4000 --rbgs 64 --rbg-size 8
ab s7_r64_$rep "$v" --sched 7 --ttis 4000 --rbgs 64 --rbg-size 8
ab s7_r25_$rep "$v" --sched 7 --ttis 4000
ab s8_r25_$rep "$v" --sched 8 --ttis 4000
ab s1_r25_$rep "$v" --sched 1 --ttis 4000
done; done
