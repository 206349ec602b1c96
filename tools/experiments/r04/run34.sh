#!/bin/bash
# Round 4, GPU run 34: MaximizeCell with the fetch ahead, now on the lean build (streamed mode); stage-1 block of its full scan again
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_run34; mkdir -p $O; cd ..
ab() { local tag=$1 extra=$2; shift 2
  RS_JIT_EXTRA="$extra" timeout 300 python bench.py --allow-variant --no-cpu-baseline --no-r64 --no-streamed --steps 5 --warmup 1 "$@" > $O/ab_$tag.log 2>&1
  grep -h '^{' $O/ab_$tag.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%-28s %-44s %.2f M TTIs/s  %.3f us' % (sys.argv[1], sys.argv[2], d['value']/1e6, d['us_per_tti_per_cell']))" "$tag" "[$extra]" || tail -3 $O/ab_$tag.log
}
for rep in 1 2; do for v in "" "-DRS_GRID_AHEAD_ALL" "-URS_P3_BLOCK_TOP -DRS_P3_BLOCK_TOP=16" "-DRS_GRID_AHEAD_ALL -URS_P3_BLOCK_TOP -DRS_P3_BLOCK_TOP=16"; do
ab s9_stream_$rep "$v" --sched 9 --ttis 2000 --cqi-refresh 1
ab s9_r64_stream_$rep "$v" --sched 9 --ttis 1000 --cqi-refresh 1 --rbgs 64 --rbg-size 8
done; done
