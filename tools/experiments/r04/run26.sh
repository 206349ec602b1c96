#!/bin/bash
# Round 4, GPU run 26: users per stage-1 block of the top-of-TTI scan in the streamed mode (a full scan every TTI)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_run26; mkdir -p $O; cd ..
ab() { local tag=$1 extra=$2; shift 2
  RS_JIT_EXTRA="$extra" timeout 300 python bench.py --allow-variant --no-cpu-baseline --no-r64 --no-streamed --steps 5 --warmup 1 "$@" > $O/ab_$tag.log 2>&1
  grep -h '^{' $O/ab_$tag.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%-28s %-36s %.2f M TTIs/s  %.3f us' % (sys.argv[1], sys.argv[2], d['value']/1e6, d['us_per_tti_per_cell']))" "$tag" "[$extra]" || tail -3 $O/ab_$tag.log
}
for v in "" "-URS_P3_BLOCK_TOP -DRS_P3_BLOCK_TOP=16" "-URS_P3_BLOCK_TOP -DRS_P3_BLOCK_TOP=32" "-URS_P3_BLOCK_TOP -DRS_P3_BLOCK_TOP=24"; do
ab s9_stream "$v" --sched 9 --cqi-refresh 1 --ttis 2000
ab s9_stream4 "$v" --sched 9 --cqi-refresh 4 --ttis 2000
done
for v in "" "-URS_P3_BLOCK -DRS_P3_BLOCK=16" "-URS_P3_BLOCK -DRS_P3_BLOCK=8"; do
ab s8_stream "$v" --sched 8 --cqi-refresh 1 --ttis 2000
done
