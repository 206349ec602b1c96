#!/bin/bash
# Round 4, GPU run 27: 8 users per stage-1 block for the top-of-TTI full scan of the held-winner kernels other than MaximizeCell
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_run27; mkdir -p $O; cd ..
ab() { local tag=$1 extra=$2; shift 2
  RS_JIT_EXTRA="$extra" timeout 300 python bench.py --allow-variant --no-cpu-baseline --no-r64 --no-streamed --steps 5 --warmup 1 "$@" > $O/ab_$tag.log 2>&1
  grep -h '^{' $O/ab_$tag.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%-28s %-28s %.2f M TTIs/s  %.3f us' % (sys.argv[1], sys.argv[2], d['value']/1e6, d['us_per_tti_per_cell']))" "$tag" "[$extra]" || tail -3 $O/ab_$tag.log
}
for v in "" "-DRS_P3_BLOCK_TOP=8" "-DRS_P3_BLOCK_TOP=16"; do
for s in 8 103; do
ab s${s}_stream "$v" --sched $s --cqi-refresh 1 --ttis 2000
ab s${s}_res "$v" --sched $s --ttis 4000
ab s${s}_u1000_stream "$v" --sched $s --cqi-refresh 1 --ttis 2000 --ues-per-slice 50
ab s${s}_u1000_res "$v" --sched $s --ttis 4000 --ues-per-slice 50
done; done
