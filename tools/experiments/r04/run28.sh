#!/bin/bash
# Round 4, GPU run 28: MaximizeCell's idle waves pull the next grid into the L2 (no fetch ahead) -- A/B against -DRS_NO_GRID_TOUCH; the block-8 rule of sched 8 / 101 / 103
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_run28; mkdir -p $O; cd ..
ab() { local tag=$1 extra=$2; shift 2
  RS_JIT_EXTRA="$extra" timeout 300 python bench.py --allow-variant --no-cpu-baseline --no-r64 --no-streamed --steps 5 --warmup 1 "$@" > $O/ab_$tag.log 2>&1
  grep -h '^{' $O/ab_$tag.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%-28s %-28s %.2f M TTIs/s  %.3f us' % (sys.argv[1], sys.argv[2], d['value']/1e6, d['us_per_tti_per_cell']))" "$tag" "[$extra]" || tail -3 $O/ab_$tag.log
}
for rep in 1 2; do for v in "" "-DRS_NO_GRID_TOUCH"; do
ab s9_stream_$rep "$v" --sched 9 --cqi-refresh 1 --ttis 2000
ab s9_res_$rep "$v" --sched 9 --ttis 8000
ab s9_r64_stream_$rep "$v" --sched 9 --cqi-refresh 1 --ttis 1000 --rbgs 64 --rbg-size 8
done; done
for s in 8 101 103; do
ab s${s}_stream "" --sched $s --cqi-refresh 1 --ttis 2000
ab s${s}_u1000_res "" --sched $s --ttis 4000 --ues-per-slice 50
ab s${s}_res "" --sched $s --ttis 4000
done
python -m pytest tests -m gpu -q -x -k "streamed or fetched or 1000 or sched" 2>&1 | tail -3
