#!/bin/bash
# Round 4, GPU run 32: the lean build (run-time options of the long runs as constants) against the general shape-specialised kernel
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_run32; mkdir -p $O; cd ..
ab() { local tag=$1 extra=$2; shift 2
  RS_JIT_EXTRA="$extra" timeout 300 python bench.py --allow-variant --no-cpu-baseline --no-r64 --no-streamed --steps 5 --warmup 1 "$@" > $O/ab_$tag.log 2>&1
  grep -h '^{' $O/ab_$tag.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%-28s %-24s %.2f M TTIs/s  %.3f us' % (sys.argv[1], sys.argv[2], d['value']/1e6, d['us_per_tti_per_cell']))" "$tag" "[$extra]" || tail -3 $O/ab_$tag.log
}
for rep in 1 2; do for v in "" "-DRS_JIT_LEAN=1"; do
ab s9_r25_$rep "$v" --sched 9 --ttis 8000
ab s9_r64_$rep "$v" --sched 9 --ttis 4000 --rbgs 64 --rbg-size 8
ab s9_stream_$rep "$v" --sched 9 --ttis 2000 --cqi-refresh 1
ab s8_r25_$rep "$v" --sched 8 --ttis 4000
ab s7_r25_$rep "$v" --sched 7 --ttis 4000
ab s1_r25_$rep "$v" --sched 1 --ttis 4000
ab s9_u1000_$rep "$v" --sched 9 --ttis 4000 --ues-per-slice 50
done; done
