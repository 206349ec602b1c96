#!/bin/bash
# Round 4, GPU run 44: the early-EWMA rules re-measured on the lean builds (per-flow PF at one user per thread; MaximizeCell / SubOpt / Vogel)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_run44; mkdir -p $O; cd ..
ab() { local tag=$1 extra=$2; shift 2
  RS_JIT_EXTRA="$extra" timeout 300 python bench.py --allow-variant --no-cpu-baseline --no-r64 --no-streamed --steps 5 --warmup 1 "$@" > $O/ab_$tag.log 2>&1
  grep -h '^{' $O/ab_$tag.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%-20s %-24s %.2f M TTIs/s  %.3f us' % (sys.argv[1], sys.argv[2], d['value']/1e6, d['us_per_tti_per_cell']))" "$tag" "[$extra]" || tail -3 $O/ab_$tag.log
}
for rep in 1 2; do
for v in "" "-DRS_EARLY17_ALL"; do
ab s1_r25_$rep "$v" --sched 1 --ttis 4000
ab s1_r64_$rep "$v" --sched 1 --ttis 4000 --rbgs 64 --rbg-size 8
done
for v in "" "-DRS_EWMA_NEXT_ALL"; do
ab s9_r25_$rep "$v" --sched 9 --ttis 8000
ab s9_u1000_$rep "$v" --sched 9 --ttis 4000 --ues-per-slice 50
ab s103_r25_$rep "$v" --sched 103 --ttis 2000
ab s101_r25_$rep "$v" --sched 101 --ttis 2000
done; done
