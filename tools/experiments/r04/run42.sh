#!/bin/bash
# Round 4, GPU run 42: optimisation level of the hiprtc build (code size against speed)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_run42; mkdir -p $O; cd ..
ab() { local tag=$1 extra=$2; shift 2
  RS_JIT_EXTRA="$extra" timeout 300 python bench.py --allow-variant --no-cpu-baseline --no-r64 --no-streamed --steps 5 --warmup 1 "$@" > $O/ab_$tag.log 2>&1
  grep -h '^{' $O/ab_$tag.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%-20s %-36s %.2f M TTIs/s  %.3f us' % (sys.argv[1], sys.argv[2], d['value']/1e6, d['us_per_tti_per_cell']))" "$tag" "[$extra]" || tail -3 $O/ab_$tag.log
}
for v in "" "-O2" "-Os" "-mllvm -inline-threshold=100" "-fno-unroll-loops"; do
ab s9_r25 "$v" --sched 9 --ttis 8000
ab s7_r25 "$v" --sched 7 --ttis 4000
ab s8_r25 "$v" --sched 8 --ttis 4000
done
