#!/bin/bash
# Round 5: NVS non-greedy sampler, roles of the waves -- who adds the samples up, at what priority (RS_JIT_EXTRA builds of one source)
cd $GRAFT_REPO_ROOT; O=gpurun_out/nvs_exp4; mkdir -p $O
ab() { local tag=$1 extra=$2; shift 2
  RS_JIT_EXTRA="$extra" timeout 300 python bench.py --allow-variant --no-cpu-baseline --no-r64 --no-streamed --steps 3 --warmup 1 "$@" > $O/ab_$tag.log 2>&1
  grep -h '^{' $O/ab_$tag.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%-14s %-60s %.2f M TTIs/s  %.3f us' % (sys.argv[1], sys.argv[2], d['value']/1e6, d['us_per_tti_per_cell']))" "$tag" "[$extra]" || tail -3 $O/ab_$tag.log
}
i=0
for x in "-DRS_NVS_PIPE=0" "" "-DRS_NVS_SERVE_PRIO=2" "-DRS_NVS_SUM_WAVE0=1" "-DRS_NVS_SUM_WAVE0=1 -DRS_NVS_SERVE_PRIO=2" "-DRS_NVS_PIPE=0 -DRS_NVS_SERVE_PRIO=2"; do
  i=$((i+1))
  ab r25_$i "$x" --sched 11 --ttis 2000
  ab r64_$i "$x" --sched 11 --ttis 1000 --rbgs 64 --rbg-size 8
  ab u200r64_$i "$x" --sched 11 --ttis 2000 --ues-per-slice 10 --rbgs 64 --rbg-size 8
done
