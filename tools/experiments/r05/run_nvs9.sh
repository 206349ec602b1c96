#!/bin/bash
# Round 5: the sampler's adding-up with every unit of winners loaded first (-DRS_NVS_SUM_AHEAD=1)
cd $GRAFT_REPO_ROOT; O=gpurun_out/nvs_exp11; mkdir -p $O
ab() { local tag=$1 extra=$2; shift 2
  RS_JIT_EXTRA="$extra" timeout 300 python bench.py --allow-variant --no-cpu-baseline --no-r64 --no-streamed --steps 3 --warmup 1 "$@" > $O/ab_$tag.log 2>&1
  grep -h '^{' $O/ab_$tag.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%-14s %-28s %.2f M TTIs/s  %.3f us' % (sys.argv[1], sys.argv[2], d['value']/1e6, d['us_per_tti_per_cell']))" "$tag" "[$extra]" || tail -3 $O/ab_$tag.log
}
for x in "" "-DRS_NVS_SUM_AHEAD=0"; do
  t=$(echo "$x" | tr -d ' -=' )
  ab r64_$t "$x" --sched 11 --ttis 1000 --rbgs 64 --rbg-size 8
  ab ng20_$t "$x" --sched 11 --ttis 1000 --ues-per-slice 20 --rbgs 64 --rbg-size 8
  ab r25_$t "$x" --sched 11 --ttis 2000
done
