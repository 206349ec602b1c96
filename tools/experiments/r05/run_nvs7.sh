#!/bin/bash
# Round 5: the sampler's drawing wave through both phases of a step: share drawn beside the adding-up, its issue priority
cd $GRAFT_REPO_ROOT; O=gpurun_out/nvs_exp9; mkdir -p $O
python -m pytest tests -m gpu -x -q -k "sampler or nongreedy" > $O/pytest.log 2>&1; tail -2 $O/pytest.log
ab() { local tag=$1 extra=$2; shift 2
  RS_JIT_EXTRA="$extra" timeout 300 python bench.py --allow-variant --no-cpu-baseline --no-r64 --no-streamed --steps 3 --warmup 1 "$@" > $O/ab_$tag.log 2>&1
  grep -h '^{' $O/ab_$tag.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%-14s %-50s %.2f M TTIs/s  %.3f us' % (sys.argv[1], sys.argv[2], d['value']/1e6, d['us_per_tti_per_cell']))" "$tag" "[$extra]" || tail -3 $O/ab_$tag.log
}
i=0
for x in "-DRS_NVS_P1_PCT=0" "" "-DRS_NVS_P1_PCT=65" "-DRS_NVS_DRAW_PRIO=2" "-DRS_NVS_P1_PCT=65 -DRS_NVS_DRAW_PRIO=2" "-DRS_NVS_P1_PCT=35"; do
  i=$((i+1))
  ab r25_$i "$x" --sched 11 --ttis 2000
  ab r64_$i "$x" --sched 11 --ttis 1000 --rbgs 64 --rbg-size 8
  ab ng20_$i "$x" --sched 11 --ttis 1000 --ues-per-slice 20 --rbgs 64 --rbg-size 8
done
