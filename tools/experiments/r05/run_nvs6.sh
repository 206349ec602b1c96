#!/bin/bash
# Round 5: the sampler on cells of few slices (key table sized for the longest slice), this tree against the tree before the sampler work
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/nvs_exp7; mkdir -p $O
python -m pytest tests -m gpu -x -q -k "sampler or nongreedy" > $O/pytest.log 2>&1; tail -2 $O/pytest.log
ab() { local dir=$1 tag=$2; shift 2
  (cd $dir && timeout 300 python bench.py --no-cpu-baseline --no-r64 --no-streamed --steps 3 --warmup 1 "$@" > $O/ab_$tag.log 2>&1)
  grep -h '^{' $O/ab_$tag.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%-22s %.2f M TTIs/s  %.3f us' % (sys.argv[1], d['value']/1e6, d['us_per_tti_per_cell']))" "$tag" || tail -3 $O/ab_$tag.log
}
for t in old new; do
  dir=.; [ $t = old ] && dir=scratch_nvs
  ab $dir ${t}_3x20_r64 --sched 11 --ttis 2000 --slices 3 --ues-per-slice 20 --rbgs 64 --rbg-size 8
  ab $dir ${t}_5x10_r64 --sched 11 --ttis 2000 --slices 5 --ues-per-slice 10 --rbgs 64 --rbg-size 8
  ab $dir ${t}_3x20_r25 --sched 11 --ttis 2000 --slices 3 --ues-per-slice 20
  ab $dir ${t}_4x60_r25 --sched 11 --ttis 2000 --slices 4 --ues-per-slice 60
  ab $dir ${t}_500x64 --sched 11 --ttis 1000 --rbgs 64 --rbg-size 8
done
