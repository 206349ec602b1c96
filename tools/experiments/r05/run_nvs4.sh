#!/bin/bash
# Round 5: the sampler (sched 11) on the reference's own exp-nongreedy shapes (20 slices x 10 / 20 / 30 UEs, 64 RBGs) and the sweep shapes,
# this tree against the tree before the sampler work, same lease.  scratch_nvs/ (git-ignored, travels with gpurun) is recreated HERE by:
#   mkdir scratch_nvs && git archive 0583421 radiosaber_amd bench.py oracle include tests/golden tests/conftest.py | tar -x -C scratch_nvs && (cd scratch_nvs && python -m radiosaber_amd.build --force)
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/nvs_exp5; mkdir -p $O
ab() { local dir=$1 tag=$2; shift 2
  (cd $dir && timeout 300 python bench.py --no-cpu-baseline --no-r64 --no-streamed --steps 3 --warmup 1 "$@" > $O/ab_$tag.log 2>&1)
  grep -h '^{' $O/ab_$tag.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%-22s %.2f M TTIs/s  %.3f us' % (sys.argv[1], d['value']/1e6, d['us_per_tti_per_cell']))" "$tag" || tail -3 $O/ab_$tag.log
}
for t in old new; do
  dir=.; [ $t = old ] && dir=scratch_nvs
  ab $dir ${t}_ng10 --sched 11 --ttis 2000 --ues-per-slice 10 --rbgs 64 --rbg-size 8
  ab $dir ${t}_ng20 --sched 11 --ttis 1000 --ues-per-slice 20 --rbgs 64 --rbg-size 8
  ab $dir ${t}_ng30 --sched 11 --ttis 1000 --ues-per-slice 30 --rbgs 64 --rbg-size 8
  ab $dir ${t}_500x64 --sched 11 --ttis 1000 --rbgs 64 --rbg-size 8
  ab $dir ${t}_100x64 --sched 11 --ttis 2000 --ues-per-slice 5 --rbgs 64 --rbg-size 8
  ab $dir ${t}_500x25 --sched 11 --ttis 2000
  ab $dir ${t}_1000x25 --sched 11 --ttis 1000 --ues-per-slice 50
done
