#!/bin/bash
# Round 5: the sampler with the metric array sized by the longest slice -- 1 000 UEs x 25 RBGs gets two cells per CU
cd $GRAFT_REPO_ROOT; O=gpurun_out/nvs_exp10; mkdir -p $O
python -m pytest tests -m gpu -x -q -k "sampler or nongreedy or random_shapes or lean_build" > $O/pytest.log 2>&1; tail -2 $O/pytest.log
ab() { local tag=$1; shift 1
  timeout 300 python bench.py --no-cpu-baseline --no-r64 --no-streamed --steps 3 --warmup 1 "$@" > $O/ab_$tag.log 2>&1
  grep -h '^{' $O/ab_$tag.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%-14s %.2f M TTIs/s  %.3f us' % (sys.argv[1], d['value']/1e6, d['us_per_tti_per_cell']))" "$tag" || tail -3 $O/ab_$tag.log
}
ab 1000x25 --sched 11 --ttis 1000 --ues-per-slice 50
ab 500x25 --sched 11 --ttis 2000
ab 500x64 --sched 11 --ttis 1000 --rbgs 64 --rbg-size 8
ab ng30 --sched 11 --ttis 1000 --ues-per-slice 30 --rbgs 64 --rbg-size 8
ab 1000x64 --sched 11 --ttis 500 --ues-per-slice 50 --rbgs 64 --rbg-size 8
