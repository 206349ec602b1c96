#!/bin/bash
# Round 5: the sampler at the end of the round, every shape of profiles/r05_notes.md §6 on one lease (device sources ff3f769241f35787)
cd $GRAFT_REPO_ROOT; O=gpurun_out/nvs_exp12; mkdir -p $O
python -m pytest tests -m gpu -q -k "sampler" 2>&1 | tail -1
ab() { local tag=$1; shift 1
  timeout 300 python bench.py --no-cpu-baseline --no-r64 --no-streamed --steps 3 --warmup 1 "$@" > $O/ab_$tag.log 2>&1
  grep -h '^{' $O/ab_$tag.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%-14s %.2f M TTIs/s  %.3f us' % (sys.argv[1], d['value']/1e6, d['us_per_tti_per_cell']))" "$tag" || tail -3 $O/ab_$tag.log
}
ab ng10 --sched 11 --ttis 2000 --ues-per-slice 10 --rbgs 64 --rbg-size 8
ab ng20 --sched 11 --ttis 1000 --ues-per-slice 20 --rbgs 64 --rbg-size 8
ab ng30 --sched 11 --ttis 1000 --ues-per-slice 30 --rbgs 64 --rbg-size 8
ab 500x64 --sched 11 --ttis 1000 --rbgs 64 --rbg-size 8
ab 100x64 --sched 11 --ttis 2000 --ues-per-slice 5 --rbgs 64 --rbg-size 8
ab 500x25 --sched 11 --ttis 2000
ab 1000x25 --sched 11 --ttis 1000 --ues-per-slice 50
ab 3x20_r64 --sched 11 --ttis 2000 --slices 3 --ues-per-slice 20 --rbgs 64 --rbg-size 8
ab 5x10_r64 --sched 11 --ttis 2000 --slices 5 --ues-per-slice 10 --rbgs 64 --rbg-size 8
ab 3x20_r25 --sched 11 --ttis 2000 --slices 3 --ues-per-slice 20
ab 4x60_r25 --sched 11 --ttis 2000 --slices 4 --ues-per-slice 60
