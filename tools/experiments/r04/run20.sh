#!/bin/bash
# Round 4, GPU run 20: device-resident CQI grids as the LDS image + the next grid fetched during the serial phase (kGridAhead):
# parity (whole suite), then same-box A/B against -DRS_NO_GRID_AHEAD in the streamed and the resident mode
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_run20; mkdir -p $O; cd ..
python -m pytest tests -m gpu -x -q > $O/pytest_all.log 2>&1; echo "pytest rc $?" >> $O/pytest_all.log
tail -4 $O/pytest_all.log
ab() { # tag, extra, bench args
  local tag=$1 extra=$2; shift 2
  RS_JIT_EXTRA="$extra" timeout 300 python bench.py --allow-variant --no-cpu-baseline --no-r64 --no-streamed --steps 5 --warmup 1 "$@" > $O/ab_$tag.log 2>&1
  grep -h '^{' $O/ab_$tag.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%-28s %-24s %.2f M TTIs/s  %.3f us' % (sys.argv[1], sys.argv[2], d['value']/1e6, d['us_per_tti_per_cell']))" "$tag" "[$extra]" || tail -3 $O/ab_$tag.log
}
for rep in 1 2; do
for v in "" "-DRS_NO_GRID_AHEAD"; do
ab s9_stream_$rep "$v" --sched 9 --cqi-refresh 1 --ttis 2000
ab s9_res_$rep "$v" --sched 9 --ttis 8000
ab s8_stream_$rep "$v" --sched 8 --cqi-refresh 1 --ttis 2000
ab s9_r64_stream_$rep "$v" --sched 9 --rbgs 64 --rbg-size 8 --cqi-refresh 1 --ttis 1000
done; done
ab s7_stream "" --sched 7 --cqi-refresh 1 --ttis 2000
ab s7_stream "-DRS_NO_GRID_AHEAD" --sched 7 --cqi-refresh 1 --ttis 2000
ab s1_stream "" --sched 1 --cqi-refresh 1 --ttis 2000
ab s1_stream "-DRS_NO_GRID_AHEAD" --sched 1 --cqi-refresh 1 --ttis 2000
