#!/bin/bash
# Round 4, GPU run 25: a full-scan TTI on four lanes per item (the listed items' pass over all items) -- parity, then same-box A/B against
# -DRS_HOLD_FULL_ITEMS (one lane per item) in the streamed and the resident mode
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_run25; mkdir -p $O; cd ..
python -m pytest tests -m gpu -q -x > $O/pytest_all.log 2>&1; echo "pytest rc $?" >> $O/pytest_all.log
grep -n "FAILED\|passed\|failed\|rc " $O/pytest_all.log | tail -8
ab() { local tag=$1 extra=$2; shift 2
  RS_JIT_EXTRA="$extra" timeout 300 python bench.py --allow-variant --no-cpu-baseline --no-r64 --no-streamed --steps 5 --warmup 1 "$@" > $O/ab_$tag.log 2>&1
  grep -h '^{' $O/ab_$tag.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%-28s %-24s %.2f M TTIs/s  %.3f us' % (sys.argv[1], sys.argv[2], d['value']/1e6, d['us_per_tti_per_cell']))" "$tag" "[$extra]" || tail -3 $O/ab_$tag.log
}
for rep in 1 2; do for v in "" "-DRS_HOLD_FULL_ITEMS"; do
ab s9_stream_$rep "$v" --sched 9 --cqi-refresh 1 --ttis 2000
ab s9_res_$rep "$v" --sched 9 --ttis 8000
ab s8_stream_$rep "$v" --sched 8 --cqi-refresh 1 --ttis 2000
ab s8_res_$rep "$v" --sched 8 --ttis 4000
ab s9_u1000_stream_$rep "$v" --sched 9 --cqi-refresh 1 --ttis 2000 --ues-per-slice 50
done; done
