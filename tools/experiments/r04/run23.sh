#!/bin/bash
# Round 4, GPU run 23: whole GPU suite on the grid-ahead build (MaximizeCell without it), then the bench line and the streamed mode per scheduler
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_run23; mkdir -p $O; cd ..
python -m pytest tests -m gpu -q > $O/pytest_all.log 2>&1; echo "pytest rc $?" >> $O/pytest_all.log
grep -n "FAILED\|passed\|failed\|rc " $O/pytest_all.log | tail -8
python bench.py --no-cpu-baseline > $O/bench.log 2> $O/bench.err; python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r04_run23/bench.log') if l.startswith('{')][-1])
print('value %.2f M  r64 %.2f M  streamed %.2f M frac %.4f' % (d['value']/1e6, d.get('value_r64',0)/1e6, d['roofline']['streamed']['value']/1e6, d['roofline']['streamed']['frac']))
PY
ab() { local tag=$1 extra=$2; shift 2
  RS_JIT_EXTRA="$extra" timeout 300 python bench.py --allow-variant --no-cpu-baseline --no-r64 --no-streamed --steps 5 --warmup 1 "$@" > $O/ab_$tag.log 2>&1
  grep -h '^{' $O/ab_$tag.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%-28s %-24s %.2f M TTIs/s  %.3f us' % (sys.argv[1], sys.argv[2], d['value']/1e6, d['us_per_tti_per_cell']))" "$tag" "[$extra]" || tail -3 $O/ab_$tag.log
}
for s in 8 7 1 101 103; do ab s${s}_stream "" --sched $s --cqi-refresh 1 --ttis 2000; ab s${s}_res "" --sched $s --ttis 4000; done
