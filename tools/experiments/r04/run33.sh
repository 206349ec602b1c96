#!/bin/bash
# Round 4, GPU run 33: the lean build chosen by the library itself -- its parity test, the whole suite, the default bench line
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_run33; mkdir -p $O; cd ..
python -m pytest tests/test_gpu_round4.py -m gpu -q -k "lean" > $O/pytest_lean.log 2>&1; echo "pytest rc $?" >> $O/pytest_lean.log
grep -n "FAILED\|passed\|failed\|rc \|Error" $O/pytest_lean.log | tail -12
python -m pytest tests -m gpu -q > $O/pytest_all.log 2>&1; echo "pytest rc $?" >> $O/pytest_all.log
grep -n "FAILED\|passed\|failed\|rc " $O/pytest_all.log | tail -8
python bench.py --no-cpu-baseline > $O/bench.log 2> $O/bench.err; python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r04_run33/bench.log') if l.startswith('{')][-1])
print('value %.2f M  r64 %.2f M  streamed %.2f M' % (d['value']/1e6, d.get('value_r64',0)/1e6, d['roofline']['streamed']['value']/1e6))
PY
RS_JIT_LEAN=0 python bench.py --no-cpu-baseline --no-r64 --no-streamed > $O/bench_nolean.log 2>&1; grep -o '"value": [0-9.]*' $O/bench_nolean.log | head -1
