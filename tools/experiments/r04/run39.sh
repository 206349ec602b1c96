#!/bin/bash
# Round 4, GPU run 39: lean build of the queue-model kernels -- parity (queue tests + fuzz through the lean build), bench_queue_mode with / without
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_run39; mkdir -p $O; cd ..
python -m pytest tests/test_gpu_queues.py -m gpu -q > $O/pytest_queues.log 2>&1; echo "pytest rc $?" >> $O/pytest_queues.log
grep -n "FAILED\|passed\|failed\|rc \|Error" $O/pytest_queues.log | tail -6
for v in 1 0 1 0; do
echo "--- RS_JIT_LEAN=$v"; RS_JIT_LEAN=$v python tools/bench_queue_mode.py 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(' ', d.get('sched'), '%.2f M TTIs/s' % (d['ttis_per_s']/1e6) if 'ttis_per_s' in d else d)
"
done
