#!/bin/bash
# Round 4, GPU run 4: specialised drop-in kernel (parity + latency), phase stamps of schedulers 7 / 1 / 8, sched-1 rule check
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_run4; mkdir -p $O; cd $R
python -m pytest tests/test_gpu_round4.py -m gpu -x -q -k "specialised or prepare" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -4 $O/pytest.log
tools/dropin_latency 2000 > $O/dropin_latency.log 2>&1; cat $O/dropin_latency.log
RS_DROPIN_TIMING=1 tools/dropin_latency 1000 > $O/dropin_latency_timing.log 2>&1; grep -i "prep\|enq\|wait" $O/dropin_latency_timing.log | head -20
export RS_HIP_LIB=radiosaber_amd/libradiosaber_hip_stamps.so
for s in 7 1 8; do
  echo "=== stamps sched $s (early preparation)"; RS_JIT_EXTRA="-DRS_STAMPS" timeout 200 python tools/phase_stamps.py --jit --sched $s 2>&1 | grep -v "^    " | tee $O/stamps_s$s.log
  echo "=== stamps sched $s wave 1"; RS_JIT_EXTRA="-DRS_STAMPS -DRS_STAMPS_W1" timeout 200 python tools/phase_stamps.py --jit --w1 --sched $s 2>&1 | grep "^    \|launch" | tee $O/stamps_w1_s$s.log
done
echo "=== stamps sched 7 base"; RS_JIT_EXTRA="-DRS_STAMPS -DRS_NO_EARLY17" timeout 200 python tools/phase_stamps.py --jit --sched 7 2>&1 | grep -v "^    " | tee $O/stamps_s7_base.log
unset RS_HIP_LIB
for s in 1; do for u in 25 50; do
RS_JIT_EXTRA="" timeout 200 python bench.py --allow-variant --no-cpu-baseline --no-r64 --no-streamed --steps 5 --warmup 1 --ttis 4000 --sched $s --ues-per-slice $u 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('sched', d['config']['sched'], 'ues', d['config']['ues'], '%.2f M TTIs/s' % (d['value']/1e6))"
done; done
