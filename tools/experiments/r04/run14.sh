#!/bin/bash
# Round 4, GPU run 14: per-flow PF's per-RBG reduction on all waves (16 lanes per RBG) -- parity, rates, stamps
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_run14; mkdir -p $O; cd ..
python -m pytest tests -m gpu -x -q > $O/pytest_all.log 2>&1; echo "pytest rc $?" >> $O/pytest_all.log; tail -3 $O/pytest_all.log
ab() { # tag, extra, bench args
  local tag=$1 extra=$2; shift 2
  RS_JIT_EXTRA="$extra" timeout 300 python bench.py --allow-variant --no-cpu-baseline --no-r64 --no-streamed --steps 4 --warmup 1 --ttis 4000 "$@" > $O/ab_$tag.log 2>&1
  grep -h '^{' $O/ab_$tag.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%-28s %-20s %.3f M TTIs/s  %.3f us' % (sys.argv[1], sys.argv[2], d['value']/1e6, d['us_per_tti_per_cell']))" "$tag" "[$extra]" || tail -3 $O/ab_$tag.log
}
for rep in 1 2; do
ab s1_r25_$rep "" --sched 1
ab s1_u1000_$rep "" --sched 1 --ues-per-slice 50
ab s1_r64_$rep "" --sched 1 --rbgs 64 --rbg-size 8
ab s7_r25_$rep "" --sched 7
ab s9_r25_$rep "" --sched 9
done
export RS_HIP_LIB=radiosaber_amd/libradiosaber_hip_stamps.so
echo "=== sched 1"; RS_JIT_EXTRA="-DRS_STAMPS" timeout 200 python tools/phase_stamps.py --jit --sched 1 2>&1 | grep -v "sort \|held winners\|greedy:\|^    " | tee $O/stamps_s1.log
