#!/bin/bash
# Round 4, GPU run 46: the two late rule changes together (NVS whole slices up to 64 users; MaximizeCell's early EWMA from two users per thread): parity subset + the shapes they touch
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_run46; mkdir -p $O; cd ..
python -m pytest tests -m gpu -q -x -k "sched or nvs or 1000 or prepare or random or config or lean or held or maximum" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
grep -n "FAILED\|passed\|failed\|rc " $O/pytest.log | tail -4
ab() { local tag=$1 extra=$2; shift 2
  RS_JIT_EXTRA="$extra" timeout 300 python bench.py --allow-variant --no-cpu-baseline --no-r64 --no-streamed --steps 5 --warmup 1 "$@" > $O/ab_$tag.log 2>&1
  grep -h '^{' $O/ab_$tag.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%-20s %-12s %.2f M TTIs/s  %.3f us' % (sys.argv[1], sys.argv[2], d['value']/1e6, d['us_per_tti_per_cell']))" "$tag" "[$extra]" || tail -3 $O/ab_$tag.log
}
ab s7_u1000 "" --sched 7 --ttis 4000 --ues-per-slice 50
ab s9_u1000 "" --sched 9 --ttis 4000 --ues-per-slice 50
ab s9_r25 "" --sched 9 --ttis 8000
ab s7_r25 "" --sched 7 --ttis 4000
timeout 600 python tools/soak.py --sched 9 --jit 1 --ues-per-slice 50 --cells 16 --ttis 6000 | grep SOAK
timeout 600 python tools/soak.py --sched 7 --jit 1 --ues-per-slice 50 --ttis 8000 | grep SOAK
timeout 600 python tools/soak.py --sched 7 --jit 1 --slices 7 --ues-per-slice 60 --ttis 6000 --phy 1 --launch 97 | grep SOAK
