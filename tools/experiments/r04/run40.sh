#!/bin/bash
# Round 4, GPU run 40: NVS non-greedy sampler (sched 11) on 16-bit keys -- parity, then the three sweep shapes
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_run40; mkdir -p $O; cd ..
python -m pytest tests -m gpu -q -k "nongreedy or non_greedy or sampler or random_shapes or specialised_drop_in or config or lean_build" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
grep -n "FAILED\|passed\|failed\|rc \|Error" $O/pytest.log | tail -8
ab() { local tag=$1 extra=$2; shift 2
  RS_JIT_EXTRA="$extra" timeout 300 python bench.py --allow-variant --no-cpu-baseline --no-r64 --no-streamed --steps 3 --warmup 1 "$@" > $O/ab_$tag.log 2>&1
  grep -h '^{' $O/ab_$tag.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%-20s %-12s %.2f M TTIs/s  %.3f us' % (sys.argv[1], sys.argv[2], d['value']/1e6, d['us_per_tti_per_cell']))" "$tag" "[$extra]" || tail -3 $O/ab_$tag.log
}
ab s11_r25 "" --sched 11 --ttis 2000
ab s11_r64 "" --sched 11 --ttis 1000 --rbgs 64 --rbg-size 8
ab s11_u1000 "" --sched 11 --ttis 2000 --ues-per-slice 50
ab s11_u100_r64 "" --sched 11 --ttis 2000 --ues-per-slice 5 --rbgs 64 --rbg-size 8
