#!/bin/bash
# Round 4, GPU run 17: kPf1 variants (packed multiply, without the second-largest tracking [timing only], item scan)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_run17; mkdir -p $O; cd ..
ab() { # tag, extra, bench args
  local tag=$1 extra=$2; shift 2
  RS_JIT_EXTRA="$extra" timeout 200 python bench.py --allow-variant --no-cpu-baseline --no-r64 --no-streamed --steps 5 --warmup 1 --ttis 4000 "$@" > $O/ab_$tag.log 2>&1
  grep -h '^{' $O/ab_$tag.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%-28s %-36s %.2f M TTIs/s  %.3f us' % (sys.argv[1], sys.argv[2], d['value']/1e6, d['us_per_tti_per_cell']))" "$tag" "[$extra]" || tail -3 $O/ab_$tag.log
}
for v in "" "-DRS_PF1_NO_PK" "-DRS_PF1_NOTOP2" "-DRS_NO_PF1_LANES"; do
ab s1_r25 "$v" --sched 1
ab s1_u1000 "$v" --sched 1 --ues-per-slice 50
ab s1_r64 "$v" --sched 1 --rbgs 64 --rbg-size 8
done
python -m pytest tests -m gpu -x -q -k "sched or sweep or 1000 or random or config or early or prepar" > $O/pytest_s1.log 2>&1; echo "pytest rc $?" >> $O/pytest_s1.log
tail -3 $O/pytest_s1.log
