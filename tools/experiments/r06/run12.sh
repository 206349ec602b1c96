#!/bin/bash
# round 6, second session: the working tree's library against the previous commit's (built into scratch_prev/ by
#   git archive HEAD radiosaber_amd include | tar -x -C scratch_prev && (cd scratch_prev && python -m radiosaber_amd.build)
# ): micro-benchmark of the sort, the sort parity tests on the working tree, alternating bench runs on one lease.
# usage: run12.sh <tag> [bench argument sets, ';'-separated]
cd "$GRAFT_REPO_ROOT" || exit 1
tag=${1:-ab}
mkdir -p gpurun_out/r06
export TMPDIR=/tmp
out=gpurun_out/r06/run12_$tag.log
{
cd tools/microbench
for n in 1280 500; do
  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -DMB_N=$n -I. -I../../radiosaber_amd/csrc -I../../include -o /tmp/mb_sort_$n mb_sort.hip 2>/dev/null
  echo "== mb_sort N=$n (working tree)"
  k=keys_r64.bin; [ $n = 500 ] && k=keys_r25.bin
  timeout 120 /tmp/mb_sort_$n $k | grep "workgroup levels"
done
cd "$GRAFT_REPO_ROOT"
echo "== parity tests"
timeout 900 python3 -m pytest tests/test_sort_killers.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
one() { python3 bench.py --no-cpu-baseline --no-streamed --no-cells1024 --no-r64 --steps 8 --allow-variant "$@" 2>/dev/null | python3 -c "
import sys, json, os
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-36s %-14s %.2f M TTIs/s  %.3f ms' % (' '.join(sys.argv[1:]) or '(headline)', 'prev' if os.environ.get('RS_HIP_LIB') else 'tree',
      d['value'] / 1e6, sum(d['kernel_ms_per_launch']) / len(d['kernel_ms_per_launch'])))" "$@"; }
for rep in 1 2; do
  for args in "" "--rbgs 64 --rbg-size 8" "--sched 10 --rbgs 64 --rbg-size 8" "--sched 10" "--ues-per-slice 50"; do
    one $args
    RS_HIP_LIB=$GRAFT_REPO_ROOT/scratch_prev/radiosaber_amd/libradiosaber_hip.so one $args
  done
done
} > $out 2>&1
cat $out
