#!/bin/bash
# round 6, second session: the single-wave finish of the sort as straight-line code against the form of rounds 3-5 (-DRS_FINISH_V1)
# (-DRS_SORT_BRANCHY): the micro-benchmark (cycles per introsort loop, mismatches against std::sort), the sort parity tests, and a
# same-lease alternating A/B of the bench at 25 and 64 RBGs.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06
export TMPDIR=/tmp
out=gpurun_out/r06/run11_finish.log
{
cd tools/microbench
for n in 1280 500; do
  for v in "" "-DRS_FINISH_V1"; do
    hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -DMB_N=$n $v -I. -I../../radiosaber_amd/csrc -I../../include -o /tmp/mb_sort_$n$v mb_sort.hip 2>/dev/null
    echo "== mb_sort N=$n ${v:-staged}"
    k=keys_r64.bin; [ $n = 500 ] && k=keys_r25.bin
    timeout 120 /tmp/mb_sort_$n$v $k | grep "workgroup levels"
  done
done
cd "$GRAFT_REPO_ROOT"
echo "== parity tests"
timeout 900 python3 -m pytest tests/test_sort_killers.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
one() { python3 bench.py --no-cpu-baseline --no-streamed --no-cells1024 --no-r64 --steps 8 --allow-variant "$@" 2>/dev/null | python3 -c "
import sys, json, os
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-30s %-28s %.2f M TTIs/s  %.3f ms' % (' '.join(sys.argv[1:]) or '(headline)', 'EXTRA=' + os.environ.get('RS_JIT_EXTRA', ''),
      d['value'] / 1e6, sum(d['kernel_ms_per_launch']) / len(d['kernel_ms_per_launch'])))" "$@"; }
for rep in 1 2; do
  for args in "" "--rbgs 64 --rbg-size 8" "--sched 10 --rbgs 64 --rbg-size 8" "--sched 10"; do
    one $args
    RS_JIT_EXTRA=-DRS_FINISH_V1 one $args
  done
done
} > $out 2>&1
cat $out
