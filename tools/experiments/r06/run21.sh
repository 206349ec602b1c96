#!/bin/bash
# round 6, second session: scheduler 11 under clang 22 -- does the SLP vectoriser (it packs the sampler's four draw bytes into
# v_pk_mul_lo_u16 behind one ds_read_b32) or the load / store vectoriser explain the 24 % against clang 20?
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06
export TMPDIR=/tmp
out=gpurun_out/r06/run21_sched11_flags.log
one() { python3 bench.py --no-cpu-baseline --no-streamed --no-cells1024 --no-r64 --steps 6 --allow-variant "$@" 2>&1 | python3 -c "
import sys, json, os
ls = [l for l in sys.stdin.read().strip().splitlines() if l.startswith('{')]
if not ls: print('%-30s %-60s FAILED' % (' '.join(sys.argv[1:]), os.environ.get('RS_JIT_EXTRA', ''))); sys.exit(0)
d = json.loads(ls[-1])
print('%-30s %-10s %-60s %.2f M TTIs/s  %.3f ms' % (' '.join(sys.argv[1:]) or '(headline)', d['compiler'].split(' clang ')[1][:9], os.environ.get('RS_JIT_EXTRA', ''),
      d['value'] / 1e6, sum(d['kernel_ms_per_launch']) / len(d['kernel_ms_per_launch'])))" "$@"; }
{
for args in "--sched 11" "--sched 11 --rbgs 64 --rbg-size 8"; do
  one $args
  RS_JIT_EXTRA="-fno-slp-vectorize" one $args
  RS_JIT_EXTRA="-mllvm -vectorize-slp=false" one $args
  RS_JIT_EXTRA="-mllvm -amdgpu-load-store-vectorizer=false" one $args
  RS_JIT_EXTRA="-fno-slp-vectorize -mllvm -amdgpu-load-store-vectorizer=false" one $args
  RS_SYSTEM_COMGR=0 one $args
done
for args in "" "--rbgs 64 --rbg-size 8" "--sched 8" "--sched 7" "--sched 1"; do
  one $args
  RS_JIT_EXTRA="-fno-slp-vectorize" one $args
done
} > $out 2>&1
cat $out
