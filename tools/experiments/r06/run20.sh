#!/bin/bash
# round 6, second session: scheduler 11 under the two compilers of the image (RS_SYSTEM_COMGR=0: the torch wheel's clang 20, as every bench
# run before the first session's last commit; default: /opt/rocm's clang 22, the compiler of the C++ hosts and of the tests)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06
export TMPDIR=/tmp
out=gpurun_out/r06/run20_sched11_compilers.log
one() { python3 bench.py --no-cpu-baseline --no-streamed --no-cells1024 --no-r64 --steps 6 --allow-variant "$@" 2>/dev/null | python3 -c "
import sys, json, os
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-36s %-22s %-34s %.2f M TTIs/s  %.3f ms' % (' '.join(sys.argv[1:]) or '(headline)', d['compiler'].split(' clang ')[1][:9], os.environ.get('RS_JIT_EXTRA', '') + ' ' + os.environ.get('RS_JIT_SCHED_STRATEGY', ''),
      d['value'] / 1e6, sum(d['kernel_ms_per_launch']) / len(d['kernel_ms_per_launch'])))" "$@"; }
{
for rep in 1 2; do
  for args in "--sched 11" "--sched 11 --rbgs 64 --rbg-size 8" "--sched 11 --ues-per-slice 50"; do
    one $args
    RS_SYSTEM_COMGR=0 one $args
    RS_JIT_SCHED_STRATEGY=default one $args
    RS_JIT_SCHED_STRATEGY=iterative-ilp one $args
    RS_JIT_SCHED_STRATEGY=max-ilp one $args
  done
done
} > $out 2>&1
cat $out
