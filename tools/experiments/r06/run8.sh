#!/bin/bash
# round 6, eighth lease: the same kernel sources built by the wheel's clang 20 (RS_SYSTEM_COMGR=0: the import order decides, torch first)
# and by the system's clang 22 (radiosaber_amd.toolchain maps /opt/rocm's comgr before torch), alternating on one lease
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06
export TMPDIR=/tmp
one() { python3 bench.py --no-cpu-baseline --no-streamed --no-cells1024 --steps 10 "$@" 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-28s %.2f M TTIs/s  %.3f ms per launch  r64 %s  %s MHz  %s | comgr %s' % (' '.join(sys.argv[1:]) or '(default)', d['value'] / 1e6, sum(d['kernel_ms_per_launch']) / len(d['kernel_ms_per_launch']),
      ('%.2f M' % (d['value_r64'] / 1e6)) if 'value_r64' in d else '-', round(d['shader_mhz']), d['compiler'].split(' clang ')[1][:12], d.get('comgr')))" "$@"; }
{
for i in 1 2 3; do
  RS_SYSTEM_COMGR=0 one
  one
done
for s in 8 1 7 10; do
  RS_SYSTEM_COMGR=0 one --sched $s --no-r64
  one --sched $s --no-r64
done
} > gpurun_out/r06/run8_compiler_ab.log 2>&1
timeout 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_round4.py -m gpu -x -q -k "bench" 2>&1 | tail -5 > gpurun_out/r06/run8_tests.log
cat gpurun_out/r06/run8_compiler_ab.log; tail -2 gpurun_out/r06/run8_tests.log
