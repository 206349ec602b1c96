#!/bin/bash
# round 6, second session: drop-in contexts whose specialised one-TTI kernels run 1 024 threads where the sort has more than one
# position per lane at 512 (rs_ctx_specialize): the drop-in tests, the fuzz, and tools/dropin_latency with RS_DROPIN_THREADS=512 / default
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06
export TMPDIR=/tmp
out=gpurun_out/r06/run18_dropin_threads.log
{
echo "== GPU tests that use drop-in contexts"
timeout 1200 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_round4.py tests/test_gpu_round5.py tests/test_gpu_round6.py -m gpu -x -q -k "drop or dropin or context or ctx or specialis or epoch or thread" 2>&1 | tail -4
echo "== fuzz_dropin, 12 seeds"
timeout 600 python3 tools/fuzz_dropin.py 700 16 20 2>&1 | tail -3
for t in 512 default; do
  echo "== dropin_latency, RS_DROPIN_THREADS=$t"
  if [ $t = default ]; then ./tools/dropin_latency 2000 2>&1 | grep -i "specialised"; else RS_DROPIN_THREADS=$t ./tools/dropin_latency 2000 2>&1 | grep -i "specialised"; fi
done
} > $out 2>&1
cat $out
