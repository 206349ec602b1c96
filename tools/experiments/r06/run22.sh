#!/bin/bash
# round 6, second session: the round's spare GPU minutes on the final sources -- more seeds of the fuzzers, the sort's position-slot
# fuzz, two long soaks at 64 RBGs
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
{
python3 -c "import radiosaber_amd as rs; print('device sources', rs.device_source_hash(), '| compiler', rs.jit_compiler_identity())"
timeout 900 python3 tools/fuzz_sort_slots.py 61000 120 | tail -4
timeout 900 python3 tools/fuzz_parity.py 41000 120
timeout 600 python3 tools/fuzz_lean.py 43000 50
timeout 600 python3 tools/fuzz_queues.py 45000 30
timeout 600 python3 tools/fuzz_dropin.py 47000 40 30 | tail -1
for a in "--sched 9 --jit 1 --rbgs 64 --rbg-size 8 --cells 16 --ttis 8000" "--sched 10 --jit 1 --rbgs 64 --rbg-size 8 --cells 8 --ttis 4000" "--sched 9 --jit 1 --ttis 20000 --cells 8" "--sched 10 --jit 1 --ttis 8000 --cells 8"; do
  timeout 600 python3 tools/soak.py $a | grep SOAK
done
} > gpurun_out/r06_more_fuzz.log 2>&1
tail -12 gpurun_out/r06_more_fuzz.log
