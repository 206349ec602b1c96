#!/bin/bash
# Round-6 parity campaign beyond the pytest suite (GPU box), on the round's final device sources: round 5's list (random shapes, the
# queue model, lean builds, soaks -- every batch now self-checks its run-time builds before its first launch, that is part of what runs
# here) plus the drop-in fuzz (a specialised, self-checked context with cqi_epoch against a built-in twin and the oracle).
set -x
python -c "import radiosaber_amd as rs; print('device sources', rs.device_source_hash(), '| compiler', rs.jit_compiler_identity())"
timeout 1500 python tools/fuzz_parity.py 17000 150
timeout 900 python tools/fuzz_queues.py 21000 60
timeout 1200 python tools/fuzz_lean.py 23000 60
timeout 1500 python tools/fuzz_dropin.py 27000 120 30
timeout 900 python tools/fuzz_sampler.py 29000 40
for a in "--sched 9 --jit 1 --ttis 8000" "--sched 9 --jit 0" "--sched 8 --jit 1 --ttis 8000" "--sched 9 --jit 1 --rbgs 64 --rbg-size 8 --cells 16" \
         "--sched 7 --jit 1 --ttis 20000" "--sched 7 --jit 1 --ttis 8000 --phy 1 --launch 37" "--sched 7 --jit 1 --rbgs 64 --rbg-size 8 --ttis 8000" \
         "--sched 7 --jit 1 --ues-per-slice 50 --ttis 8000" "--sched 1 --jit 1 --ttis 8000" "--sched 1 --jit 1 --ues-per-slice 50 --ttis 8000 --phy 1" \
         "--sched 103 --jit 1 --ttis 4000" "--sched 101 --jit 1" "--sched 8 --jit 1 --launch 41 --phy 1" "--sched 9 --jit 1 --launch 37" "--sched 9 --jit 1 --threads 256" \
         "--sched 9 --jit 1 --ues-per-slice 50 --cells 16 --phy 1" "--sched 9 --jit 1 --rbgs 64 --rbg-size 8 --cells 8 --threads 640 --ttis 2000" \
         "--sched 11 --jit 1 --cells 8 --ttis 1000" "--sched 9 --jit 1 --slices 40 --ues-per-slice 3 --cells 8 --ttis 4000"; do
  timeout 600 python tools/soak.py $a | grep SOAK
done
