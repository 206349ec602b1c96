#!/bin/bash
# Round-5 parity campaign beyond the pytest suite (GPU box): random shapes, queue-model fuzz, long soaks -- on the round's final device
# sources: the rewritten counting sorts, the heap-sort fallback on whole waves, the NVS carve keyed on the longest window, the trimmed
# carve of schedulers 1 / 7 / 11 (round 4's list of shapes, plus ragged NVS batches and a 600-UE per-flow PF cell).
set -x
python -c "import radiosaber_amd as rs; print('device sources', rs.device_source_hash())"
timeout 1500 python tools/fuzz_parity.py 7000 150
timeout 900 python tools/fuzz_queues.py 11000 60
timeout 1200 python tools/fuzz_lean.py 13000 60
for a in "--sched 9 --jit 1 --ttis 8000" "--sched 9 --jit 0" "--sched 8 --jit 1 --ttis 8000" "--sched 9 --jit 1 --rbgs 64 --rbg-size 8 --cells 16" \
         "--sched 7 --jit 1 --ttis 20000" "--sched 7 --jit 1 --ttis 8000 --phy 1 --launch 37" "--sched 7 --jit 1 --rbgs 64 --rbg-size 8 --ttis 8000" \
         "--sched 7 --jit 1 --ues-per-slice 50 --ttis 8000" "--sched 7 --jit 1 --slices 3 --ues-per-slice 40 --ttis 8000" "--sched 7 --jit 1 --threads 128 --ttis 8000" \
         "--sched 1 --jit 1 --ttis 8000" "--sched 1 --jit 1 --ues-per-slice 50 --ttis 8000 --phy 1" "--sched 1 --jit 1 --ues-per-slice 50 --launch 41 --threads 256" \
         "--sched 103 --jit 1 --ttis 4000" "--sched 103 --jit 1 --rbgs 64 --rbg-size 8 --cells 16 --ttis 2000" "--sched 103 --jit 0 --slices 40 --ues-per-slice 5 --ttis 2000" \
         "--sched 101 --jit 1" "--sched 8 --jit 1 --launch 41 --phy 1" "--sched 9 --jit 1 --launch 37" "--sched 9 --jit 1 --threads 256" \
         "--sched 9 --jit 1 --ues-per-slice 50 --cells 16 --phy 1" "--sched 8 --jit 1 --slices 64 --ues-per-slice 7 --cells 8" \
         "--sched 9 --jit 1 --rbgs 64 --rbg-size 8 --cells 8 --threads 640 --ttis 2000" \
         "--sched 1 --jit 1 --ues-per-slice 30 --rbgs 64 --rbg-size 8 --cells 8 --ttis 4000" "--sched 11 --jit 1 --cells 8 --ttis 1000" \
         "--sched 9 --jit 1 --slices 40 --ues-per-slice 3 --cells 8 --ttis 4000" "--sched 9 --jit 1 --rbgs 64 --rbg-size 8 --cells 8 --threads 256 --ttis 2000"; do
  timeout 600 python tools/soak.py $a | grep SOAK
done
