#!/bin/bash
# Round-3 parity campaign beyond the pytest suite (GPU box): random shapes, queue model fuzz, long soaks on the held-winner paths
# (shape-specialised kernels of up to 32 RBGs) and on round 2's paths that remain (64 RBGs: speculation; built-in kernels).
set -x
timeout 900 python tools/fuzz_parity.py 5000 100
timeout 900 python tools/fuzz_queues.py 9000 60
for a in "--sched 9 --jit 1 --ttis 8000" "--sched 9 --jit 0" "--sched 8 --jit 1 --ttis 8000" "--sched 9 --jit 1 --rbgs 64 --rbg-size 8 --cells 16" \
         "--sched 8 --jit 1 --rbgs 64 --rbg-size 8 --cells 16" "--sched 101 --jit 1" "--sched 103 --jit 1 --ttis 2000" "--sched 9 --jit 1 --launch 37" \
         "--sched 8 --jit 1 --launch 41 --phy 1" "--sched 9 --jit 1 --threads 256" "--sched 9 --jit 0 --threads 128 --phy 1" "--sched 9 --jit 1 --ues-per-slice 40" \
         "--sched 8 --jit 1 --ues-per-slice 50 --cells 16" "--sched 9 --jit 1 --ues-per-slice 50 --cells 16 --phy 1" "--sched 8 --jit 1 --slices 64 --ues-per-slice 7 --cells 8" \
         "--sched 9 --jit 1 --slices 3 --ues-per-slice 60 --cells 8" "--sched 9 --jit 1 --rbgs 32 --rbg-size 3 --cells 16"; do
  timeout 400 python tools/soak.py $a | grep SOAK
done
