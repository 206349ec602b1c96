#!/bin/bash
# Round-2 parity campaign beyond the pytest suite (GPU box): random shapes, queue model fuzz, long soaks on the speculative path.
set -x
timeout 600 python tools/fuzz_parity.py 200 40
timeout 600 python tools/fuzz_queues.py 1000 30
for a in "--sched 9 --jit 1" "--sched 9 --jit 0" "--sched 8 --jit 1" "--sched 9 --jit 1 --rbgs 64 --rbg-size 8 --cells 16" "--sched 8 --jit 1 --rbgs 64 --rbg-size 8 --cells 16" "--sched 101 --jit 1" "--sched 103 --jit 1 --ttis 2000" "--sched 9 --jit 1 --launch 37" "--sched 9 --jit 1 --threads 256" "--sched 9 --jit 0 --threads 128 --phy 1" "--sched 9 --jit 1 --ues-per-slice 40"; do
  timeout 300 python tools/soak.py $a | grep SOAK
done
