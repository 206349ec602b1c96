#!/bin/bash
# Round 4, extra parity on the final device sources beyond tools/campaign_r04.sh: fresh seeds for the three fuzzers and soaks of the shapes this
# half of the round changed (per-flow PF on lanes, streamed batches with the grid fetched ahead, the keyed NVS sampler, lean builds throughout).
set -x
python -c "import radiosaber_amd as rs; print('device sources', rs.device_source_hash())"
timeout 1200 python tools/fuzz_lean.py 21000 120
timeout 1200 python tools/fuzz_parity.py 31000 100
timeout 600 python tools/fuzz_queues.py 41000 40
for a in "--sched 1 --jit 1 --ues-per-slice 50 --ttis 12000" "--sched 11 --jit 1 --ttis 3000 --cells 16" "--sched 11 --jit 1 --rbgs 64 --rbg-size 8 --ttis 1500 --cells 8" \
         "--sched 10 --jit 1 --ttis 6000" "--sched 101 --jit 1 --ttis 6000" "--sched 8 --jit 1 --ues-per-slice 50 --ttis 12000" "--sched 9 --jit 1 --ttis 16000" \
         "--sched 9 --jit 1 --ues-per-slice 50 --ttis 8000 --cells 16"; do
  timeout 900 python tools/soak.py $a | grep SOAK
done
