#!/bin/bash
# Round 5: extra parity runs for the rewritten sampler (scheduler 11) on the final device sources: random shapes and soaks
cd $GRAFT_REPO_ROOT
python -c "import radiosaber_amd as rs; print('device sources', rs.device_source_hash())"
timeout 1500 python tools/fuzz_sampler.py 17000 80
for a in "--sched 11 --jit 1 --cells 8 --ttis 3000" "--sched 11 --jit 1 --rbgs 64 --rbg-size 8 --cells 8 --ttis 2000" "--sched 11 --jit 0 --rbgs 64 --rbg-size 8 --cells 4 --ttis 1000" \
         "--sched 11 --jit 1 --rbgs 64 --rbg-size 8 --ues-per-slice 10 --cells 8 --ttis 3000 --phy 1 --launch 37" "--sched 11 --jit 1 --slices 3 --ues-per-slice 20 --cells 8 --ttis 3000" \
         "--sched 11 --jit 1 --ues-per-slice 50 --cells 4 --ttis 1000" "--sched 11 --jit 1 --slices 4 --ues-per-slice 70 --cells 4 --ttis 1000 --phy 1" \
         "--sched 11 --jit 1 --threads 64 --rbgs 64 --rbg-size 8 --ues-per-slice 5 --cells 8 --ttis 2000" "--sched 11 --jit 1 --threads 256 --launch 41 --cells 8 --ttis 2000"; do
  timeout 600 python tools/soak.py $a | grep SOAK
done
