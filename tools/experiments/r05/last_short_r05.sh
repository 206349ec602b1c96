#!/bin/bash
# Round 5, the very last lease (device sources 7d7b9dfecc1d588c: the campaign ran on ff3f769241f35787, one carve rule earlier; here the
# sampler fuzz, the phase stamps, smoke and the default bench line with the committed records in place)
R=$GRAFT_REPO_ROOT; cd $R
python -c "import radiosaber_amd as rs; print('device sources', rs.device_source_hash())"
timeout 200 python tools/fuzz_sampler.py 19000 60 2>&1 | tail -1
timeout 300 python tools/fuzz_parity.py 7400 40 2>&1 | tail -1
( export RS_HIP_LIB=radiosaber_amd/libradiosaber_hip_stamps.so
  RS_JIT_EXTRA="-DRS_STAMPS" python3 tools/phase_stamps.py --jit --ttis 400
  RS_JIT_EXTRA="-DRS_STAMPS -DRS_STAMPS_HOLD" python3 tools/phase_stamps.py --jit --hold --ttis 400 | grep "hold:"
  RS_JIT_EXTRA="-DRS_STAMPS" python3 tools/phase_stamps.py --jit --ttis 400 --sched 8
  RS_JIT_EXTRA="-DRS_STAMPS" python3 tools/phase_stamps.py --jit --ttis 400 --rbgs 64 --rbg-size 8 ) 2>&1 | grep -v "^    -" > gpurun_out/r05_stamps_final.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python3 bench.py > gpurun_out/r05_bench_default.log 2> gpurun_out/r05_bench_default.err; cut -c1-330 gpurun_out/r05_bench_default.log
