#!/bin/bash
# Round 5, last lease: the parity campaign, the phase stamps (stamps library rebuilt on the final sources), smoke and the default bench line
# with the committed records in place (no stale flags)
R=$GRAFT_REPO_ROOT; cd $R
bash tools/campaign_r05.sh > gpurun_out/r05_campaign.log 2>&1
grep -c "SOAK OK" gpurun_out/r05_campaign.log; grep -v "^+" gpurun_out/r05_campaign.log | grep -E "fuzz|device sources"
export RS_HIP_LIB_SAVE=$RS_HIP_LIB
( export RS_HIP_LIB=radiosaber_amd/libradiosaber_hip_stamps.so
  RS_JIT_EXTRA="-DRS_STAMPS" python3 tools/phase_stamps.py --jit --ttis 400
  RS_JIT_EXTRA="-DRS_STAMPS -DRS_STAMPS_HOLD" python3 tools/phase_stamps.py --jit --hold --ttis 400 | grep "hold:"
  RS_JIT_EXTRA="-DRS_STAMPS" python3 tools/phase_stamps.py --jit --ttis 400 --sched 8
  RS_JIT_EXTRA="-DRS_STAMPS" python3 tools/phase_stamps.py --jit --ttis 400 --rbgs 64 --rbg-size 8 ) 2>&1 | grep -v "^    -" > gpurun_out/r05_stamps_final.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
python3 bench.py > gpurun_out/r05_bench_default.log 2> gpurun_out/r05_bench_default.err; cut -c1-400 gpurun_out/r05_bench_default.log
