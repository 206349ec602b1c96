#!/bin/bash
# round 6, second session: phase stamps of the working tree (-DRS_STAMPS library + run-time builds), 25 and 64 RBGs
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06
export TMPDIR=/tmp
export RS_HIP_LIB=radiosaber_amd/libradiosaber_hip_stamps.so
(RS_JIT_EXTRA="-DRS_STAMPS" python3 tools/phase_stamps.py --jit --ttis 400
 RS_JIT_EXTRA="-DRS_STAMPS" python3 tools/phase_stamps.py --jit --ttis 400 --rbgs 64 --rbg-size 8) 2>&1 | grep -v "^    -" > gpurun_out/r06/run13_stamps.log
cat gpurun_out/r06/run13_stamps.log
