cd $GRAFT_REPO_ROOT
export RS_HIP_LIB=radiosaber_amd/libradiosaber_hip_stamps.so
RS_JIT_EXTRA="-DRS_STAMPS" python3 tools/phase_stamps.py --jit --ttis 400 --sched 11 --rbgs 64 --rbg-size 8 2>&1 | grep -v "^    -" | head -40
RS_JIT_EXTRA="-DRS_STAMPS" python3 tools/phase_stamps.py --jit --ttis 400 --sched 7 --rbgs 64 --rbg-size 8 2>&1 | grep -v "^    -" | head -30
