export RS_HIP_LIB=radiosaber_amd/libradiosaber_hip_stamps.so
echo "=== 512 cells, spec"; RS_JIT_EXTRA="-DRS_STAMPS" timeout 200 python tools/phase_stamps.py --jit --cells 512
echo "=== 256 cells, spec"; RS_JIT_EXTRA="-DRS_STAMPS" timeout 200 python tools/phase_stamps.py --jit --cells 256
echo "=== 256 cells, no spec"; RS_JIT_EXTRA="-DRS_STAMPS -DRS_NO_SPEC" timeout 200 python tools/phase_stamps.py --jit --cells 256
echo "=== 512 cells, no spec"; RS_JIT_EXTRA="-DRS_STAMPS -DRS_NO_SPEC" timeout 200 python tools/phase_stamps.py --jit --cells 512
