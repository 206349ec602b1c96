#!/bin/bash
# Round 4, GPU run 36: lean form of the specialised drop-in kernel -- latency with / without (RS_JIT_LEAN=0), then the drop-in parity tests
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_run36; mkdir -p $O; cd ..
echo "--- lean"; tools/dropin_latency 2000 2>&1 | grep specialised
echo "--- RS_JIT_LEAN=0"; RS_JIT_LEAN=0 tools/dropin_latency 2000 2>&1 | grep specialised
echo "--- lean"; tools/dropin_latency 2000 2>&1 | grep specialised
python -m pytest tests -m gpu -q -k "drop_in or dropin or adapter or specialised" 2>&1 | tail -3
