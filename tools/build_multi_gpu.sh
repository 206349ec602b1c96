#!/bin/bash
# Build tools/rs_multi_gpu (C++ host over the C ABI, RCCL called directly); needs radiosaber_amd/libradiosaber_hip.so (python -m radiosaber_amd.build).
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
HIPCC=${HIPCC:-$(command -v hipcc || echo /opt/rocm/bin/hipcc)}
"$HIPCC" -O2 -std=c++17 "$R/tools/rs_multi_gpu.cpp" -I"$R/include" -L"$R/radiosaber_amd" -lradiosaber_hip -L/opt/rocm/lib -lrccl \
  -Wl,-rpath,'$ORIGIN/../radiosaber_amd' -Wl,-rpath,/opt/rocm/lib -o "$R/tools/rs_multi_gpu"
echo "$R/tools/rs_multi_gpu"
