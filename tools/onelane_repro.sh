#!/bin/bash
# Rebuild round 4's wrong kernel from git history and point at the instructions that make it wrong (profiles/r05_onelane.md).
#   tools/onelane_repro.sh [workdir]          CPU only: hipcc -S of the good and the bad source + tools/lint_exec_restore.py
#   then, on a GPU box:  cd <workdir>/tree && RS_HIP_LIB=$PWD/lib_bad.so python repro.py     (12 of 12 calls differ from the oracle)
#                                             RS_HIP_LIB=$PWD/lib_good.so python repro.py    (0 of 12)
# good = commit 0adb0e5 (SubOpt's unordered_map walk on every lane), bad = the same tree with that one line back to
#        `if (lane == 0 && more_mask && fewer_mask) n_ord = rs_umap_order(...)`.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
W=${1:-/tmp/onelane_repro}
rm -rf "$W" && mkdir -p "$W/tree"
git -C "$ROOT" archive 0adb0e5 | tar -x -C "$W/tree"
cp "$ROOT/tools/microbench/onelane_repro.py" "$W/tree/repro.py"
cd "$W/tree"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -Wno-unused-function -Wno-missing-braces -mllvm -amdgpu-atomic-optimizer-strategy=None -Iinclude"
python -m radiosaber_amd.build --force > /dev/null && cp radiosaber_amd/libradiosaber_hip.so lib_good.so
hipcc $FLAGS --cuda-device-only -S -o "$W/good.s" radiosaber_amd/csrc/rs_kernels.hip 2> /dev/null
sed -i 's/  if (more_mask \&\& fewer_mask) n_ord = rs_umap_order(/  if (lane == 0 \&\& more_mask \&\& fewer_mask) n_ord = rs_umap_order(/' radiosaber_amd/csrc/rs_interslice.h
grep -q "lane == 0 && more_mask && fewer_mask" radiosaber_amd/csrc/rs_interslice.h
python -m radiosaber_amd.build --force > /dev/null && cp radiosaber_amd/libradiosaber_hip.so lib_bad.so
hipcc $FLAGS --cuda-device-only -S -o "$W/bad.s" radiosaber_amd/csrc/rs_kernels.hip 2> /dev/null
make -C oracle > /dev/null 2>&1 || true
echo "== good:"; python "$ROOT/tools/lint_exec_restore.py" "$W/good.s" || true
echo "== bad:";  python "$ROOT/tools/lint_exec_restore.py" "$W/bad.s" || true
