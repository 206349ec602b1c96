#!/bin/bash
# Sanitizers on the CPU side -- run HERE (no GPU; GPU-side ASan is not available on this pool):
#   1. oracle/librs_oracle_asan.so                    the restatement,               -O1 -g -fsanitize=address,undefined
#   2. radiosaber_amd/libradiosaber_hip_hostasan.so   the library's HOST side (rs_api.cpp + rs_jit.cpp compiled by g++, the kernel
#                                                     launchers stubbed out: tests/csrc/host_launch_stubs.cpp), same flags -- covers
#                                                     rs_trace_*, rs_internet_flow_arrivals, rs_link_tables, config validation, the
#                                                     checked create functions, the hiprtc option handling
#   3. tests/csrc/{sort_emul_check,umap_emul_check}.cpp   the host+device sort / unordered_map emulation against the real containers
# then `pytest -m "not gpu"` with both libraries selected and libasan preloaded into python.  Output: profiles/r06_sanitizers.log
set -u
R=$(cd "$(dirname "$0")/.." && pwd); cd "$R"
LOG=${1:-$R/profiles/r06_sanitizers.log}
SAN="-O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer"
ASAN_LIB=$(g++ -print-file-name=libasan.so)
STDCXX_LIB=$(g++ -print-file-name=libstdc++.so)  # preloaded too: ASan resolves __cxa_throw when it starts, and python itself does not link libstdc++
{
echo "# tools/sanitize_cpu.sh  $(date -u +%Y-%m-%dT%H:%MZ)  g++ $(g++ -dumpversion), flags: $SAN"
echo "## 1. oracle"; make -C oracle asan 2>&1 | tail -2
echo "## 2. host side of the library"
python3 -c "from radiosaber_amd import build; build.write_tables_inc(); build.write_link_pinned_inc(); build.write_jit_sources()"
g++ -std=c++17 $SAN -ffp-contract=off -fno-fast-math -fPIC -shared -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude \
    radiosaber_amd/csrc/rs_api.cpp radiosaber_amd/csrc/rs_jit.cpp tests/csrc/host_launch_stubs.cpp \
    -L/opt/rocm/lib -lamdhip64 -lhiprtc -Wl,-rpath,/opt/rocm/lib -o radiosaber_amd/libradiosaber_hip_hostasan.so && echo "built radiosaber_amd/libradiosaber_hip_hostasan.so"
echo "## 3. emulation checks (C++ programs)"
for t in sort_emul_check umap_emul_check; do
  g++ -std=c++17 $SAN -o /tmp/${t}_san tests/csrc/$t.cpp && /tmp/${t}_san > /tmp/${t}_san.out 2>&1; echo "$t rc $? : $(tail -1 /tmp/${t}_san.out)"
done
echo "## 4. pytest -m 'not gpu' on the sanitized libraries (libasan preloaded; leak check off: CPython itself leaks at exit)"
LD_PRELOAD="$ASAN_LIB $STDCXX_LIB" ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
  RS_ORACLE_LIB=$R/oracle/librs_oracle_asan.so RS_HIP_LIB=$R/radiosaber_amd/libradiosaber_hip_hostasan.so \
  python3 -m pytest tests -q -s -m "not gpu" -p no:cacheprovider 2>&1 | grep -v "^cpu_baseline\|^$" | tail -40; echo "pytest exit code ${PIPESTATUS[0]}"
echo "## sanitizer reports in this log: $(grep -c 'ERROR: AddressSanitizer\|runtime error:' "$LOG.tmp" 2>/dev/null || echo 0)"
} > "$LOG.tmp" 2>&1
n=$(grep -c 'ERROR: AddressSanitizer\|runtime error:' "$LOG.tmp")
sed -i "s/^## sanitizer reports in this log: .*/## sanitizer reports in this log: $n/" "$LOG.tmp"
mv "$LOG.tmp" "$LOG"; tail -25 "$LOG"
