#!/bin/bash
# check_integration.sh -- the kept, reproducible check of integration/*.{h,cpp}.
#
#   tools/check_integration.sh <reference-tree> [scratch-dir]        compile + link one LTE-Sim with the GPU schedulers in it
#   tools/check_integration.sh --patch-only <reference-tree> [scratch-dir]   only apply the edits of enodeb-cases.inc to a
#                                                                   scratch copy of the three files they touch (runs anywhere)
#
# Nothing is written into <reference-tree>: its src/ is copied to the scratch directory first.  The compile step needs what the
# reference itself needs and this repository's image lacks -- jsoncpp (<jsoncpp/json/json.h>, -ljsoncpp) -- and the generated
# src/load-parameters.h (made in the scratch copy by the reference's own recipe, CONFIG/make_load-parameter-file.sh: the
# concatenation of CONFIG/load-parameters-start, the path line and CONFIG/global_config).  No stand-ins: without jsoncpp the
# script stops with exit code 77 ("not checkable here").  Flags = Debug/src/**/subdir.mk of the reference
# (g++ -O0 -g3 -Wall -Wno-unused-variable -c -fmessage-length=0) plus -I<repo>/include; link = Debug/makefile's line plus
# -L<repo>/radiosaber_amd -lradiosaber_hip.  With a GPU visible the script also runs the CPU scheduler 9 and its GPU twin 29 for
# 0.05 simulated seconds and compares stdout/stderr byte for byte (only the one printed pointer may differ).
set -euo pipefail
REPO=$(cd "$(dirname "$0")/.." && pwd)
PATCH_ONLY=0
if [ "${1:-}" = "--patch-only" ]; then PATCH_ONLY=1; shift; fi
REF=${1:?usage: $0 [--patch-only] <reference-tree> [scratch-dir]}
W=${2:-$(mktemp -d /tmp/rs_integration.XXXXXX)}
[ -d "$REF/src/protocolStack/mac/packet-scheduler" ] || { echo "$REF is not a RadioSaber / LTE-Sim tree" >&2; exit 2; }
mkdir -p "$W"

SCHED_DIR=src/protocolStack/mac/packet-scheduler
if [ $PATCH_ONLY = 1 ]; then
  mkdir -p "$W/src/device" "$W/src/scenarios"
  cp "$REF/src/device/ENodeB.h" "$REF/src/device/ENodeB.cpp" "$W/src/device/"
  cp "$REF/src/scenarios/single-cell-with-interference.h" "$W/src/scenarios/"
else
  if ! echo '#include <jsoncpp/json/json.h>' | g++ -x c++ -fsyntax-only - 2>/dev/null; then
    echo "jsoncpp (<jsoncpp/json/json.h>) is not installed: integration/ is not checkable in this image (no stand-ins are written)" >&2
    exit 77
  fi
  cp -r "$REF/src" "$REF/CONFIG" "$W/"
  if [ -d "$REF/cqi-traces-noise0" ]; then # a directory of links + the chosen mapping file (run_backlogged.sh:7-8)
    mkdir -p "$W/cqi-traces-noise0"
    ln -sf "$REF"/cqi-traces-noise0/ue*.log "$W/cqi-traces-noise0/"
    cp "$REF/cqi-traces-noise0/mapping0.config" "$W/cqi-traces-noise0/mapping.config"
  fi
  (cd "$W" && { rm -f src/load-parameters.h; cat CONFIG/load-parameters-start > src/load-parameters.h; \
     echo "static std::string path (\"$W/\");" >> src/load-parameters.h; cat CONFIG/global_config >> src/load-parameters.h; })
  cp "$REPO"/integration/*.h "$REPO"/integration/*.cpp "$W/$SCHED_DIR/"
fi

# ---- the edits of integration/enodeb-cases.inc, applied mechanically
# 1. enum DLSchedulerType: the GPU twins after the last entry
python3 - "$W" "$REPO" <<'EOF'
import re, sys
from pathlib import Path
W, REPO = Path(sys.argv[1]), Path(sys.argv[2])
h = W / "src/device/ENodeB.h"
t = h.read_text()
m = re.search(r"enum\s+DLSchedulerType\s*\{([^}]*)\}", t)
assert m, "ENodeB.h: enum DLSchedulerType not found"
twins = ["DLScheduler_GPU_PF", "DLScheduler_GPU_NVS", "DLScheduler_GPU_NVS_NONGREEDY", "DLScheduler_GPU_SEQUENTIAL",
         "DLScheduler_GPU_SUBOPT", "DLScheduler_GPU_UpperBound", "DLScheduler_GPU_MAXCELL", "DLScheduler_GPU_VOGEL"]
body = m.group(1).rstrip()
t = t[:m.start(1)] + body + ",\n    " + ",\n    ".join(twins) + "\n  " + t[m.end(1):]
h.write_text(t)

# 2. ENodeB.cpp: the three headers + the cases before SetDLScheduler's `default:`
c = W / "src/device/ENodeB.cpp"
t = c.read_text()
inc = "".join(f'#include "../protocolStack/mac/packet-scheduler/{n}"\n'
              for n in ("downlink-gpu-scheduler.h", "dl-gpu-pf-packet-scheduler.h", "downlink-gpu-nvs-scheduler.h"))
anchor = '#include "../protocolStack/mac/packet-scheduler/downlink-transport-scheduler.h"\n'
assert anchor in t, "ENodeB.cpp: include anchor not found"
t = t.replace(anchor, anchor + inc, 1)
cases = (REPO / "integration/enodeb-cases.inc").read_text()
cases = cases[cases.index("*/") + 2:].strip("\n") + "\n\n"
start = t.index("ENodeB::SetDLScheduler")
d = re.compile(r"^[ \t]*default:", re.M).search(t, start)
assert d, "ENodeB.cpp: SetDLScheduler's default: not found"
t = t[:d.start()] + cases + t[d.start():]
c.write_text(t)

# 3. CLI numbers 21 / 27 / 28 / 29 / 30 / 31 = the GPU twins of 1 / 7 / 8 / 9 / 10 / 11
s = W / "src/scenarios/single-cell-with-interference.h"
t = s.read_text()
start = t.index("switch (sched_type)")
d = re.compile(r"^[ \t]*default:", re.M).search(t, start)
cli = {21: "DLScheduler_GPU_PF", 27: "DLScheduler_GPU_NVS", 28: "DLScheduler_GPU_SEQUENTIAL", 29: "DLScheduler_GPU_MAXCELL",
       30: "DLScheduler_GPU_UpperBound", 31: "DLScheduler_GPU_NVS_NONGREEDY"}
add = "".join(f"    case {k}:\n      downlink_scheduler_type = ENodeB::{v};\n      break;\n" for k, v in cli.items())
t = t[:d.start()] + add + t[d.start():]
s.write_text(t)
print("patched:", h, c, s)
EOF
if [ $PATCH_ONLY = 1 ]; then echo "$W"; exit 0; fi

# ---- compile everything the reference's Debug/ makefiles compile, with its flags, then link
[ -f "$REPO/radiosaber_amd/libradiosaber_hip.so" ] || { echo "build the library first: python -m radiosaber_amd.build" >&2; exit 2; }
mkdir -p "$W/obj"
FLAGS="-O0 -g3 -Wall -Wno-unused-variable -c -fmessage-length=0 -I$REPO/include"
cd "$W"
find src \( -name '*.cpp' -o -name '*.cc' \) -not -path '*make_fast_fading*' -print0 |
  xargs -0 -P "$(nproc)" -I{} sh -c 'o=obj/$(echo "$1" | tr / _).o; g++ '"$FLAGS"' -o "$o" "$1" || exit 255' _ {}
g++ -o LTE-Sim obj/*.o -ljsoncpp -L"$REPO/radiosaber_amd" -lradiosaber_hip -Wl,-rpath,"$REPO/radiosaber_amd"
echo "linked: $W/LTE-Sim (integration/*.cpp compiled with the reference's flags)"

# ---- with a GPU: CPU scheduler 9 vs its GPU twin, byte for byte
if [ -d "$W/cqi-traces-noise0" ] && python3 -c "import sys; sys.path.insert(0, '$REPO'); import radiosaber_amd as rs; sys.exit(0 if rs.device_count() > 0 else 1)" 2>/dev/null; then
  CFG="$REF/NSDI23-radiosaber-experiments/exp-fix20slices/5ues/config-pf.json"
  for s in 9 29; do ./LTE-Sim SingleCellWithI 1 $s 1 30 0 0.05 "$CFG" > out.$s 2> err.$s; done
  sed -i -e '/BandwidthManager: 0x/d' -e '/^Scheduler /d' out.9 out.29
  cmp out.9 out.29 && cmp err.9 err.29 && echo "scheduler 29 (GPU) == scheduler 9 (CPU): stdout and stderr identical"
fi
