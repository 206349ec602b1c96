#!/bin/bash
# usage: tools/pmc_insts.sh <lib.so> <tag> [bench args]   -- instruction counters of the cell kernel for one build
R=$GRAFT_REPO_ROOT; LIB=$1; TAG=$2; shift 2
cd /tmp && export TMPDIR=/tmp
RS_HIP_LIB=$R/radiosaber_amd/$LIB rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY --output-format csv -d $R/gpurun_out/pmc_$TAG -- python3 $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 "$@" > $R/gpurun_out/pmc_$TAG.log 2>&1
cd $R
python3 - <<PY
import csv,glob,collections,json
f=sorted(glob.glob("gpurun_out/pmc_$TAG/*/*_counter_collection.csv"))[-1]
acc=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "rs_cell_kernel" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
line=[l for l in open("gpurun_out/pmc_$TAG.log") if l.startswith("{")][-1]
d=json.loads(line); n=d["config"]["cells_per_gpu"]*d["config"]["ttis_per_step"]
print("$TAG", "us/TTI/cell %.2f" % d["us_per_tti_per_cell"], " per cell-TTI:", {k: round(sum(v)/len(v)/n) for k,v in sorted(acc.items())})
PY
