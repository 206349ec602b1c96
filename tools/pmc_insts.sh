#!/bin/bash
# usage (ON THE GPU BOX): tools/pmc_insts.sh <lib.so> <tag> [bench args]   -- instruction counters of the cell kernel for one build
# then HERE: python tools/summarize_insts.py <tag>  -> profiles/inst_counts.json (read by bench.py's roofline_issue block)
# (measurement passes: the summaries take per-launch means over every dispatch of the cell kernel, so the self-check's three short trial
#  launches -- on by default since round 6 for builds without the mark -- are switched off here; results are checked everywhere else)
export RS_JIT_SELFCHECK=0
R=$GRAFT_REPO_ROOT; LIB=$1; TAG=$2; shift 2
cd /tmp && export TMPDIR=/tmp
RS_HIP_LIB=$R/radiosaber_amd/$LIB rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY --output-format csv -d $R/gpurun_out/pmc_$TAG -- python3 $R/bench.py --no-cpu-baseline --no-r64 --no-streamed --no-cells1024 --steps 2 --warmup 1 "$@" > $R/gpurun_out/pmc_$TAG.log 2>&1
cd $R
python3 tools/summarize_insts.py $TAG --print-only
