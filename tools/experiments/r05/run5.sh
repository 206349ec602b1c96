#!/bin/bash
# Round 5, GPU run 5: does the cell kernel miss its instruction cache?  (A box in its slow state runs MaximizeCell 2.5 % slower at the
# same shader clock; the lean kernel's code object is ~73 KB, a CU pair's instruction cache 64 KB.)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_run5; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -i -o -E "\b(SQC?_[A-Z_]*(ICACHE|IFETCH|INST_LEVEL|INSTS_SMEM|WAIT_INST)[A-Z_]*)\b" | sort -u | tr '\n' ' ' > $O/counters.txt; cat $O/counters.txt; echo
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --output-format csv -d $O/pmc_$tag -- python3 $R/tools/gap_probe.py --launches 4 --tag $tag > $O/$tag.json 2> $O/$tag.err
  f=$(find $O/pmc_$tag -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "rs_cell_kernel" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print("%-22s per launch %.4g  (per cell-TTI %.1f)" % (k, sum(v) / len(v), sum(v) / len(v) / (512 * 8000)))
PY
done 2>&1 | tee $O/summary.log
