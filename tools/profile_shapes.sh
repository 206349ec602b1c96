#!/bin/bash
# Run ON THE GPU BOX (through gpurun): PMC passes for every workload of DESIGN.md section 6's table, so that each number there
# has a keyed, source-hash-matching entry in profiles/traffic.json and profiles/inst_counts.json.
# Per shape three rocprofv3 runs of bench.py, each counter set in its own pass (no tracing beside --pmc):
#   FETCH_SIZE | WRITE_SIZE | SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY
# then HERE: python tools/summarize_shapes.py
# (measurement passes: the summaries take per-launch means over every dispatch of the cell kernel, so the self-check's three short trial
#  launches -- on by default since round 6 for builds without the mark -- are switched off here; results are checked everywhere else)
export RS_JIT_SELFCHECK=0
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
shape() { # tag, bench args...
  local tag=$1; shift
  for pass in fetch write insts; do
    case $pass in
      fetch) pmc="FETCH_SIZE";;
      write) pmc="WRITE_SIZE";;
      insts) pmc="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY";;
    esac
    rm -rf $R/gpurun_out/ps_${tag}_$pass
    rocprofv3 --pmc $pmc --output-format csv -d $R/gpurun_out/ps_${tag}_$pass -- python3 $R/bench.py --no-cpu-baseline --no-r64 --no-streamed --no-cells1024 --steps 3 --warmup 1 --ttis 2000 "$@" > $R/gpurun_out/ps_${tag}_$pass.log 2>&1
  done
  grep '^{' $R/gpurun_out/ps_${tag}_insts.log | tail -1 | cut -c1-140
}
shape s9_r25
shape s9_r25_refresh1 --cqi-refresh 1   # streamed-CQI mode: a grid from HBM every TTI (roofline.streamed)
shape s9_r64 --rbgs 64 --rbg-size 8
shape s9_u1000 --ues-per-slice 50
shape s8_r25 --sched 8
shape s8_r64 --sched 8 --rbgs 64 --rbg-size 8
shape s8_u1000 --sched 8 --ues-per-slice 50
shape s1_r25 --sched 1
shape s1_u1000 --sched 1 --ues-per-slice 50
shape s7_r25 --sched 7
shape s7_u1000 --sched 7 --ues-per-slice 50
# two of the reference's own experiment shapes (64 RBGs, ragged slices; profiles/r05_shipped_shapes.md)
shape cfg_fixranues20_s9 --config-key exp-fixranues/20slices-ip/config-pf.json
shape cfg_fix15_s8 --config-key exp-fix20slices/15ues-ip/config-pf.json --sched 8
