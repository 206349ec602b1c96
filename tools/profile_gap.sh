#!/bin/bash
# Run ON THE GPU BOX (through gpurun): why does rocprofv3 see a shorter cell kernel than bench.py's HIP events (117.9 ms
# against 121.2 ms)?  One lease, back to back: plain bench -> bench under `rocprofv3 --kernel-trace --stats` -> plain bench, the shader clock sampled
# (rocm-smi) beside each.  Output: gpurun_out/gap/*.log; HERE: python tools/summarize_gap.py -> profiles/r04_rocprof_vs_events.md
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/gap; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
ARGS="--no-cpu-baseline --no-r64 --no-streamed --no-cells1024 --steps 12 --warmup 2"
sample() { # tag: shader / memory clock and power twice a second while the run lasts
  ( while true; do echo "$(date +%s.%N) $(rocm-smi --showclocks --showpower 2>/dev/null | grep -E 'sclk|mclk|Power' | tr -s ' ' | tr '\n' ';')"; sleep 0.5; done ) > $O/clocks_$1.log 2>&1 &
  SAMPLER=$!
}
sample plain1; python3 $R/bench.py $ARGS > $O/plain1.log 2>&1; kill $SAMPLER
sample rocprof; rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py $ARGS > $O/rocprof.log 2>&1; kill $SAMPLER
sample plain2; python3 $R/bench.py $ARGS > $O/plain2.log 2>&1; kill $SAMPLER
for f in plain1 rocprof plain2; do grep '^{' $O/$f.log | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); ms=d['kernel_ms_per_launch']; print('%-8s value %.3f M  HIP-event ms per launch: mean %.3f min %.3f max %.3f' % (sys.argv[1], d['value']/1e6, sum(ms)/len(ms), min(ms), max(ms)))" $f; done
grep rs_cell_kernel $O/kt/*/*_kernel_stats.csv | head -3
