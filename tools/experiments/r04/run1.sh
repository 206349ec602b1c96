#!/bin/bash
# Round 4, GPU run 1 (through gpurun): new parity tests, 64-RBG grid at 512 vs 640 threads, the hybrid sort at 1 280 records, streamed mode.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_run1; mkdir -p $O; cd ..
python -m pytest tests/test_gpu_round4.py -m gpu -x -q > $O/pytest_round4.log 2>&1; echo "pytest rc $?" >> $O/pytest_round4.log
B="python bench.py --no-cpu-baseline --no-r64 --no-streamed --steps 6 --warmup 1 --ttis 4000"
for rep in 1 2; do
  for nt in 512 640; do
    $B --rbgs 64 --rbg-size 8 --threads $nt > $O/r64_nt${nt}_$rep.log 2>&1
  done
done
$B --rbgs 64 --rbg-size 8 --threads 768 > $O/r64_nt768_1.log 2>&1
cd ..ols/microbench
for x in mb_sort_r64_nt512_k2 mb_sort_r64_nt640_k2 mb_sort_r64_nt512_k1; do timeout 300 ./$x keys_r64.bin > $O/$x.log 2>&1; done
cd ..
python bench.py --no-cpu-baseline --steps 5 --warmup 1 > $O/bench_default.log 2>&1
python bench.py --no-cpu-baseline --no-r64 --no-streamed --cqi-refresh 1 --steps 5 --warmup 1 --ttis 2000 > $O/bench_refresh1.log 2>&1
grep -h '^{' $O/r64_*.log | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['config']['rbgs'], d['kernel'], round(d['value']/1e6,3), 'M TTIs/s', d['us_per_tti_per_cell'])
"
tail -3 $O/pytest_round4.log
