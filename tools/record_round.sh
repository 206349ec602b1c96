#!/bin/bash
# Run ON THE GPU BOX (through gpurun) once the kernels are final: every record under profiles/ that carries the device sources'
# hash, in one go (about ten minutes).  Then HERE:
#   python tools/summarize_rocprof.py <round>; python tools/summarize_shapes.py --phase-shares-headline '{...}';
#   python tools/summarize_queue_prof.py <tag> after; cp gpurun_out/<round>_bench_default.log profiles/
# usage: tools/record_round.sh <round tag, e.g. r03>
R=$GRAFT_REPO_ROOT; TAG=${1:-rxx}
$R/tools/profile_round.sh > $R/gpurun_out/prof_round.log 2>&1
$R/tools/profile_shapes.sh > $R/gpurun_out/prof_shapes.log 2>&1
$R/tools/profile_queue_mode.sh ${TAG}q > $R/gpurun_out/prof_queue.log 2>&1
cd $R
python3 bench.py > gpurun_out/${TAG}_bench_default.log 2> gpurun_out/${TAG}_bench_default.err
tools/sweep_round.sh 2>&1 | grep "^|" > gpurun_out/${TAG}_sweep.log
if [ -f radiosaber_amd/libradiosaber_hip_stamps.so ]; then
  export RS_HIP_LIB=radiosaber_amd/libradiosaber_hip_stamps.so
  (RS_JIT_EXTRA="-DRS_STAMPS" python3 tools/phase_stamps.py --jit --ttis 400
   RS_JIT_EXTRA="-DRS_STAMPS -DRS_STAMPS_HOLD" python3 tools/phase_stamps.py --jit --hold --ttis 400 | grep "hold:"
   RS_JIT_EXTRA="-DRS_STAMPS" python3 tools/phase_stamps.py --jit --ttis 400 --sched 8
   RS_JIT_EXTRA="-DRS_STAMPS" python3 tools/phase_stamps.py --jit --ttis 400 --rbgs 64 --rbg-size 8) 2>&1 | grep -v "^    -" > gpurun_out/${TAG}_stamps_final.log
fi
cut -c1-150 gpurun_out/${TAG}_bench_default.log
