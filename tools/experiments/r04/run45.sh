#!/bin/bash
# Round 4, GPU run 45: NVS scanning 50-user slices whole (four lanes per item in the serial phase) instead of in split runs
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_run45; mkdir -p $O; cd ..
ab() { local tag=$1 lib=$2 extra=$3; shift 3
  RS_HIP_LIB=$lib RS_JIT_EXTRA="$extra" timeout 300 python bench.py --allow-variant --no-cpu-baseline --no-r64 --no-streamed --steps 5 --warmup 1 "$@" > $O/ab_$tag.log 2>&1
  grep -h '^{' $O/ab_$tag.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%-20s %-28s %.2f M TTIs/s  %.3f us' % (sys.argv[1], sys.argv[2], d['value']/1e6, d['us_per_tti_per_cell']))" "$tag" "[$extra]" || tail -3 $O/ab_$tag.log
}
for rep in 1 2; do
ab s7_u1000_$rep radiosaber_amd/libradiosaber_hip.so "" --sched 7 --ttis 4000 --ues-per-slice 50
ab s7_u1000_$rep radiosaber_amd/libradiosaber_hip_nvs64.so "-DRS_NVS_WHOLE_SLICE=64" --sched 7 --ttis 4000 --ues-per-slice 50
ab s7_u1000_r64_$rep radiosaber_amd/libradiosaber_hip.so "" --sched 7 --ttis 2000 --ues-per-slice 50 --rbgs 64 --rbg-size 8
ab s7_u1000_r64_$rep radiosaber_amd/libradiosaber_hip_nvs64.so "-DRS_NVS_WHOLE_SLICE=64" --sched 7 --ttis 2000 --ues-per-slice 50 --rbgs 64 --rbg-size 8
done
ab s7_u1000_nojit radiosaber_amd/libradiosaber_hip.so "" --sched 7 --ttis 2000 --ues-per-slice 50 --no-jit
ab s7_u1000_nojit radiosaber_amd/libradiosaber_hip_nvs64.so "-DRS_NVS_WHOLE_SLICE=64" --sched 7 --ttis 2000 --ues-per-slice 50 --no-jit
RS_HIP_LIB=radiosaber_amd/libradiosaber_hip_nvs64.so RS_JIT_EXTRA="-DRS_NVS_WHOLE_SLICE=64" python -m pytest tests -m gpu -q -x -k "sched or nvs or 1000 or prepare or random or config or lean" 2>&1 | tail -3
