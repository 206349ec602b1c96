cd $GRAFT_REPO_ROOT; O=gpurun_out/nvs_exp2; mkdir -p $O
python -m pytest tests -m gpu -x -q -k "sampler or nongreedy or nvs_non" > $O/pytest.log 2>&1; tail -5 $O/pytest.log
ab() { local tag=$1 extra=$2; shift 2
  RS_JIT_EXTRA="$extra" timeout 300 python bench.py --allow-variant --no-cpu-baseline --no-r64 --no-streamed --steps 3 --warmup 1 "$@" > $O/ab_$tag.log 2>&1
  grep -h '^{' $O/ab_$tag.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%-20s %-22s %.2f M TTIs/s  %.3f us' % (sys.argv[1], sys.argv[2], d['value']/1e6, d['us_per_tti_per_cell']))" "$tag" "[$extra]" || tail -3 $O/ab_$tag.log
}
ab r25 "" --sched 11 --ttis 2000
ab r64 "" --sched 11 --ttis 1000 --rbgs 64 --rbg-size 8
ab u100r64 "" --sched 11 --ttis 2000 --ues-per-slice 5 --rbgs 64 --rbg-size 8
ab u200r64 "" --sched 11 --ttis 2000 --ues-per-slice 10 --rbgs 64 --rbg-size 8
