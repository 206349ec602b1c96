one() { RS_JIT_EXTRA="$1" timeout 150 python bench.py --no-cpu-baseline --no-r64 --steps 4 $2 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-40s %-34s %.2f M  %.2f us' % (sys.argv[1], sys.argv[2], d['value']/1e6, d['us_per_tti_per_cell']))" "$1" "$2"; }
for a in "--rbgs 64 --rbg-size 8" "--sched 8 --rbgs 64 --rbg-size 8" "--sched 9"; do
  for v in "" "-DRS_NO_SPEC"; do one "$v" "$a"; done
done
