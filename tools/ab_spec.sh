one() { RS_JIT_EXTRA="$1" timeout 120 python bench.py --no-cpu-baseline --no-r64 --steps 8 $2 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-40s %-28s %.2f M  %.2f us' % (sys.argv[1], sys.argv[2], d['value']/1e6, d['us_per_tti_per_cell']))" "$1" "$2"; }
for v in "" "-DRS_NO_SPEC" "-DRS_SPEC_ALL" "-DRS_SPEC_NAP=8" "-DRS_SPEC_PRIO=1" "-DRS_SPEC_ALL -DRS_SPEC_PRIO=1"; do one "$v" ""; done
for v in "" "-DRS_NO_SPEC" "-DRS_SPEC_ALL" ; do one "$v" "--sched 8"; one "$v" "--rbgs 64 --rbg-size 8 --steps 3"; done
