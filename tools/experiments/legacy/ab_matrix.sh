# usage: tools/ab_matrix.sh "<cells list>" "<bench args>" "<extra 1>" ...
cells="$1"; args="$2"; shift 2
for c in $cells; do
  for v in "$@"; do
    RS_JIT_EXTRA="$v" timeout 150 python bench.py --no-cpu-baseline --no-r64 --steps 6 --cells $c $args 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('cells %4d %-50s %.2f M  %.2f us/TTI/cell' % (int(sys.argv[2]), sys.argv[1], d['value']/1e6, d['us_per_tti_per_cell']))" "[$v]" $c
  done
done
