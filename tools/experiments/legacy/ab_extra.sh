#!/bin/bash
# A/B of shape-specialised builds on one box: tools/ab_extra.sh "<bench args>" "<extra 1>" "<extra 2>" ...   (each extra = RS_JIT_EXTRA value)
args="$1"; shift
for rep in 1 2; do
for v in "$@"; do
  RS_JIT_EXTRA="$v" timeout 150 python bench.py --no-cpu-baseline --steps 6 $args 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-50s %.2f M  r64 %s' % (sys.argv[1], d['value']/1e6, ('%.2f M' % (d['value_r64']/1e6)) if 'value_r64' in d else '-'))" "[$v]"
done
done
