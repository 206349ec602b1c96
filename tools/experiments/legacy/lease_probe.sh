#!/bin/bash
# What kind of box is this lease?  Firmware / partition / clock facts next to a short headline bench (profiles/r05_rocprof_vs_events.md:
# the round's leases fall into two groups, 34.6 and 35.5 M TTIs/s on identical code at the same shader clock).
R=$GRAFT_REPO_ROOT; cd $R
echo "== $(date -u +%H:%M:%S) host $(hostname)"
for f in vbios_version current_compute_partition current_memory_partition pp_dpm_fclk pp_dpm_mclk pp_dpm_socclk power_dpm_force_performance_level; do
  for d in /sys/class/drm/card*/device; do [ -r $d/$f ] && echo "$f: $(cat $d/$f 2>/dev/null | tr '\n' ' ')"; done
done
rocm-smi --showfwinfo 2>/dev/null | grep -E "SMC|MEC|RLC|SDMA|VCN|PSP|TA |IMU" | head -12
rocm-smi --showcomputepartition --showmemorypartition --showperflevel 2>/dev/null | grep -v "^=" | grep -v "^$" | head -8
rocminfo 2>/dev/null | grep -E "Compute Unit|Max Clock|Marketing Name|Shader Engines|Shader Arrs|SIMDs per CU" | sed -n 1,14p
python bench.py --no-cpu-baseline --no-streamed --no-r64 --steps 4 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().split(chr(10))[-1])
print('HEADLINE %.2f M TTIs/s, %.1f MHz, cells %.2f / %.2f / %.2f ms (min / mean / max), CUs %d' % (d['value'] / 1e6, d['shader_mhz'], d['cell_ms_min'], d['cell_ms_mean'], d['cell_ms_max'], d['compute_units']))"
