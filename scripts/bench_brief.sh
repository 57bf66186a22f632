#!/usr/bin/env bash
# Development aid: bench.py with the given args, one short line (step, kernel, gap, frac).
cd "$(dirname "${BASH_SOURCE[0]}")/.."
python bench.py --no-cpu-baseline --no-e2e "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']
print('%-40s step %8.2f us  kernel %8.2f us  gap %6.2f  frac %.3f  value %.4g  bursts %s' % ('$*', j['ms_per_step']*1e3, r['kernel_ms_avg']*1e3, (j['ms_per_step']-r['kernel_ms_avg'])*1e3, r['frac'], j['value'], j['config'].get('burst_ms')))"
