#!/usr/bin/env bash
# Development aid (GPU box): bench.py step and kernel time for every lab/libgfm_*.so, two rounds (A/B on one box).
#   scripts/lab_bench.sh [bench args]
cd "$(dirname "${BASH_SOURCE[0]}")/.."
for round in 1 2; do
  for lib in lab/libgfm_*.so; do
    GRAFIMO_HIP_LIB="$PWD/$lib" python bench.py --no-cpu-baseline --no-e2e "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']
print('%-26s step %.2f us  kernel %.2f us  gap %.2f' % ('$lib'.split('/')[-1], j['ms_per_step']*1e3, r['kernel_ms_avg']*1e3, (j['ms_per_step']-r['kernel_ms_avg'])*1e3))"
  done
done
