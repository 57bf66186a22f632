#!/usr/bin/env bash
# Development aid (GPU box): the resident pipeline's step under library variants, alternating, three rounds.
#   scripts/lab_step.sh <tag> <lib> [<lib> ...]        (lib: product | lab/libgfm_X.so)
root="$GRAFT_REPO_ROOT"; tag="$1"; shift; out="$root/gpurun_out/$tag.txt"; cd "$root"
for round in 1 2 3; do
  for lib in "$@"; do
    unset GRAFIMO_HIP_LIB; [ "$lib" != "product" ] && export GRAFIMO_HIP_LIB="$root/$lib"
    python3 bench.py --no-cpu-baseline --no-e2e --no-extras 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; t=d.get('tail_ms') or {}
print('%-26s value %.4g  step %.2f us  kernel %.2f us  gap %.2f us  tail avg %.1f max %.1f us' % ('$lib', d['value'], 1e3*d['ms_per_step'], 1e3*r['kernel_ms_avg'], 1e3*(d['ms_per_step']-r['kernel_ms_avg']), 1e3*t.get('avg',0), 1e3*t.get('max',0)))"
  done
done | tee "$out"
