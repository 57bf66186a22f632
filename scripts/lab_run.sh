#!/usr/bin/env bash
# Development aid (GPU box): times every lab/libgfm_*.so with scripts/prof_score.py.
#   scripts/lab_run.sh [reps] [modes...]
cd "$(dirname "${BASH_SOURCE[0]}")/.."
reps="${1:-40}"; shift || true
modes="${*:-select nohist}"
for lib in lab/libgfm_*.so; do
  for m in $modes; do
    GRAFIMO_HIP_LIB="$PWD/$lib" python scripts/prof_score.py "$reps" 10000 "$m" 2>&1 | tail -1
  done
done
