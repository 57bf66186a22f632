#!/usr/bin/env bash
# Development aid (GPU box): pipelined config-2 step and kernel time (scripts/tail_probe2.py) for every
# lab/libgfm_*.so, two rounds (same-box A/B).  LAB_MODES="0 1 2": also vary GRAFIMO_LAB_POST (labsw builds).
cd "$(dirname "${BASH_SOURCE[0]}")/.."
for round in 1 2 3; do
  for lib in lab/libgfm_*.so; do
    modes=0
    case "$lib" in *labsw*) modes="${LAB_MODES:-0 1 2 3 4}";; esac
    for m in $modes; do
      echo "== $lib mode $m"
      GRAFIMO_LAB_POST=$m GRAFIMO_HIP_LIB="$PWD/$lib" python scripts/tail_probe2.py 2>/dev/null | tail -1
    done
  done
done
