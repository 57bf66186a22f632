#!/usr/bin/env bash
# Development aid (GPU box): pipelined config-2 step against the score kernel's grid size (GRAFIMO_SCORE_GRID).
cd "$(dirname "${BASH_SOURCE[0]}")/.."
for round in 1 2; do
  for g in 0 224 233 240 245 248 252 256; do
    echo -n "grid $g: "
    GRAFIMO_SCORE_GRID=$g python scripts/tail_probe2.py 2>/dev/null | tail -1
  done
done
