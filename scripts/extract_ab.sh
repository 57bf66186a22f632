#!/usr/bin/env bash
# Development aid (GPU box): the extraction's emit stage -- bench.py's `extract` block three times, then a
# rocprofv3 kernel trace with the two emit kernels run one after the other (GRAFIMO_EXTRACT_SERIAL=1: what
# each takes alone), one as shipped (side by side), and serial without haplotype counts / without deletions.  Output: gpurun_out/extract_ab/.
cd "$(dirname "${BASH_SOURCE[0]}")/.."
out=gpurun_out/extract_ab; mkdir -p $out
for i in 1 2 3; do python scripts/extract_bench.py 2>/dev/null | tail -1; done | tee $out/blocks.txt
root="$PWD"
out="$root/$out"
cd /tmp && export TMPDIR=/tmp
for mode in 1 0 nocounts nodels; do
  unset EXTRACT_NO_COUNTS EXTRACT_NO_DELS
  export GRAFIMO_EXTRACT_SERIAL=1
  case $mode in 0) export GRAFIMO_EXTRACT_SERIAL=0;; nocounts) export EXTRACT_NO_COUNTS=1;; nodels) export EXTRACT_NO_DELS=1;; esac
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/xprof$mode -- python3 "$root/scripts/extract_bench.py" > /dev/null 2> $out/prof$mode.log
  f=$(ls -t /tmp/xprof$mode/*/*kernel_stats.csv 2>/dev/null | head -1)
  echo "== serial=$mode" | tee -a $out/kernels.txt
  python3 - "$f" <<'PY' | tee -a $out/kernels.txt
import csv, sys
for r in csv.reader(open(sys.argv[1])):
    if r[0].startswith('(anonymous namespace)::graph_'):
        print('%-40s calls %4s avg %9.1f us' % (r[0].split('::')[1].split('(')[0], r[1], float(r[3]) / 1e3))
PY
done
