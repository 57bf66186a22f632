#!/usr/bin/env bash
# Development aid (GPU box): kernel trace of scripts/extract_bench.py (emit kernels one after the other) for every
# lab/libgx_*.so -- variants of graph_extract.hip built on the side.  Output: gpurun_out/extract_lab.txt
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
out="$root/gpurun_out/extract_lab.txt"; : > "$out"
cd /tmp && export TMPDIR=/tmp
export GRAFIMO_EXTRACT_SERIAL=${GRAFIMO_EXTRACT_SERIAL:-1}
for lib in "$root"/lab/libgx_*.so; do
  export GRAFIMO_HIP_LIB="$lib"
  rm -rf /tmp/xlab
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/xlab -- python3 "$root/scripts/extract_bench.py" > /tmp/xlab.json 2> /tmp/xlab.log
  f=$(ls -t /tmp/xlab/*/*kernel_stats.csv 2>/dev/null | head -1)
  echo "== $(basename $lib)" >> "$out"
  python3 - "$f" >> "$out" <<'PY'
import csv, sys
for r in csv.reader(open(sys.argv[1])):
    if r[0].startswith('(anonymous namespace)::graph_emit') or 'scatter' in r[0]:
        print('%-40s calls %4s avg %9.1f us' % (r[0].split('::')[1].split('(')[0], r[1], float(r[3]) / 1e3))
PY
done
cat "$out"
