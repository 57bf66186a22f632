#!/usr/bin/env bash
# usage: scripts/pmc_bench_kernels.sh [bench args]   (GPU box): FETCH_SIZE / WRITE_SIZE (KiB per launch) of EVERY kernel of a
# short bench run, separate --pmc passes -- a look for traffic nobody asked for (scratch memory, zero fills).
set -u
root="$GRAFT_REPO_ROOT"
out="$root/gpurun_out/pmc_bench"
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$out/$c" -- python3 "$root/bench.py" --steps 20 --no-cpu-baseline --no-e2e --no-extras "$@" > "$out/$c.log" 2>&1
done
python3 - "$out" <<'PY'
import sys, glob, csv, collections, os, re
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sorted(glob.glob(out + "/*/")):
    files = sorted(glob.glob(d + "*/*counter_collection.csv"), key=os.path.getmtime)
    for f in files[-1:]:
        for r in csv.DictReader(open(f)):
            m = re.search(r"(\w+_kernel\w*|\w+Kernel\w*)", r["Kernel_Name"])
            k = (m.group(1) if m else r["Kernel_Name"])[:44]
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/summary.txt", "w") as fh:
    for k, d in sorted(agg.items()):
        line = f"{k:46s} " + "  ".join(f"{c} n={len(v):4d} mean={sum(v)/len(v):11.1f} KiB" for c, v in sorted(d.items()))
        print(line); fh.write(line + "\n")
PY
