#!/usr/bin/env bash
# usage: scripts/pmc_extract.sh   (GPU box, through gpurun): HBM traffic of the extraction kernels (separate --pmc passes of
# scripts/extract_bench.py: FETCH_SIZE, WRITE_SIZE in KiB per launch; no tracing domains beside --kernel-trace)
set -u
root="$GRAFT_REPO_ROOT"
out="$root/gpurun_out/pmc_extract"
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$out/$c" -- python3 "$root/scripts/extract_bench.py" > "$out/$c.log" 2>&1
done
python3 - "$out" <<'PY'
import sys, glob, csv, collections, os
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sorted(glob.glob(out + "/*/")):
    files = sorted(glob.glob(d + "*/*counter_collection.csv"), key=os.path.getmtime)
    for f in files[-1:]:
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][-40:]
            if k.startswith("graph_"):
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/summary.txt", "w") as fh:
    for k, d in sorted(agg.items()):
        for c, v in sorted(d.items()):
            line = f"{k:28s} {c:12s} launches={len(v):3d} mean={sum(v)/len(v):12.1f} KiB"
            print(line); fh.write(line + "\n")
PY
