#!/usr/bin/env bash
# usage: scripts/pmc.sh <tag> [args to prof_score.py]   (run on the GPU box through gpurun)
# separate --pmc passes (no hip/hsa tracing with counters, per the pool's rules)
set -u
tag="$1"; shift
out="$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag"
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
run() { name="$1"; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$out/$name" -- python3 "$GRAFT_REPO_ROOT/scripts/prof_score.py" $ARGS > "$out/$name.log" 2>&1; }
ARGS="$*"
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_WAVES
run fetch FETCH_SIZE
run write WRITE_SIZE
run grbm GRBM_GUI_ACTIVE
python3 - "$out" <<'PY'
import sys, glob, csv, collections, os
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sorted(glob.glob(out + "/*/")):
    files = sorted(glob.glob(d + "*/*counter_collection.csv"), key=os.path.getmtime)
    if not files:
        continue
    for r in csv.DictReader(open(files[-1])):      # newest process of this pass only
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][-40:]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/summary.txt", "w") as fh:
    for k, d in agg.items():
        for c, v in sorted(d.items()):
            line = f"{k:42s} {c:24s} n={len(v):3d} mean={sum(v)/len(v):.6g} min={min(v):.6g} max={max(v):.6g}"
            print(line); fh.write(line + "\n")
PY
