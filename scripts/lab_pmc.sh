#!/usr/bin/env bash
# Development aid (GPU box): PMC counters of the score kernel for one library build.
#   scripts/lab_pmc.sh <lib.so|default> <tag> [prof_score.py args]
# Separate --pmc passes with --kernel-trace only (the pool's rule); the program after -- is python3 itself.
set -u
lib="$1"; tag="$2"; shift 2
root="$GRAFT_REPO_ROOT"
[ "$lib" != default ] && export GRAFIMO_HIP_LIB="$root/$lib"
out="$root/gpurun_out/pmc_$tag"
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
ARGS="${*:-6 10000 select}"
run() { name="$1"; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$out/$name" -- python3 "$root/scripts/prof_score.py" $ARGS > "$out/$name.log" 2>&1; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU
run sq2 SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_VMEM SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU
run sq3 SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_VALU
run sq4 SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_SMEM SQ_WAVES SQ_CYCLES
run ta TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
run tcc TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_BUSY_sum
run grbm GRBM_GUI_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
python3 - "$out" <<'PY'
import sys, glob, csv, collections, os
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sorted(glob.glob(out + "/*/")):
    files = sorted(glob.glob(d + "*/*counter_collection.csv"), key=os.path.getmtime)
    if not files:
        continue
    for r in csv.DictReader(open(files[-1])):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][-40:]
        if "score" not in k:
            continue
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/summary.txt", "w") as fh:
    for k, d in agg.items():
        for c, v in sorted(d.items()):
            line = f"{k:36s} {c:36s} n={len(v):3d} mean={sum(v)/len(v):.6g}"
            print(line); fh.write(line + "\n")
PY
