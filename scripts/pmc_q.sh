#!/usr/bin/env bash
set -u
out="$GRAFT_REPO_ROOT/gpurun_out/pmc_q"; mkdir -p "$out"; cd /tmp; export TMPDIR=/tmp PROF_Q=1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d "$out/sq1" -- python3 "$GRAFT_REPO_ROOT/scripts/prof_score.py" 3 1000 hist > "$out/sq1.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM --output-format csv -d "$out/sq2" -- python3 "$GRAFT_REPO_ROOT/scripts/prof_score.py" 3 1000 hist > "$out/sq2.log" 2>&1
python3 - "$out" <<'PY'
import sys, glob, csv, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "qvalue" not in k: continue
        agg["qvalue"][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for r in csv.DictReader(open(f.replace("counter_collection","kernel_trace"))):
        if "qvalue" in r["Kernel_Name"]:
            agg["qvalue"]["duration_ns"].append(float(r["End_Timestamp"])-float(r["Start_Timestamp"]))
for k, d in agg.items():
    for c, v in sorted(d.items()):
        print(f"{k} {c:24s} n={len(v)} mean={sum(v)/len(v):.6g}")
PY
