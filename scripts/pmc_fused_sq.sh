#!/usr/bin/env bash
# Development aid (GPU box): SQ counters of the fused extraction -> scoring kernels (two --pmc passes of scripts/fused_prof.py).
#   scripts/pmc_fused_sq.sh   (gpurun_out/pmc_fused_sq/summary.txt)
root="$GRAFT_REPO_ROOT"; out="$root/gpurun_out/pmc_fused_sq"; rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d "$out/a" -- python3 "$root/scripts/fused_prof.py" > "$out/a.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INSTS_BRANCH SQ_WAVES --output-format csv -d "$out/b" -- python3 "$root/scripts/fused_prof.py" > "$out/b.log" 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_WAIT_ANY SQ_INST_CYCLES_VMEM --output-format csv -d "$out/c" -- python3 "$root/scripts/fused_prof.py" > "$out/c.log" 2>&1
python3 - "$out" > "$out/summary.txt" <<'PY'
import sys, glob, csv, collections, os, re
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sorted(glob.glob(out + "/*/")):
    files = sorted(glob.glob(d + "*/*counter_collection.csv"), key=os.path.getmtime)
    for f in files[-1:]:
        for r in csv.DictReader(open(f)):
            m = re.search(r"graph_(score|del_score|del_count|annotate|hist_reduce)_kernel", r["Kernel_Name"])
            if m: agg[m.group(0)][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(agg.items()):
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:24s} n={len(v):3d} mean={sum(v)/len(v):14.0f}")
PY
rm -rf "$out"/a "$out"/b "$out"/c
cat "$out/summary.txt"
