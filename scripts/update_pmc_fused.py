"""profiles/pmc_fused.json from a scripts/pmc_fused_sq.sh summary (SQ instruction counters of graph_score_kernel on the
bench's graph: 10 000 regions x 200 bp, W = 19), keyed to a hash of the fused kernels' sources as they are NOW -- run it on the
tree the counters were taken from.  bench.py's `extract.roofline` prices the kernel against its instruction-issue floor
with these counts.
    python scripts/update_pmc_fused.py profiles/r05_pmc_fused_sq.txt "scripts/pmc_fused_sq.sh, three --pmc passes" """
import hashlib, json, os, re, sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
summary, note = sys.argv[1], sys.argv[2]
vals, cur = {}, None
for line in open(summary):
    if not line.startswith((" ", "#")):
        cur = line.strip()
    m = re.match(r"\s+(\w+)\s+n=\s*\d+\s+mean=\s*([0-9.e+]+)", line)
    if m and cur == "graph_score_kernel":
        vals[m.group(1)] = float(m.group(2))
files = ["grafimo_amd/csrc/gfm_graph_fused.hpp", "grafimo_amd/csrc/graph_extract.hip"]
src = b"".join(open(os.path.join(root, f), "rb").read() for f in files)
out = {
    "kernel": "graph_score_kernel<1, false, true>", "workload": "bench.py extract block: 10 000 regions x 200 bp, 69 224 sites, W = 19, 6.04e6 rows",
    "insts_valu": vals["SQ_INSTS_VALU"], "insts_salu": vals["SQ_INSTS_SALU"], "insts_lds": vals["SQ_INSTS_LDS"],
    "insts_branch": vals.get("SQ_INSTS_BRANCH"), "waves": vals.get("SQ_WAVES"), "wave_cycles": vals.get("SQ_WAVE_CYCLES"),
    "wait_any_cycles": vals.get("SQ_WAIT_ANY"), "active_inst_any": vals.get("SQ_ACTIVE_INST_ANY"),
    "lds_idx_active_cycles": vals.get("SQ_LDS_IDX_ACTIVE"), "lds_bank_conflict_cycles": vals.get("SQ_LDS_BANK_CONFLICT"),
    "source": f"{os.path.relpath(summary, root)} ({note})",
    "kernel_source_sha16": hashlib.sha256(src).hexdigest()[:16], "kernel_source_files": files,
}
with open(os.path.join(root, "profiles", "pmc_fused.json"), "w") as fh:
    json.dump(out, fh, indent=1)
    fh.write("\n")
print(json.dumps(out, indent=1))
