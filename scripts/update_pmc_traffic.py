"""profiles/pmc_traffic.json from a scripts/lab_pmc.sh summary (FETCH_SIZE / WRITE_SIZE of score_quad_kernel<19, 1> at 2e7
rows), keyed to a hash of the kernel's sources as they are NOW -- run it on the tree the counters were taken from.
    python scripts/update_pmc_traffic.py profiles/r04_pmc_score_kernel.txt "scripts/lab_pmc.sh default r04b inside scripts/final_pass.sh" """
import hashlib, json, os, re, sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
summary, note = sys.argv[1], sys.argv[2]
vals = {}
for line in open(summary):
    m = re.match(r"score_quad_kernel<19, 1(?:, true)?>\s+(FETCH_SIZE|WRITE_SIZE)\s+n=\s*\d+\s+mean=([0-9.e+]+)", line)
    if m:
        vals[m.group(1)] = float(m.group(2))
files = ["grafimo_amd/csrc/gfm_score_quad.hpp", "grafimo_amd/csrc/score_quad_tu.hip"]
src = b"".join(open(os.path.join(root, f), "rb").read() for f in files)
fetch2 = int(round(vals["FETCH_SIZE"] * 1024 * 2))
out = {
    "kernel": "score_quad_kernel<19, 1>", "rows_per_launch": 20000000, "width": 19,
    "fetch_size_kib": vals["FETCH_SIZE"], "fetch_bytes_x2_gfx950_rule": fetch2, "write_size_kib": vals["WRITE_SIZE"],
    "hbm_bytes_per_launch": fetch2 + vals["WRITE_SIZE"] * 1024,
    "source": f"{os.path.relpath(summary, root)} ({note}: separate --pmc passes, FETCH_SIZE doubled per MI355X_MICROARCH.md's "
              f"gfx950 rule for 16 B/lane streams)",
    "kernel_source_sha16": hashlib.sha256(src).hexdigest()[:16], "kernel_source_files": files,
}
with open(os.path.join(root, "profiles", "pmc_traffic.json"), "w") as fh:
    json.dump(out, fh, indent=1)
    fh.write("\n")
print(json.dumps(out, indent=1))
