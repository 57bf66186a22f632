"""Development aid (GPU box): the extraction kernels against the per-haplotype brute force (oracle/extract_bruteforce.py)
and the walk enumerator (oracle/extract_oracle.py) on random conflict-free graphs of every allele kind, seed after seed
for a fixed time.  TEST INFRASTRUCTURE (imports oracle/): not part of the product.
    python scripts/extract_fuzz.py [seconds] [first_seed]"""
import os
import sys
import tempfile
import time

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "tests"))
from extract_helpers import make_consistent_graph_files
from grafimo_amd.extract_regions import DeviceGraph, GraphIndex
from oracle import extract_bruteforce as bf
from oracle import extract_oracle as xo

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
kinds_list = ["s", "sd", "si", "sm", "sc", "sidm", "sD", "sO", "sidmDO", "sidmDOcS", "dc", "ic", "cS"]
t0 = time.time()
n_graphs = n_rows = n_carried = n_heavy = 0
devnull = open(os.devnull, "w")


def hip_rows(rows):
    km = rows.kmers.cpu().numpy()
    st, sp = rows.start.cpu().numpy(), rows.stop.cpu().numpy()
    sd, fr, rf = rows.strand.cpu().numpy(), rows.freq.cpu().numpy(), rows.is_ref.cpu().numpy()
    return [(km[i].tobytes(), int(st[i]), int(sp[i]), chr(sd[i]), int(fr[i]), "ref" if rf[i] else "non.ref")
            for i in range(len(rows))]


with tempfile.TemporaryDirectory() as tmp:
    while time.time() - t0 < budget:
        kinds = kinds_list[seed % len(kinds_list)]
        n_samples = [3, 16, 40, 70][seed % 4]            # 6 .. 140 haplotypes: one to three bitset words
        fasta, vcf = make_consistent_graph_files(tmp, chrom="c", length=360, n_samples=n_samples, seed=seed, kinds=kinds,
                                                 dense=seed % 3 != 0)
        ref = xo.read_fasta(fasta)["c"]
        recs, H = bf.read_vcf_records(vcf, "c")
        assert bf.consistent(ref, recs, H), (seed, kinds)
        v = xo.read_vcf_variants(vcf, "c")
        stderr, sys.stderr = sys.stderr, devnull
        try:
            idx = GraphIndex.from_fasta_vcf(fasta, vcf, "c")
        finally:
            sys.stderr = stderr
        g = DeviceGraph(idx)
        for (S, E), W in [((0, 120), 19), ((100, 260), [5, 8, 11, 14][seed % 4]), ((200, 360), [24, 30, 33][seed % 3]),
                          ((330, 360), 12)]:
            import numpy as np
            best = 0.0                    # log2 of the allele product of the densest window
            for p0 in range(S, E):
                inside = (idx.pos >= p0) & (idx.pos < p0 + W)
                best = max(best, float(np.log2(1.0 + idx.n_alts[inside]).sum()))
            got = hip_rows(g.extract([(S, E)], W))       # (round 3 refused plans with a window of more than 2^20 walks)
            if best >= 17.0:              # too many walks for the Python enumerator: the brute force still checks every row
                n_heavy += 1
                freq, flags = bf.window_counts(ref, recs, H, S, E, W)
                carried, n = bf.check_rows(got, freq, flags)
                print(f"heavy: seed {seed} kinds {kinds} region {S}-{E} W {W}: densest window 2^{best:.1f} allele combinations, "
                      f"{n} rows checked against the per-haplotype brute force", flush=True)
                n_rows += n
                n_carried += carried
                continue
            freq, flags = bf.window_counts(ref, recs, H, S, E, W)
            carried, n = bf.check_rows(got, freq, flags)
            exp = xo.enumerate_region_variants("c", ref, v, S, E, W, with_counts=True)
            want = [(r[1].encode(), int(r[2].split(":")[1][:-1]), int(r[3].split(":")[1][:-1]), r[2][-1], r[4], r[5]) for r in exp]
            assert got == want, (seed, kinds, S, E, W, len(got), len(want),
                                 next((i, a, b) for i, (a, b) in enumerate(zip(got, want)) if a != b) if len(got) == len(want) else None)
            n_rows += n
            n_carried += carried
        g.close()
        n_graphs += 1
        seed += 1
print(f"extract_fuzz: {n_graphs} graphs, {n_rows} rows ({n_carried} keys carried by a haplotype) in {time.time() - t0:.0f} s: "
      f"kernels == enumerator == brute force ({n_heavy} plans with a window of more than 2^17 walks: brute force only); next seed {seed}")
