"""Synthetic reference + phased SNP VCF for the k-mer extraction tests (shared by CPU and GPU tests)."""
import gzip
import os

import numpy as np


def make_graph_files(tmpdir, chrom="7", length=3000, n_sites=260, n_samples=65, seed=5, gz=True):
    """FASTA + VCF with: clustered and isolated SNPs, multi-allelic sites, a few indel / MNP records
    (to be skipped), unphased and missing genotypes.  Returns (fasta, vcf)."""
    rng = np.random.default_rng(seed)
    ref = rng.choice(list("ACGT"), size=length, p=[0.3, 0.2, 0.2, 0.3])
    ref[rng.integers(0, length, 5)] = "N"
    fasta = os.path.join(tmpdir, "ref.fa")
    with open(fasta, "w") as fh:
        fh.write(">other\nACGTACGT\n")
        fh.write(f">{chrom} synthetic\n")
        s = "".join(ref)
        for i in range(0, length, 60):
            fh.write(s[i:i + 60] + "\n")
    # positions: half uniformly spread, half in tight clusters (several sites inside one window)
    pos = set(rng.integers(0, length, n_sites // 2).tolist())
    for c in rng.integers(0, length - 40, n_sites // 12):
        pos.update((c + rng.integers(0, 12, 6)).tolist())
    pos = sorted(p for p in pos if ref[p] != "N")
    vcf = os.path.join(tmpdir, "var.vcf.gz" if gz else "var.vcf")
    op = gzip.open if gz else open
    with op(vcf, "wt") as fh:
        fh.write("##fileformat=VCFv4.1\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" +
                 "\t".join(f"s{i}" for i in range(n_samples)) + "\n")
        fh.write("other\t3\t.\tG\tA\t99\t.\t.\tGT\t" + "\t".join(["0|1"] * n_samples) + "\n")
        for p in pos:
            r = ref[p]
            others = [b for b in "ACGT" if b != r]
            rng.shuffle(others)
            kind = rng.random()
            if kind < 0.06:                      # deletion record: skipped by the graph
                alt, refa = r, r + "".join(ref[p + 1:p + 3])
            elif kind < 0.10:                    # insertion
                alt, refa = r + "GA", r
            else:
                n_alt = 1 if kind < 0.8 else (2 if kind < 0.95 else 3)
                alt, refa = ",".join(others[:n_alt]), r
            n_all = 1 + len(alt.split(","))
            af = rng.random() ** 2
            gts = []
            for _ in range(n_samples):
                a = [int(rng.integers(1, n_all)) if rng.random() < af else 0 for _ in range(2)]
                u = rng.random()
                if u < 0.02:
                    gts.append(f"{a[0]}/{a[1]}")     # unphased: taken in file order
                elif u < 0.03:
                    gts.append(".|.")                # missing -> reference allele
                else:
                    gts.append(f"{a[0]}|{a[1]}")
            fh.write(f"{chrom}\t{p + 1}\t.\t{refa}\t{alt}\t99\t.\t.\tGT\t" + "\t".join(gts) + "\n")
    return fasta, vcf
