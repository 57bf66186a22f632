"""Throughput of the native VCF reader (gfm_vcf_*) on a 1000-Genomes-shaped file: 2504 samples
(5008 haplotypes), 20 000 records, plain text in /dev/shm."""
import os, sys, time, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grafimo_amd.extract_regions import GraphIndex
rng = np.random.default_rng(1)
d = tempfile.mkdtemp(dir="/dev/shm")
n_rec, n_s, L = 20_000, 2504, 2_000_000
ref = rng.choice(list("ACGT"), size=L)
open(os.path.join(d, "r.fa"), "w").write(">22\n" + "".join(ref) + "\n")
pos = np.sort(rng.choice(L - 10, n_rec, replace=False))
with open(os.path.join(d, "v.vcf"), "w") as fh:
    fh.write("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join(f"s{i}" for i in range(n_s)) + "\n")
    for p in pos:
        r = ref[p]
        alt = "ACGT"[("ACGT".index(r) + 1) % 4]
        g = rng.random(2 * n_s) < rng.random() ** 3
        gt = "\t".join(f"{int(a)}|{int(b)}" for a, b in zip(g[0::2], g[1::2]))
        fh.write(f"22\t{p + 1}\t.\t{r}\t{alt}\t.\t.\t.\tGT\t{gt}\n")
size = os.path.getsize(os.path.join(d, "v.vcf"))
for th in (1, 8, os.cpu_count()):
    t = time.perf_counter()
    idx = GraphIndex.from_fasta_vcf(os.path.join(d, "r.fa"), os.path.join(d, "v.vcf"), "22", threads=th)
    dt = time.perf_counter() - t
    print(f"threads={th}: {dt:.2f} s  {size / dt / 1e6:.0f} MB/s  {n_rec / dt:.0f} records/s  ({idx.n_haplotypes} haplotypes, {len(idx.pos)} sites)")
