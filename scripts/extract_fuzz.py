"""Development aid (GPU box): the extraction kernels against the per-haplotype brute force (oracle/extract_bruteforce.py)
and the walk enumerator (oracle/extract_oracle.py) on random conflict-free graphs of every allele kind, seed after seed
for a fixed time; and the FUSED extraction -> scoring path (compute_results_from_graph at threshold 1 with --recomb: every
row is reported) against those same rows, scored here with the motif's integer matrix.
TEST INFRASTRUCTURE (imports oracle/): not part of the product.
    python scripts/extract_fuzz.py [seconds] [first_seed]"""
import os
import sys
import tempfile
import time

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "tests"))
import contextlib
import io

import numpy as np

from extract_helpers import make_consistent_graph_files
from grafimo_amd import synth
from grafimo_amd.extract_regions import DeviceGraph, GraphIndex, compute_results_from_graph
from grafimo_amd.workflow import Findmotif
from oracle import extract_bruteforce as bf
from oracle import extract_oracle as xo

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
kinds_list = ["s", "sd", "si", "sm", "sc", "sidm", "sD", "sO", "sidmDO", "sidmDOcS", "dc", "ic", "cS"]
t0 = time.time()
n_graphs = n_rows = n_carried = n_heavy = n_fused = 0
devnull = open(os.devnull, "w")


class SynMotif:
    """the members the scoring path reads (grafimo_amd.motif.MOTIF_FIELDS); the DP runs on the device"""

    def __init__(self, W, seed):
        rec = synth.synthetic_motif(W, np.random.default_rng(5000 + seed), np.array([0.3, 0.2, 0.2, 0.3]))
        self.score_matrix, self.nucsmap = rec["sm"], {n: i for i, n in enumerate("ACGT")}
        self.bg = {n: float(rec["bg"][i]) for i, n in enumerate("ACGT")}
        self.min_val, self.scale, self.offset, self.width = int(rec["min_val"]), int(rec["scale"]), np.double(rec["offset"]), W
        self.motif_id, self.motif_name = f"SYN{W}", f"syn{W}"


def check_fused(g, S, E, W, got, seed):
    """every row of `got` (already checked against the brute force) comes out of the fused path with its coordinates,
    haplotype count, ref flag and the score of its k-mer"""
    m = SynMotif(W, seed)
    with contextlib.redirect_stdout(io.StringIO()):
        df = compute_results_from_graph(m, g, [(S, E)], True, Findmotif(threshold=1.0, recomb=True))
    sm = np.asarray(m.score_matrix, dtype=np.int64).reshape(4, W)
    code = np.full(256, -1, dtype=np.int64)
    for i, c in enumerate(b"ACGT"):
        code[c] = i
    want = []
    for km, st, sp, sd, fr, rf in got:
        c = code[np.frombuffer(km, dtype=np.uint8)]
        sc = int(m.min_val) if (c < 0).any() else int(sm[c, np.arange(W)].sum())
        rf = "ref" if rf == "ref" and abs(sp - st) == W else "non.ref"          # score_sequences.py:305-307
        want.append((km.decode(), st, sp, sd, fr, rf, float(sc) / float(m.scale) + float(W) * float(m.offset)))
    have = list(zip(df["matched_sequence"].tolist(), df["start"].tolist(), df["stop"].tolist(), df["strand"].tolist(),
                    df["haplotype_frequency"].tolist(), df["reference"].tolist(), df["score"].tolist()))
    # p < 1 is strict (resultsTmp.py:303): a k-mer with the lowest score the matrix can give has p = 1 and is not reported
    floor = float(int(sm.min(0).sum())) / float(m.scale) + float(W) * float(m.offset)
    may_miss = sorted(r for r in want if r[6] <= floor)
    want = sorted(r for r in want if r[6] > floor)
    have = sorted(r for r in have if r[6] > floor)
    assert have == want, (seed, S, E, W, len(have), len(want), next(((a, b) for a, b in zip(have, want) if a != b), None))
    assert len(df) - len(have) <= len(may_miss)


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
            best = 0.0                    # log2 of the allele product of the densest window
            for p0 in range(S, E):
                inside = (idx.pos >= p0) & (idx.pos < p0 + W)
                best = max(best, float(np.log2(1.0 + idx.n_alts[inside]).sum()))
            got = hip_rows(g.extract([(S, E)], W))       # (round 3 refused plans with a window of more than 2^20 walks)
            if 0 < len(got) <= 400_000:
                check_fused(g, S, E, W, got, seed)
                n_fused += 1
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
            try:
                carried, n = bf.check_rows(got, freq, flags)
            except AssertionError:
                print(f"FAILED at seed {seed} kinds {kinds} samples {n_samples} region {S}-{E} W {W}", flush=True)
                raise
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
      f"kernels == enumerator == brute force, {n_fused} plans also through the fused scoring path ({n_heavy} plans with a window of more than 2^17 walks: brute force only); next seed {seed}")
