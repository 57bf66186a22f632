"""TEST INFRASTRUCTURE (imports oracle/; not part of the product): one seed of the extraction fuzz -- a random conflict-free
graph of a given mix of allele kinds; the extraction kernels' rows against the per-haplotype brute force
(oracle/extract_bruteforce.py) and the walk enumerator (oracle/extract_oracle.py); the FUSED extraction -> scoring path
(compute_results_from_graph at threshold 1 with --recomb: every row is reported) against those same rows, scored here with
the motif's integer matrix.  Used by scripts/extract_fuzz.py (seed after seed for a fixed time) and by
tests/test_gpu_fused.py (a bounded seed set inside `pytest -m gpu`)."""
import contextlib
import io
import os
import sys

import numpy as np

from extract_helpers import make_consistent_graph_files

KINDS = ["s", "sd", "si", "sm", "sc", "sidm", "sD", "sO", "sidmDO", "sidmDOcS", "dc", "ic", "cS"]


class SynMotif:
    """the members the scoring path reads (grafimo_amd.motif.MOTIF_FIELDS); the DP runs on the device"""

    def __init__(self, W, seed):
        from grafimo_amd import synth
        rec = synth.synthetic_motif(W, np.random.default_rng(5000 + seed), np.array([0.3, 0.2, 0.2, 0.3]))
        self.score_matrix, self.nucsmap = rec["sm"], {n: i for i, n in enumerate("ACGT")}
        self.bg = {n: float(rec["bg"][i]) for i, n in enumerate("ACGT")}
        self.min_val, self.scale, self.offset, self.width = int(rec["min_val"]), int(rec["scale"]), np.double(rec["offset"]), W
        self.motif_id, self.motif_name = f"SYN{W}", f"syn{W}"


def check_fused(g, S, E, W, got, seed):
    """every row of `got` (already checked against the brute force) comes out of the fused path with its coordinates,
    haplotype count, ref flag and the score of its k-mer"""
    # THREE motifs over the same (graph, region, width): a plan's first call lists the windows that touch an insertion / deletion,
    # its second -- the item count is back -- also stores their walks, from the third on graph_score_kernel scores them from that
    # cache and no deletion kernel runs (gfm_graph_fused.hpp: LwMeta): every stage against the same brute-force rows
    for k in range(3):
        _check_fused_once(g, S, E, W, got, seed + 7919 * k)


def _check_fused_once(g, S, E, W, got, seed):
    from grafimo_amd.extract_regions import compute_results_from_graph
    from grafimo_amd.workflow import Findmotif
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


def fuzz_seed(seed, tmp, stats, log=None):
    """One graph, four (region, width) plans.  `stats`: dict with graphs / rows / carried / heavy / fused counters."""
    from grafimo_amd.extract_regions import DeviceGraph, GraphIndex
    from oracle import extract_bruteforce as bf
    from oracle import extract_oracle as xo
    kinds = KINDS[seed % len(KINDS)]
    n_samples = [3, 16, 40, 70][seed % 4]            # 6 .. 140 haplotypes: one to three bitset words
    fasta, vcf = make_consistent_graph_files(tmp, chrom="c", length=360, n_samples=n_samples, seed=seed, kinds=kinds,
                                             dense=seed % 3 != 0)
    ref = xo.read_fasta(fasta)["c"]
    recs, H = bf.read_vcf_records(vcf, "c")
    assert bf.consistent(ref, recs, H), (seed, kinds)
    v = xo.read_vcf_variants(vcf, "c")
    stderr = sys.stderr
    with open(os.devnull, "w") as devnull:
        sys.stderr = devnull
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
            stats["fused"] += 1
        if best >= 17.0:              # too many walks for the Python enumerator: the brute force still checks every row
            stats["heavy"] += 1
            freq, flags = bf.window_counts(ref, recs, H, S, E, W)
            carried, n = bf.check_rows(got, freq, flags)
            if log:
                log(f"heavy: seed {seed} kinds {kinds} region {S}-{E} W {W}: densest window 2^{best:.1f} allele combinations, "
                    f"{n} rows checked against the per-haplotype brute force")
            stats["rows"] += n
            stats["carried"] += carried
            continue
        freq, flags = bf.window_counts(ref, recs, H, S, E, W)
        try:
            carried, n = bf.check_rows(got, freq, flags)
        except AssertionError:
            if log:
                log(f"FAILED at seed {seed} kinds {kinds} samples {n_samples} region {S}-{E} W {W}")
            raise
        exp = xo.enumerate_region_variants("c", ref, v, S, E, W, with_counts=True)
        want = [(r[1].encode(), int(r[2].split(":")[1][:-1]), int(r[3].split(":")[1][:-1]), r[2][-1], r[4], r[5]) for r in exp]
        assert got == want, (seed, kinds, S, E, W, len(got), len(want),
                             next((i, a, b) for i, (a, b) in enumerate(zip(got, want)) if a != b) if len(got) == len(want) else None)
        stats["rows"] += n
        stats["carried"] += carried
    g.close()
    stats["graphs"] += 1
