"""The walk enumerator of the extraction step (oracle/extract_oracle.py) against an independent algorithm: every
haplotype of the VCF materialised as a sequence, W-windows slid over it (oracle/extract_bruteforce.py).  A row's
haplotype count must be the number of haplotypes whose sequence holds that k-mer at those coordinates; rows no haplotype
carries must report 0.  tests/test_gpu_extract.py runs the same check on the HIP kernels' rows."""
import os

import pytest

from conftest import REF_DATA
from extract_helpers import make_consistent_graph_files


def _rows(tuples):
    """enumerator rows (label, kmer, 'c:start+', 'c:stop+', count, flag[, path]) -> (kmer, start, stop, strand, count, flag)"""
    for r in tuples:
        yield (r[1].encode(), int(r[2].split(":")[1][:-1]), int(r[3].split(":")[1][:-1]), r[2][-1], r[4], r[5])


def test_reference_test_graph_counts_from_first_principles():
    """tests/test_data/input/test.fa + test.vcf.gz (the graph behind the reference's own test_sequence_extraction,
    tests/grafimo_run_test.py:49-63): the -H counts of both oracles equal the per-haplotype brute force."""
    from oracle import extract_bruteforce as bf
    from oracle import extract_oracle as xo
    ref = xo.read_fasta(os.path.join(REF_DATA, "test.fa"))["x"]
    vcf = os.path.join(REF_DATA, "test.vcf.gz")
    recs, H = bf.read_vcf_records(vcf, "x")
    assert H == 2 and bf.consistent(ref, recs, H)
    sites = xo.read_vcf_snps(vcf, "x")
    v = xo.read_vcf_variants(vcf, "x")
    for S, E, W in [(0, 20, 19), (0, 50, 19), (5, 45, 8), (0, 50, 30), (0, 50, 1)]:
        freq, flags = bf.window_counts(ref, recs, H, S, E, W)
        assert len(freq) > 0
        carried, n = bf.check_rows(_rows(xo.enumerate_region("x", ref, sites, S, E, W, with_counts=True)), freq, flags)
        assert carried == len(freq) and n > 2 * carried - 1
        bf.check_rows(_rows(xo.enumerate_region_variants("x", ref, v, S, E, W, with_counts=True)), freq, flags)


@pytest.mark.parametrize("kinds", ["s", "sd", "si", "sm", "sidm", "id", "sD", "sO", "sidmDO", "c", "sc", "sS", "sidmDOcS"])
def test_enumerator_counts_equal_the_per_haplotype_brute_force(tmp_path, kinds):
    """Random conflict-free VCFs with every modelled allele kind: substitutions (multi-allelic, second records at a
    position), insertions (also several at one anchor, behind a substituted anchor), deletions, multi-base
    substitutions, complex alleles (c: the brute force substitutes ALT for REF, the enumerator walks the substitutions
    + indel they are taken apart into), records with symbolic ALTs (S: in neither); clustered so that windows hold
    several sites; regions that touch both chromosome ends."""
    from oracle import extract_bruteforce as bf
    from oracle import extract_oracle as xo
    total = 0
    for seed in range(3):
        fasta, vcf = make_consistent_graph_files(str(tmp_path), chrom="c", length=420, n_samples=16,
                                                 seed=100 * len(kinds) + seed, kinds=kinds)
        ref = xo.read_fasta(fasta)["c"]
        recs, H = bf.read_vcf_records(vcf, "c")
        assert H == 32 and bf.consistent(ref, recs, H)
        v = xo.read_vcf_variants(vcf, "c")
        assert (v.skipped > 0) == ("S" in kinds) and len(v) > 15     # only the alleles of symbolic records are left out
        for (S, E), W in [((0, 90), 19), ((100, 260), 8), ((250, 420), 30), ((30, 200), 3), ((395, 420), 12)]:
            freq, flags = bf.window_counts(ref, recs, H, S, E, W)
            rows = xo.enumerate_region_variants("c", ref, v, S, E, W, with_counts=True)
            carried, n = bf.check_rows(_rows(rows), freq, flags)
            total += carried
            if kinds in ("s", "sd"):                       # the deletion oracle pinned by the 704-row fixture says the same
                sites, dels, skipped = xo.read_vcf_graph(vcf, "c")
                if skipped == 0:
                    bf.check_rows(_rows(xo.enumerate_region_graph("c", ref, sites, dels, S, E, W, with_counts=True)),
                                  freq, flags)
    assert total > 1000


def test_brute_force_refuses_conflicting_haplotypes(tmp_path):
    from oracle import extract_bruteforce as bf
    head = "##fileformat=VCFv4.1\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\ta\n"
    (tmp_path / "x.vcf").write_text(head + "c\t3\t.\tGTA\tG\t.\t.\t.\tGT\t1|0\nc\t4\t.\tT\tC\t.\t.\t.\tGT\t1|1\n")
    recs, H = bf.read_vcf_records(str(tmp_path / "x.vcf"), "c")
    ref = b"ACGTACGTAC"
    assert not bf.consistent(ref, recs, H)          # haplotype 0 carries the deletion AND a base inside it
    seq, coord, ins, alt = bf.haplotype_sequence(ref, recs, 1)
    assert bytes(seq) == b"ACGCACGTAC" and coord == list(range(10)) and alt[3] and not any(ins)
    (tmp_path / "y.vcf").write_text(head + "c\t3\t.\tG\tT\t.\t.\t.\tGT\t1|0\nc\t3\t.\tG\tGAA\t.\t.\t.\tGT\t1|1\n"
                                    "c\t6\t.\tCGT\tC\t.\t.\t.\tGT\t0|1\n")
    recs, H = bf.read_vcf_records(str(tmp_path / "y.vcf"), "c")
    s0 = bf.haplotype_sequence(ref, recs, 0)
    s1 = bf.haplotype_sequence(ref, recs, 1)
    assert bytes(s0[0]) == b"ACTAATACGTAC" and s0[1] == [0, 1, 2, 2, 2, 3, 4, 5, 6, 7, 8, 9]
    assert bytes(s1[0]) == b"ACGAATACAC" and s1[1] == [0, 1, 2, 2, 2, 3, 4, 5, 8, 9] and s1[2][3] and s1[2][4]


def test_records_with_more_than_three_alt_alleles(tmp_path):
    """A substitution site has at most three alternates (four bases), but a RECORD may list more alleles -- an STR
    site with four insertion lengths beside a substitution.  Every one of them is part of the graph (insertions are
    sites of their own): the C++ reader equals the oracle reader, nothing is skipped, and the enumerator's counts
    equal the per-haplotype brute force."""
    from grafimo_amd.extract_regions import GraphIndex
    from oracle import extract_bruteforce as bf
    from oracle import extract_oracle as xo
    ref = "ACGTTGCAATCGGATCCATGCAAGTCTAGGCTTAACG"
    head = "##fileformat=VCFv4.1\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\ta\tb\tc\n"
    body = ["s\t9\t.\tA\tG,AT,ATT,ATTT,ATTTT\t.\t.\t.\tGT\t1|2\t3|4\t5|0",
            "s\t20\t.\tG\tGA,GAC,T,C,A,GACA\t.\t.\t.\tGT\t6|3\t4|5\t1|2"]
    (tmp_path / "m.vcf").write_text(head + "\n".join(body) + "\n")
    (tmp_path / "m.fa").write_text(">s\n" + ref + "\n")
    v = xo.read_vcf_variants(str(tmp_path / "m.vcf"), "s")
    assert v.skipped == 0 and sum(k == 1 for k in v.kind) == 7 and v.alts[[i for i, k in enumerate(v.kind) if k == 0][1]] == ["T", "C", "A"]
    idx = GraphIndex.from_fasta_vcf(str(tmp_path / "m.fa"), str(tmp_path / "m.vcf"), "s")
    assert idx.skipped == 0 and int((idx.ins_len > 0).sum()) == 7
    recs, H = bf.read_vcf_records(str(tmp_path / "m.vcf"), "s")
    assert H == 6 and bf.consistent(ref.encode(), recs, H)
    for W in (3, 8, 19):
        freq, flags = bf.window_counts(ref.encode(), recs, H, 0, len(ref), W)
        bf.check_rows(_rows(xo.enumerate_region_variants("s", ref.encode(), v, 0, len(ref), W, with_counts=True)), freq, flags)


def test_complex_allele_rows_by_hand(tmp_path):
    """REF=CGT ALT=TA at 0-based 5 of ACGTACGTACGT (carried by haplotype 0 only): the haplotype reads ACGTA TA ACGT.  The
    reader takes the allele apart into C>T at 5, G>A at 6 and a deletion of T behind 6; the brute force substitutes the
    whole ALT.  Rows carried by a haplotype agree; the enumerator's extra rows (the substitutions without the deletion,
    ...) belong to no haplotype."""
    from oracle import extract_bruteforce as bf
    from oracle import extract_oracle as xo
    head = "##fileformat=VCFv4.1\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\ta\n"
    (tmp_path / "x.vcf").write_text(head + "c\t6\t.\tCGT\tTA\t.\t.\t.\tGT\t1|0\n"
                                           "c\t10\t.\tC\t<DEL>\t.\t.\t.\tGT\t1|1\n")
    ref = b"ACGTACGTACGT"
    recs, H = bf.read_vcf_records(str(tmp_path / "x.vcf"), "c")
    assert H == 2 and len(recs) == 1                      # the symbolic record is in no graph
    seq, coord, ins, alt = bf.haplotype_sequence(ref, recs, 0)
    assert bytes(seq) == b"ACGTATAACGT" and coord == [0, 1, 2, 3, 4, 5, 6, 8, 9, 10, 11] and not any(ins)
    assert alt == [False] * 5 + [True, True] + [False] * 4
    v = xo.read_vcf_variants(str(tmp_path / "x.vcf"), "c")
    assert list(zip(v.pos, v.kind)) == [(5, 0), (6, 0), (6, 2)] and v.length[2] == 1 and v.skipped == 1
    for W in (3, 4, 6):
        freq, flags = bf.window_counts(ref, recs, H, 0, 12, W)
        rows = xo.enumerate_region_variants("c", ref, v, 0, 12, W, with_counts=True)
        carried, n = bf.check_rows(_rows(rows), freq, flags)
        assert carried == len(freq) and n > 2 * carried    # recombinant walks of the three sites: count 0
    # an insertion-type complex allele: REF=A ALT=TGG at 4 -> A>T at 4 and GG inserted behind it
    (tmp_path / "y.vcf").write_text(head + "c\t5\t.\tA\tTGG\t.\t.\t.\tGT\t0|1\n")
    recs, H = bf.read_vcf_records(str(tmp_path / "y.vcf"), "c")
    seq, coord, ins, alt = bf.haplotype_sequence(ref, recs, 1)
    assert bytes(seq) == b"ACGTTGGCGTACGT" and coord[4:8] == [4, 4, 4, 5] and ins[4:8] == [False, True, True, False]
    v = xo.read_vcf_variants(str(tmp_path / "y.vcf"), "c")
    assert list(zip(v.pos, v.kind)) == [(4, 0), (4, 1)] and v.seq[1] == b"GG"
    freq, flags = bf.window_counts(ref, recs, H, 0, 12, 5)
    bf.check_rows(_rows(xo.enumerate_region_variants("c", ref, v, 0, 12, 5, with_counts=True)), freq, flags)
