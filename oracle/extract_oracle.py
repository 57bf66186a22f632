"""CPU restatement of the k-mer extraction step -- TEST INFRASTRUCTURE ONLY (see oracle/oracle.py).

What is restated: the row semantics of ``vg find -p CHR:S-E -x XG [-H GBWT] -K W -E`` as GRAFIMO
calls it (src/grafimo/extract_regions.py:180,225,326) and consumes it
(src/grafimo/score_sequences.py:273-307), for variation graphs built from a linear reference plus
a phased VCF of single-nucleotide variants -- the inputs of ``vg construct -r REF -v VCF`` /
``vg index -G`` in src/grafimo/constructVG.py:332,394.

The algorithm itself lives in vg (not under /root/reference; Dockerfile pins
quay.io/vgteam/vg:v1.27.1).  Pinning:
  * PINNED by the reference's own golden file tests/test_data/expected_results/expected_seqs.tsv
    (32 rows, region x:0-20, W=19, test graph = tests/test_data/input/test.fa + test.vcf.gz; used by
    tests/grafimo_run_test.py:49-63): k-mer strings, start/stop strings, strand handling, the
    ref / non.ref flag and the node-id paths of every walk through the SNP bubbles;
  * NOT pinned (no vg binary here, "parity unpinned"): the haplotype counts of ``-H`` (restated as
    "number of phased haplotypes of the VCF that carry every allele of the walk", which reproduces
    the structure seen in the reference's scoring fixture: 5096 on invariant windows, n / 5096-n on
    the two arms of a SNP), node chopping at 32 bp, and anything involving indels (records whose
    REF or ALT is not a single base are skipped and counted).
What the reference's scoring fixture (tests/test_data/input/width_19/scoring_test_input.tsv) shows about
deletions, for the round that adds them: a walk that skips a deleted reference node (849133+,849135+ around
22:19723468, a 2-bp deletion carried by 1 of 5096 haplotypes) keeps its start, reports stop = start + W + 2
(reference coordinate after its last base), and is labelled `ref` by vg because every node it visits is on
the reference path -- which is exactly what GRAFIMO's `ref -> non.ref if |stop-start| != W` rule repairs
(score_sequences.py:305-307).  Insertions do not occur in that fixture.
"""
import gzip
import itertools
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

NODE_MAX = 32          # vg construct -m default (node length limit)
_COMP = bytes.maketrans(b"ACGTNacgtn", b"TGCANtgcan")


def read_fasta(path: str) -> Dict[str, bytes]:
    seqs, name, parts = {}, None, []
    with open(path, "rb") as fh:
        for line in fh:
            line = line.strip()
            if not line:
                continue
            if line.startswith(b">"):
                if name is not None:
                    seqs[name] = b"".join(parts)
                name, parts = line[1:].split()[0].decode(), []
            else:
                parts.append(line.upper())
    if name is not None:
        seqs[name] = b"".join(parts)
    return seqs


class Sites:
    """SNP sites of one chromosome: pos (0-based, ascending), ref base, alt bases (<= 3),
    hap[i, h] = allele index (0 = ref) of haplotype h at site i (two haplotypes per sample)."""

    def __init__(self, pos, ref, alts, hap, skipped=0):
        self.pos = np.asarray(pos, dtype=np.int64)
        self.ref = list(ref)
        self.alts = [list(a) for a in alts]
        self.hap = np.asarray(hap, dtype=np.int8).reshape(len(self.pos), -1)
        self.skipped = int(skipped)

    @property
    def n_haplotypes(self):
        return self.hap.shape[1]


def read_vcf_snps(path: str, chrom: Optional[str] = None) -> Sites:
    pos, ref, alts, hap, skipped = [], [], [], [], 0
    op = gzip.open if path.endswith(".gz") else open
    with op(path, "rt") as fh:
        for line in fh:
            if line.startswith("#"):
                continue
            f = line.rstrip("\n").split("\t")
            if chrom is not None and f[0] != chrom:
                continue
            r, a = f[3].upper(), f[4].upper().split(",")
            if len(r) != 1 or any(len(x) != 1 or x not in "ACGT" for x in a) or len(a) > 3:
                skipped += 1
                continue
            gts = []
            for s in f[9:]:
                gt = s.split(":")[0].replace("/", "|").split("|")
                gts += [int(x) if x.isdigit() else 0 for x in (gt + gt)[:2]] if len(gt) == 1 else \
                       [int(x) if x.isdigit() else 0 for x in gt[:2]]
            pos.append(int(f[1]) - 1)
            ref.append(r)
            alts.append(a)
            hap.append(gts)
    order = np.argsort(pos, kind="stable")
    return Sites([pos[i] for i in order], [ref[i] for i in order], [alts[i] for i in order],
                 [hap[i] for i in order] if hap and hap[0] else np.zeros((len(pos), 0), np.int8), skipped)


class NodeTable:
    """Node ids as `vg construct` numbers them on a SNP-only graph (pinned by expected_seqs.tsv as
    far as that file goes): walking the reference left to right, an invariant stretch becomes nodes
    of at most NODE_MAX bases; at a site the alternate alleles get ids first (in ALT order), then
    the reference allele."""

    def __init__(self, ref_len: int, sites: Sites, node_max: int = NODE_MAX):
        self.seg_start: List[int] = []      # invariant segments: start, end (exclusive), id
        self.seg_end: List[int] = []
        self.seg_id: List[int] = []
        self.site_ids: List[List[int]] = [] # per site: [ref id, alt ids...]
        nid, cur = 1, 0
        for p, alts in zip(sites.pos, sites.alts):
            p = int(p)
            while cur < p:
                e = min(cur + node_max, p)
                self.seg_start.append(cur); self.seg_end.append(e); self.seg_id.append(nid)
                nid += 1
                cur = e
            alt_ids = list(range(nid, nid + len(alts)))
            nid += len(alts)
            self.site_ids.append([nid] + alt_ids)
            nid += 1
            cur = p + 1
        while cur < ref_len:
            e = min(cur + node_max, ref_len)
            self.seg_start.append(cur); self.seg_end.append(e); self.seg_id.append(nid)
            nid += 1
            cur = e
        self._seg_start = np.asarray(self.seg_start, dtype=np.int64)

    def path(self, sites: Sites, p: int, W: int, first_site: int, alleles: Sequence[int]) -> List[int]:
        """node ids of the walk covering [p, p+W) that takes alleles[k] at site first_site+k"""
        out, cur, k = [], p, 0
        end = p + W
        while cur < end:
            si = first_site + k
            if k < len(alleles) and int(sites.pos[si]) == cur:
                out.append(self.site_ids[si][alleles[k]])
                cur += 1
                k += 1
            else:
                j = int(np.searchsorted(self._seg_start, cur, side="right")) - 1
                out.append(self.seg_id[j])
                cur = self.seg_end[j]
        return out


def revcomp(kmer: bytes) -> bytes:
    return kmer.translate(_COMP)[::-1]


def enumerate_region(chrom: str, ref: bytes, sites: Sites, S: int, E: int, W: int,
                     with_counts: bool = False, nodes: Optional[NodeTable] = None
                     ) -> List[Tuple[str, str, str, str, int, str, str]]:
    """Rows of `vg find -p chrom:S-E -K W -E [-H]`: for every start p in [S, E-W] and every walk
    through the SNP bubbles of [p, p+W): the forward row and its reverse complement
    (start/stop swapped, '-' strand), count = haplotypes carrying the walk (0 without -H)."""
    label = f"{chrom}:{S}-{E}"
    rows = []
    for p in range(S, E - W + 1):
        if p < 0 or p + W > len(ref):
            continue
        i0 = int(np.searchsorted(sites.pos, p, side="left"))
        i1 = int(np.searchsorted(sites.pos, p + W, side="left"))
        choices = [range(1 + len(sites.alts[i])) for i in range(i0, i1)]
        for combo in itertools.product(*choices):
            k = bytearray(ref[p:p + W])
            for i, a in zip(range(i0, i1), combo):
                if a:
                    k[int(sites.pos[i]) - p] = ord(sites.alts[i][a - 1])
            is_ref = "ref" if not any(combo) else "non.ref"
            count = 0
            if with_counts and sites.n_haplotypes:
                ok = np.ones(sites.n_haplotypes, dtype=bool)
                for i, a in zip(range(i0, i1), combo):
                    ok &= sites.hap[i] == a
                count = int(ok.sum())
            path = nodes.path(sites, p, W, i0, combo) if nodes is not None else []
            fwd = "".join(f"{n}+," for n in path)
            rev = "".join(f"{n}-," for n in reversed(path))
            kmer = bytes(k)
            rows.append((label, kmer.decode(), f"{chrom}:{p}+", f"{chrom}:{p + W}+", count, is_ref, fwd))
            rows.append((label, revcomp(kmer).decode(), f"{chrom}:{p + W}-", f"{chrom}:{p}-", count, is_ref, rev))
    return rows
