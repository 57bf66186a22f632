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
  * PINNED by the reference's scoring fixture tests/test_data/input/width_19/scoring_test_input.tsv
    (704 rows of real `vg find -K 19 -E -H` output for 22:19723256-19723526): the local graph --
    270 reference bases, five SNPs, one 2-bp deletion, carrier counts -- is recoverable from the
    fixture itself, and `enumerate_region_graph` reproduces every row: haplotype counts of -H
    (haplotypes with every SNP allele of the walk, with every deletion it takes, and without any
    deletion whose bases it uses), deletion rows (same start, stop = start + W + deleted length,
    flag `ref` because only reference-path nodes are visited -- the case GRAFIMO's
    `ref -> non.ref if |stop-start| != W` rule repairs, score_sequences.py:305-307), node ids
    incl. chopping at 32 bases and the cuts a deletion makes;
  * NOT pinned (no vg binary here): insertions and other non-SNP, non-deletion records (skipped and
    counted), overlapping deletions (the later one is skipped), multi-allelic sites' node order
    beyond "alternates first", and the fate of a deletion-crossing walk whose stop lies beyond the
    region end (dropped here: a walk is reported only if both of its ends lie inside the region,
    which is what limits the plain windows to start <= E - W in expected_seqs.tsv).
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
        hap = np.asarray(hap, dtype=np.int8)
        self.hap = hap.reshape(len(self.pos), -1) if hap.size else \
            np.zeros((len(self.pos), hap.shape[-1] if hap.ndim > 1 else 0), np.int8)
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
                if len(gt) == 1:
                    gt = gt * 2
                # an allele number beyond the ALT list (malformed record) counts as the reference allele
                gts += [int(x) if x.isdigit() and int(x) <= len(a) else 0 for x in gt[:2]]
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


# =====================================================================================
# Graphs with deletions.  Pinned by the reference's scoring fixture: the local graph of
# 22:19723256-19723526 (270 reference bases, five SNPs and one 2-bp deletion, all recoverable from
# tests/test_data/input/width_19/scoring_test_input.tsv itself) reproduces all 704 rows of vg's
# `find -K 19 -E -H` output -- k-mers, coordinates, haplotype counts, ref flags and node paths
# (tests/test_extract_host.py::test_oracle_reproduces_vg_rows_of_the_scoring_fixture).
class Dels:
    """Deletions: anchor (0-based position of the base before the deleted ones), length, and per
    haplotype whether it carries the deletion.  Deleted bases are anchor+1 .. anchor+length."""

    def __init__(self, anchor, length, hap):
        self.anchor = np.asarray(anchor, dtype=np.int64)
        self.length = np.asarray(length, dtype=np.int64)
        hap = np.asarray(hap, dtype=bool)
        self.hap = hap.reshape(len(self.anchor), -1) if hap.size else np.zeros((len(self.anchor), hap.shape[-1] if hap.ndim > 1 else 0), bool)

    def __len__(self):
        return len(self.anchor)


def read_vcf_graph(path: str, chrom: Optional[str] = None):
    """-> (Sites, Dels, skipped).  SNP records as read_vcf_snps; a record REF = anchor base + deleted
    bases, ALT = anchor base is a deletion; it is left out (counted in skipped) when its anchor or
    span touches a deletion accepted before it.  Everything else (insertions, MNPs, ...) is skipped."""
    pos, ref, alts, hap = [], [], [], []
    d_anchor, d_len, d_hap, skipped, busy = [], [], [], 0, -1
    op = gzip.open if path.endswith(".gz") else open
    with op(path, "rt") as fh:
        for line in fh:
            if line.startswith("#"):
                continue
            f = line.rstrip("\n").split("\t")
            if chrom is not None and f[0] != chrom:
                continue
            r, a = f[3].upper(), f[4].upper().split(",")
            gts = []
            for s in f[9:]:
                gt = s.split(":")[0].replace("/", "|").split("|")
                if len(gt) == 1:
                    gt = gt * 2
                gts += [int(x) if x.isdigit() and int(x) <= len(a) else 0 for x in gt[:2]]
            p = int(f[1]) - 1
            if len(r) == 1 and len(a) <= 3 and all(len(x) == 1 and x in "ACGT" for x in a):
                if pos and p == pos[-1]:
                    skipped += 1
                    continue
                pos.append(p); ref.append(r); alts.append(a); hap.append(gts)
            elif len(r) > 1 and len(a) == 1 and a[0] == r[0] and p > busy:
                d_anchor.append(p); d_len.append(len(r) - 1); d_hap.append([g == 1 for g in gts])
                busy = p + len(r) - 1
            else:
                skipped += 1
    H = len(hap[0]) if hap else (len(d_hap[0]) if d_hap else 0)
    sites = Sites(pos, ref, alts, hap if hap else np.zeros((0, H), np.int8), skipped)
    return sites, Dels(d_anchor, d_len, d_hap if d_hap else np.zeros((0, H), bool)), skipped


class GraphNodeTable:
    """vg construct's node ids on a reference + SNP + deletion graph: the reference is cut at every
    SNP (the site is a node of its own: alternates numbered first, then the reference allele) and at
    both ends of every deleted stretch; what lies between two cuts is chopped into nodes of at most
    NODE_MAX bases.  `first_id` / `forced_cuts` let a test start inside a chromosome."""

    def __init__(self, ref_len: int, sites: Sites, dels: Dels, first_id: int = 1, forced_cuts=(),
                 node_max: int = NODE_MAX):
        cuts = {0, ref_len} | set(int(c) for c in forced_cuts)
        snp_at = {int(p): i for i, p in enumerate(sites.pos)}
        for p in snp_at:
            cuts |= {p, p + 1}
        for a, ln in zip(dels.anchor, dels.length):
            cuts |= {int(a) + 1, int(a) + int(ln) + 1}
        cuts = sorted(c for c in cuts if 0 <= c <= ref_len)
        self.ref_node = np.zeros(ref_len, dtype=np.int64)      # node id of the reference base at x
        self.alt_node = {}                                     # (site index, alt number 1..) -> id
        nid = first_id
        for b, e in zip(cuts[:-1], cuts[1:]):
            if e - b == 1 and b in snp_at:
                i = snp_at[b]
                for k in range(len(sites.alts[i])):
                    self.alt_node[(i, k + 1)] = nid
                    nid += 1
                self.ref_node[b] = nid
                nid += 1
                continue
            cur = b
            while cur < e:
                nxt = min(cur + node_max, e)
                self.ref_node[cur:nxt] = nid
                nid += 1
                cur = nxt


def enumerate_region_graph(chrom: str, ref: bytes, sites: Sites, dels: Dels, S: int, E: int, W: int,
                           with_counts: bool = False, nodes: Optional[GraphNodeTable] = None):
    """Rows of `vg find -p chrom:S-E -K W -E [-H]` on a graph with SNPs and deletions: every walk of
    W bases from every start p in [S, E-W].  Order inside a window: layout-major -- the jump vectors
    (at the anchor of a deletion a walk that needs more bases first continues on the reference, then
    takes the deletion) in lexicographic order, and on one layout the SNP alleles like
    itertools.product (reference first, last SNP fastest).  stop = reference coordinate after the last base; the flag
    is vg's (`ref` = only reference-path nodes: a taken deletion does not change it -- GRAFIMO's
    own rule flips such rows, score_sequences.py:305-307); count = haplotypes with every SNP allele
    of the walk, with every deletion it takes, and without any deletion whose bases it uses."""
    label = f"{chrom}:{S}-{E}"
    snp_at = {int(p): i for i, p in enumerate(sites.pos)}
    del_at = {int(a): j for j, a in enumerate(dels.anchor)}
    H = sites.n_haplotypes if len(sites.pos) else (dels.hap.shape[1] if len(dels) else 0)
    rows = []

    def emit(p, bases, used, snps, taken):
        last = used[-1]
        is_ref = "ref" if not any(a for _, a in snps) else "non.ref"
        count = 0
        if with_counts and H:
            ok = np.ones(H, dtype=bool)
            for i, a in snps:
                ok &= sites.hap[i] == a
            for j in taken:
                ok &= dels.hap[j]
            for j in range(len(dels)):
                lo, hi = int(dels.anchor[j]) + 1, int(dels.anchor[j]) + int(dels.length[j])
                if j not in taken and any(lo <= x <= hi for x in used):
                    ok &= ~dels.hap[j]
            count = int(ok.sum())
        path = []
        if nodes is not None:
            alt_of = {int(sites.pos[i]): (i, a) for i, a in snps if a}
            for x in used:
                nid = nodes.alt_node[alt_of[x]] if x in alt_of else int(nodes.ref_node[x])
                if not path or path[-1] != nid:
                    path.append(nid)
        kmer = bytes(bases)
        rows.append((label, kmer.decode(), f"{chrom}:{p}+", f"{chrom}:{last + 1}+", count, is_ref,
                     "".join(f"{n}+," for n in path)))
        rows.append((label, revcomp(kmer).decode(), f"{chrom}:{last + 1}-", f"{chrom}:{p}-", count, is_ref,
                     "".join(f"{n}-," for n in reversed(path))))

    def layouts(x, used, taken):
        """reference positions used by the walks from x (jump vectors in lexicographic order: at the
        anchor of a deletion first the walk that stays on the reference, then the one that jumps)"""
        if x >= len(ref):
            return
        nu = used + [x]
        if len(nu) == W:
            if x + 1 <= min(E, len(ref)):          # both ends of a walk lie inside the region
                yield nu, taken
            return
        yield from layouts(x + 1, nu, taken)
        j = del_at.get(x)
        if j is not None:
            yield from layouts(x + int(dels.length[j]) + 1, nu, taken + [j])

    for p in range(max(S, 0), min(E, len(ref)) - W + 1):
        for used, taken in layouts(p, [], []):
            on = [snp_at[x] for x in used if x in snp_at]
            for combo in itertools.product(*[range(1 + len(sites.alts[i])) for i in on]):   # last SNP fastest
                allele = dict(zip(on, combo))
                bases = [ref[x] if not allele.get(snp_at.get(x), 0) else ord(sites.alts[snp_at[x]][allele[snp_at[x]] - 1])
                         for x in used]
                emit(p, bases, used, list(zip(on, combo)), taken)
    return rows
