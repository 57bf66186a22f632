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
  * NOT pinned (no vg binary here): everything about insertions and multi-base substitutions (modelled since
    round 2 by read_vcf_variants / enumerate_region_variants at the end of this file, whose header states
    what is assumed; complex alleles are taken apart there into substitutions + one indel, records with a symbolic
    ALT are left out whole as `vg construct` without --handle-sv does), the rows of overlapping deletions,
    haplotype counts beyond consistency with the scoring fixture (but see oracle/extract_bruteforce.py, which
    derives counts and row sets from the haplotype sequences alone), multi-allelic sites' node order
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


# =====================================================================================
# Graphs with insertions and multi-base substitutions (round 2).  **UNPINNED**: no file of the reference
# shows what `vg find` prints for a k-mer that runs through, starts or ends inside inserted bases, and vg is
# not in this image.  What follows is the natural extension of the rules the two fixtures pin (an insertion is
# an alternate node behind its anchor base that consumes window bases but no reference span -- like a taken
# deletion consumes reference span but no window bases), stated here so that the HIP kernels have something
# exact to be compared with (tests/test_gpu_extract.py) and a reader knows what was assumed:
#   * VCF records are taken apart per ALT allele: a single-base substitution, a deletion (REF = anchor +
#     deleted bases, ALT = anchor), an insertion (REF = anchor, ALT = anchor + inserted bases), a
#     multi-base substitution of equal length (one substitution per mismatching position, all carried by
#     the same haplotypes: `vg construct` aligns ALT to REF and cuts the graph at every edit); anything else
#     is skipped and counted, like the ALT alleles of a record beyond its sixteenth.  Before that every ALT is
#     NORMALISED against REF the way VCF normalisation does it per allele: common trailing bases are dropped while
#     both strings keep one base, then common leading bases (the position moves right) -- so the alleles of a
#     short-tandem-repeat record (REF=ATTT ALT=A,AT,ATT,ATTTT) become deletions of 3, 2, 1 bases and an insertion, all
#     anchored on the record's first base.  Substitutions at one position become ONE site with up to three
#     alternates (a fourth is skipped).  Round 3: deletions may OVERLAP -- several at one anchor (different lengths:
#     the STR case), anchors inside another deletion's span -- the graph simply holds all of them; two records that
#     delete the same bases merge their carriers.
#   * walks start on a reference position p (first base: the reference base or a substitution), or on base
#     t of an insertion anchored at p - 1 (start coordinate p).  Behind the base at position x, with bases
#     still to go, the walk decides for every insertion anchored at x (in site order) whether to read it
#     (0 = no, 1 = yes; after a yes nothing else anchored at x is considered), then for every deletion
#     anchored at x (site order) whether to jump it (a yes ends the site).  Decision vectors in lexicographic order, on one vector the
#     substitution alleles like itertools.product; walks without a start inside an insertion first, then
#     per insertion anchored at p - 1 (site order) the offsets t = 0, 1, ...
#   * stop = reference coordinate behind the last reference base consumed (the anchor, for a walk that
#     ends inside an insertion; p for a walk that never leaves the insertion it started in); a walk is
#     reported if p >= S and stop <= E; windows start at p in [S, E - 1] (E - W without insertions).
#   * flag: `ref` iff no alternate substitution and no insertion was taken (a taken deletion keeps it: pinned).
#   * count: haplotypes with every substitution allele of the walk, every insertion and deletion it takes,
#     none of the insertions it decided against, none of the deletions whose bases it uses.
class Variants:
    """Sites of one chromosome in graph order (position, then substitution < insertion < deletion).
    kind 0 substitution site (alts: 1..3 bases), 1 insertion (seq), 2 deletion (length);
    carriers[i][k] = bool [H]: haplotypes with alternate k of site i (insertions / deletions: k = 0)."""

    def __init__(self):
        self.pos, self.kind, self.alts, self.seq, self.length, self.carriers = [], [], [], [], [], []
        self.skipped = 0
        self.n_haplotypes = 0

    def add(self, pos, kind, alts=(), seq=b"", length=0, carriers=()):
        self.pos.append(int(pos)); self.kind.append(kind); self.alts.append(list(alts)); self.seq.append(bytes(seq))
        self.length.append(int(length)); self.carriers.append([np.asarray(c, dtype=bool) for c in carriers])

    def __len__(self):
        return len(self.pos)


def read_vcf_variants(path: str, chrom: Optional[str] = None, ref: Optional[bytes] = None) -> Variants:
    atoms = []      # (pos, kind, payload, carriers, order)
    skipped, H = 0, 0
    op = gzip.open if path.endswith(".gz") else open
    with op(path, "rt") as fh:
        for line in fh:
            if line.startswith("#"):
                continue
            f = line.rstrip("\n").split("\t")
            if chrom is not None and f[0] != chrom:
                continue
            r, alts = f[3].upper(), f[4].upper().split(",")
            gts = []
            for s in f[9:]:
                gt = s.split(":")[0].replace("/", "|").split("|")
                if len(gt) == 1:
                    gt = gt * 2
                gts += [int(x) if x.isdigit() and int(x) <= len(alts) else 0 for x in gt[:2]]
            gts = np.asarray(gts, dtype=np.int64)
            p = int(f[1]) - 1
            before = len(atoms)
            # a record with an ALT that is no string of A, C, G, T (symbolic, breakend, '*') is left out whole --
            # vg construct without --handle-sv (constructVG.py:332) skips it with a warning
            if not (len(r) > 0 and all(c in "ACGTN" for c in r)
                    and all(len(a) > 0 and all(c in "ACGT" for c in a) for a in alts[:64])):
                skipped += len(alts)
                continue
            for k, a in enumerate(alts):
                car = gts == k + 1
                ok = k < 64
                rr, aa, pp = r, a, p
                if ok:                                   # per-allele normalisation: trailing, then leading bases
                    while len(rr) > 1 and len(aa) > 1 and rr[-1] == aa[-1]:
                        rr, aa = rr[:-1], aa[:-1]
                    while len(rr) > 1 and len(aa) > 1 and rr[0] == aa[0]:
                        rr, aa, pp = rr[1:], aa[1:], pp + 1
                if ok and len(rr) == 1 and len(aa) == 1:
                    if rr != aa:
                        atoms.append((pp, 0, aa, car))
                elif ok and len(rr) > 1 and len(aa) == 1 and aa[0] == rr[0]:
                    atoms.append((pp, 2, len(rr) - 1, car))
                elif ok and len(rr) == 1 and len(aa) > 1 and aa[0] == rr[0]:
                    atoms.append((pp, 1, aa[1:].encode(), car))
                elif ok and len(rr) == len(aa) and len(rr) > 1:
                    for j, (x, y) in enumerate(zip(rr, aa)):
                        if x != y:
                            atoms.append((pp + j, 0, y, car))
                elif ok:
                    # a complex allele: substitutions over the common length, the rest an insertion / a deletion
                    # behind the last of those bases, all with the allele's carriers (vg decomposes by alignment;
                    # the haplotype sequences are the same either way)
                    n = min(len(rr), len(aa))
                    for j in range(n):
                        if rr[j] != aa[j]:
                            atoms.append((pp + j, 0, aa[j], car))
                    if len(aa) > len(rr):
                        atoms.append((pp + n - 1, 1, aa[n:].encode(), car))
                    else:
                        atoms.append((pp + n - 1, 2, len(rr) - n, car))
                else:
                    skipped += 1
            if len(atoms) > before:
                H = max(H, len(gts))
    order = sorted(range(len(atoms)), key=lambda i: (atoms[i][0], atoms[i][1]))
    v = Variants()
    v.n_haplotypes = H
    for i in order:
        p, kind, payload, car = atoms[i]
        car = np.concatenate([car, np.zeros(H - len(car), bool)]) if len(car) < H else car
        if kind == 0:
            if len(v) and v.kind[-1] == 0 and v.pos[-1] == p:      # same position: one site, more alternates
                if payload in v.alts[-1]:
                    k = v.alts[-1].index(payload)
                    v.carriers[-1][k] = v.carriers[-1][k] | car
                elif len(v.alts[-1]) < 3:
                    v.alts[-1].append(payload); v.carriers[-1].append(car)
                else:
                    skipped += 1
            else:
                v.add(p, 0, alts=[payload], carriers=[car])
        elif kind == 1:
            dup = [j for j in range(len(v)) if v.pos[j] == p and v.kind[j] == 1 and v.seq[j] == payload]
            if dup:
                v.carriers[dup[0]][0] = v.carriers[dup[0]][0] | car
            else:
                v.add(p, 1, seq=payload, carriers=[car])
        else:
            dup = [j for j in range(len(v)) if v.pos[j] == p and v.kind[j] == 2 and v.length[j] == payload]
            if dup:
                v.carriers[dup[0]][0] = v.carriers[dup[0]][0] | car
            else:
                v.add(p, 2, length=payload, carriers=[car])
    v.skipped = skipped
    return v


def enumerate_region_variants(chrom: str, ref: bytes, v: Variants, S: int, E: int, W: int,
                              with_counts: bool = False):
    """Rows (label, kmer, start, stop, count, flag) x 2 strands per walk, in the order stated above; the
    node-path column is left empty (vg's node numbering around insertions is not known)."""
    label = f"{chrom}:{S}-{E}"
    at: Dict[int, List[int]] = {}
    for i, p in enumerate(v.pos):
        at.setdefault(p, []).append(i)
    has_ins = any(k == 1 for k in v.kind)
    del_sites = [i for i in range(len(v)) if v.kind[i] == 2]
    H = v.n_haplotypes
    L = len(ref)
    rows = []

    def emit(p, bases, last, subs, took, passed, used):
        stop = last + 1
        if stop > min(E, L):
            return
        flag = "ref" if not any(a for _, a in subs) and not any(v.kind[i] == 1 for i in took) else "non.ref"
        count = 0
        if with_counts and H:
            ok = np.ones(H, dtype=bool)
            for i, a in subs:
                if a:
                    ok &= v.carriers[i][a - 1]
                else:
                    for c in v.carriers[i]:
                        ok &= ~c
            for i in took:
                ok &= v.carriers[i][0]
            for i in passed:
                ok &= ~v.carriers[i][0]
            for i in del_sites:
                if i not in took:
                    lo, hi = v.pos[i] + 1, v.pos[i] + v.length[i]
                    if any(lo <= x <= hi for x in used):
                        ok &= ~v.carriers[i][0]
            count = int(ok.sum())
        kmer = bytes(bases)
        rows.append((label, kmer.decode(), f"{chrom}:{p}+", f"{chrom}:{stop}+", count, flag))
        rows.append((label, revcomp(kmer).decode(), f"{chrom}:{stop}-", f"{chrom}:{p}-", count, flag))

    def layouts(x, n, plan, used, took, passed):
        """yields (plan, used, took, passed, last): plan = per window base either ('r', x) a reference position
        or ('i', site, t) an inserted base"""
        if x >= L:
            return
        plan = plan + [("r", x)]
        used = used + [x]
        n += 1
        if n == W:
            yield plan, used, took, passed, x
            return
        here = at.get(x, [])
        ins = [i for i in here if v.kind[i] == 1]
        dele = [i for i in here if v.kind[i] == 2]

        def after_insertions(k, passed_now):
            # decisions for insertions ins[k:], all answered "no" so far; then the deletion
            if k == len(ins):
                # the deletions anchored here, in site order: jump this one?  (0 before 1; a yes ends the site)
                def after_deletions(d):
                    if d == len(dele):
                        yield from layouts(x + 1, n, plan, used, took, passed_now)
                        return
                    yield from after_deletions(d + 1)
                    j = dele[d]
                    yield from layouts(x + v.length[j] + 1, n, plan, used, took + [j], passed_now)
                yield from after_deletions(0)
                return
            i = ins[k]
            yield from after_insertions(k + 1, passed_now + [i])          # 0: do not read insertion i
            take = min(len(v.seq[i]), W - n)                                # 1: read it
            plan2 = plan + [("i", i, t) for t in range(take)]
            if n + take == W:
                yield plan2, used, took + [i], passed_now, x
            else:
                yield from layouts(x + 1, n + take, plan2, used, took + [i], passed_now)

        yield from after_insertions(0, passed)

    def walks_from(p):
        yield from layouts(p, 0, [], [], [], [])
        for i in at.get(p - 1, []):
            if v.kind[i] != 1:
                continue
            for t in range(len(v.seq[i])):
                take = min(len(v.seq[i]) - t, W)
                plan = [("i", i, t + j) for j in range(take)]
                if take == W:
                    yield plan, [], [i], [], p - 1
                else:
                    yield from layouts(p, take, plan, [], [i], [])

    hi = min(E, L) - (1 if has_ins else W)
    for p in range(max(S, 0), hi + 1):
        for plan, used, took, passed, last in walks_from(p):
            subs_sites = [i for x in used for i in at.get(x, []) if v.kind[i] == 0]
            for combo in itertools.product(*[range(1 + len(v.alts[i])) for i in subs_sites]):
                allele = dict(zip(subs_sites, combo))
                bases = []
                for step in plan:
                    if step[0] == "i":
                        bases.append(v.seq[step[1]][step[2]])
                    else:
                        x = step[1]
                        s = [i for i in at.get(x, []) if v.kind[i] == 0]
                        a = allele.get(s[0], 0) if s else 0
                        bases.append(ord(v.alts[s[0]][a - 1]) if a else ref[x])
                emit(p, bases, last, list(zip(subs_sites, combo)), took, passed, used)
    return rows
