"""Per-haplotype brute force for the k-mer extraction step -- TEST INFRASTRUCTURE ONLY (see oracle/oracle.py).

`oracle/extract_oracle.py` and the HIP kernels (grafimo_amd/csrc/graph_extract.hip) both ENUMERATE WALKS of the
variation graph and then count, per walk, the haplotypes that are compatible with it (AND of allele bitsets).  This
file derives the same rows from first principles, with no graph and no walk enumeration:

  * `vg find -p CHR:S-E -x XG -H GBWT -K W -E` (src/grafimo/extract_regions.py:180,225) reports, per k-mer walk, how
    many GBWT threads -- haplotypes -- contain it; a haplotype's thread spells that haplotype's own sequence;
  * so: materialise every haplotype of the VCF as a linear sequence (the reference with the ALT alleles it carries
    substituted in, a left-to-right pass over the records), remember for every base which reference coordinate it is
    aligned to -- (x) for a matched or substituted base, (anchor) for an inserted one -- slide a W-window over the
    sequence, and key every window by (k-mer, start, stop): start = coordinate of the first base (anchor + 1 for an
    inserted base), stop = coordinate behind the last reference base the window has consumed (last aligned
    coordinate + 1; for a window that ends on inserted bases that is anchor + 1) -- the TSV columns GRAFIMO parses
    (src/grafimo/score_sequences.py:279-293);
  * freq(key) = number of (haplotype, window) pairs that produce the key.  By the definition of a GBWT thread every
    key with freq > 0 must be among the enumerator's rows with exactly that count (summed over the rows sharing the
    key: two walks can spell the same k-mer over the same coordinates), and rows only the enumerator emits must
    have count 0 (recombinant walks: vg -E lists them, no haplotype carries them).

What this pins, independently of the walk enumerators: the COUNTING (which allele constraints a walk carries, also
the negative ones: reference alleles, insertions passed by, deletions whose bases are used) and the COMPLETENESS of the
row set (every window of every haplotype is there, with the same coordinates), for SNPs, multi-allelic sites,
multi-base substitutions, insertions and deletions.  What it does NOT pin: that vg prints these coordinates for
k-mers that start or end inside inserted bases (the convention above is the enumerators' own, stated in
oracle/extract_oracle.py), node ids, and the order of rows -- those need vg's own output.

Genotypes must be conflict-free per haplotype: the alleles one haplotype carries have disjoint REF spans, except
that a single-base substitution may sit on the anchor base of an insertion or deletion carried too.  (Overlapping
alleles on one haplotype have no defined sequence; `consistent` checks it.)
"""
import gzip
from collections import defaultdict
from typing import Dict, List, Optional, Tuple

_COMP = bytes.maketrans(b"ACGTNacgtn", b"TGCANtgcan")


def read_vcf_records(path: str, chrom: Optional[str] = None):
    """-> (records, H): records = [(pos0, REF, [ALT...], [allele index per haplotype])] in file order; two haplotypes
    per sample ("a|b", "a/b" as written, a lone allele doubled, "." = reference)."""
    recs, H = [], 0
    op = gzip.open if path.endswith(".gz") else open
    with op(path, "rt") as fh:
        for line in fh:
            if line.startswith("#"):
                continue
            f = line.rstrip("\n").split("\t")
            if chrom is not None and f[0] != chrom:
                continue
            alts = f[4].upper().split(",")
            if any(not a or set(a) - set("ACGT") for a in alts):
                continue          # a record with a symbolic ALT is not in the graph (vg construct without --handle-sv)
            gts: List[int] = []
            for s in f[9:]:
                gt = s.split(":")[0].replace("/", "|").split("|")
                if len(gt) == 1:
                    gt = gt * 2
                gts += [int(x) if x.isdigit() and int(x) <= len(alts) else 0 for x in gt[:2]]
            recs.append((int(f[1]) - 1, f[3].upper(), alts, gts))
            H = max(H, len(gts))
    return recs, H


def haplotype_sequence(ref: bytes, recs, h: int):
    """The sequence of haplotype h and, per base, the reference coordinate it is aligned to and whether it is an
    inserted base: -> (bases bytearray, coord list[int], inserted list[bool], altered list[bool]).  Raises ValueError
    on alleles with overlapping REF spans (other than a single-base substitution on an indel's anchor)."""
    out, coord, ins, alt = bytearray(), [], [], []
    cur = 0                      # next reference position to copy
    last_sub = -1                # position of the last single-base substitution applied (may be an indel's anchor)

    def copy_ref(upto):
        nonlocal cur
        while cur < upto:
            out.append(ref[cur]); coord.append(cur); ins.append(False); alt.append(False)
            cur += 1

    for pos, r, alts, gts in recs:
        a = gts[h] if h < len(gts) else 0
        if a == 0:
            continue
        al = alts[a - 1]
        # the allele against REF, reduced the way VCF normalisation does it: common trailing bases go while both
        # strings keep one base, then common leading bases (restated here; the readers under test have their own)
        while len(r) > 1 and len(al) > 1 and r[-1] == al[-1]:
            r, al = r[:-1], al[:-1]
        while len(r) > 1 and len(al) > 1 and r[0] == al[0]:
            r, al, pos = r[1:], al[1:], pos + 1
        if r == al:
            continue
        if len(r) == 1 and len(al) == 1:                         # substitution
            if pos < cur:
                raise ValueError(f"haplotype {h}: overlapping alleles at {pos}")
            copy_ref(pos)
            out.append(ord(al)); coord.append(pos); ins.append(False); alt.append(True)
            cur = pos + 1
            last_sub = pos
        elif len(r) == len(al):                                  # multi-base substitution
            if pos < cur:
                raise ValueError(f"haplotype {h}: overlapping alleles at {pos}")
            copy_ref(pos)
            for j, c in enumerate(al):
                out.append(ord(c)); coord.append(pos + j); ins.append(False); alt.append(c != r[j])
            cur = pos + len(r)
        elif al[0] == r[0] and (len(r) == 1 or len(al) == 1):    # insertion / deletion behind the anchor base
            if pos < cur and not (pos == cur - 1 and last_sub == pos):
                raise ValueError(f"haplotype {h}: overlapping alleles at {pos}")
            copy_ref(pos + 1)                                    # the anchor (already there if it was substituted)
            for c in al[1:]:
                out.append(ord(c)); coord.append(pos); ins.append(True); alt.append(True)
            if len(r) > 1:
                cur = pos + len(r)                               # the deleted bases are not copied
            last_sub = -1                                        # one indel per (possibly substituted) anchor
        else:
            # a complex allele (REF=ACG ALT=TC after trimming): ALT replaces REF; its first min(|REF|, |ALT|) bases are
            # aligned to the reference bases they replace, bases beyond that are inserted behind the last of them
            if pos < cur:
                raise ValueError(f"haplotype {h}: overlapping alleles at {pos}")
            copy_ref(pos)
            n = min(len(r), len(al))
            for j in range(n):
                out.append(ord(al[j])); coord.append(pos + j); ins.append(False); alt.append(al[j] != r[j])
            for c in al[n:]:
                out.append(ord(c)); coord.append(pos + n - 1); ins.append(True); alt.append(True)
            cur = pos + len(r)
            last_sub = -1
    copy_ref(len(ref))
    return out, coord, ins, alt


def consistent(ref: bytes, recs, H: int) -> bool:
    try:
        for h in range(H):
            haplotype_sequence(ref, recs, h)
    except ValueError:
        return False
    return True


def window_counts(ref: bytes, recs, H: int, S: int, E: int, W: int):
    """-> (freq, flags): freq[(kmer bytes, start, stop)] = number of (haplotype, window) pairs inside the region
    [S, E) (S <= start < E, stop <= E) that spell it; flags[key] = set of vg-style flags seen ('ref' iff no base of
    the window is substituted or inserted).  Forward strand; the '-' row of a walk is its mirror."""
    freq: Dict[Tuple[bytes, int, int], int] = defaultdict(int)
    flags: Dict[Tuple[bytes, int, int], set] = defaultdict(set)
    for h in range(H):
        seq, coord, ins, alt = haplotype_sequence(ref, recs, h)
        n = len(seq)
        nalt = [0]
        for a in alt:
            nalt.append(nalt[-1] + (1 if a else 0))
        for o in range(0, n - W + 1):
            start = coord[o] + (1 if ins[o] else 0)
            if start < S:
                continue
            if start >= E:            # (a k-mer made of nothing but the bases inserted behind the region's LAST base would start
                break                 # at E with an empty extent: not a window start of [S, E) -- the enumerator's rule, and the kernels')
            stop = coord[o + W - 1] + 1
            if stop > E:
                continue
            key = (bytes(seq[o:o + W]), start, stop)
            freq[key] += 1
            flags[key].add("ref" if nalt[o + W] == nalt[o] else "non.ref")
    return freq, flags


def revcomp(kmer: bytes) -> bytes:
    return kmer.translate(_COMP)[::-1]


def check_rows(rows, freq, flags=None):
    """rows: iterable of (kmer bytes, start, stop, strand '+'/'-', count[, flag]) as an enumerator emits them
    ('-' rows: reverse complement, start / stop swapped).  Asserts the statement in the module header; returns
    (keys carried by a haplotype, rows checked)."""
    agg = defaultdict(int)
    nrow = defaultdict(int)
    flag_of = {}
    n = 0
    for row in rows:
        kmer, start, stop, strand, count = row[:5]
        n += 1
        if strand == "-":
            kmer, start, stop = revcomp(kmer), stop, start
        key = (kmer, int(start), int(stop), strand)
        agg[key] += int(count)
        nrow[key] += 1
        if len(row) > 5:
            flag_of[key] = row[5]
    for strand in "+-":
        for (kmer, start, stop), c in freq.items():
            key = (kmer, start, stop, strand)
            assert key in agg, f"a haplotype carries {kmer.decode()} {start}-{stop} ({strand}) x{c}: no such row"
            assert agg[key] == c, f"{kmer.decode()} {start}-{stop} ({strand}): {agg[key]} haplotypes reported, {c} carry it"
            if flags is not None and key in flag_of and nrow[key] == 1:
                assert flags[(kmer, start, stop)] == {flag_of[key]}, (key, flags[(kmer, start, stop)], flag_of[key])
    for key, c in agg.items():
        if key[:3] not in freq:
            assert c == 0, f"{key[0].decode()} {key[1]}-{key[2]} ({key[3]}): {c} haplotypes reported, none carries it"
    return len(freq), n
