"""Synthetic reference + phased SNP VCF for the k-mer extraction tests (shared by CPU and GPU tests)."""
import gzip
import os

import numpy as np

REF_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_data")


def make_graph_files(tmpdir, chrom="7", length=3000, n_sites=260, n_samples=65, seed=5, gz=True, rich=False):
    """FASTA + VCF with: clustered and isolated SNPs, multi-allelic sites, deletions, a few records with a symbolic
    ALT (to be skipped), unphased and missing genotypes.  `rich`: also insertions of 1..6 bases (alone or next to a
    substitution in one record), equal-length multi-base substitutions and second records at a position
    already used -- what only the round-2 reader / oracle (read_vcf_variants) take apart.  Returns (fasta, vcf)."""
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
        if rich:                                 # some positions get a second record
            pos = sorted(pos + [p for p in pos if rng.random() < 0.08])
        for p in pos:
            r = ref[p]
            others = [b for b in "ACGT" if b != r]
            rng.shuffle(others)
            kind = rng.random()
            if kind < 0.06:                      # deletion record
                alt, refa = r, r + "".join(ref[p + 1:p + 3])
            elif kind < 0.10 and not rich:       # a symbolic allele: the record belongs to no graph (skipped, counted)
                alt, refa = "<DEL>", r
            elif kind < 0.14 and rich:           # insertion of 1..6 bases behind the anchor
                alt, refa = r + "".join(rng.choice(list("ACGT"), size=int(rng.integers(1, 7)))), r
            elif kind < 0.18 and rich:           # multi-base substitution (at least one base differs)
                span = "".join(ref[p:p + 3])
                sub = "".join(rng.choice(list("ACGT"), size=len(span)))
                alt, refa = (sub if sub != span else others[0] + span[1:]), span
                if "N" in span or len(span) < 2:
                    alt, refa = others[0], r
            elif kind < 0.21 and rich:           # complex allele: a substitution and a deletion behind it
                alt, refa = others[0], r + "".join(ref[p + 1:p + 2])
                if len(refa) < 2:
                    alt, refa = others[0], r
            elif kind < 0.225 and rich:          # a base allele and a symbolic one: the whole record is left out
                alt, refa = others[0] + ",<CN0>", r
            elif kind < 0.25 and rich:           # a substitution and an insertion in one record
                alt, refa = others[0] + "," + r + "".join(rng.choice(list("ACGT"), size=int(rng.integers(1, 4)))), r
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


def scoring_fixture_graph():
    """The local graph behind the reference's scoring fixture (22:19723256-19723526), recovered from
    the fixture itself: reference bases from the `ref` rows, five SNPs (alt base + carrier count from
    the single-difference `non.ref` rows), one 2-bp deletion after 22:19723467 carried by one of the
    5096 haplotypes.  Carrier sets are disjoint (the one window that holds two SNPs reports 0
    haplotypes with both alternates).  Coordinates are shifted so that the region starts at 0."""
    from oracle import extract_oracle as xo
    with open(os.path.join(REF_DATA, "width_19", "scoring_test_input.tsv")) as fh:
        rows = [tuple(line.rstrip("\n").split("\t")) for line in fh]
    at = lambda s: int(s.split(":")[1][:-1])
    S, E, H = 19723256, 19723526, 5096
    refd, snps = {}, {}
    for r in rows:
        if r[2].endswith("+") and r[5] == "ref" and at(r[3]) - at(r[2]) == 19:
            for j, c in enumerate(r[1]):
                refd[at(r[2]) + j] = c
    for r in rows:
        if r[2].endswith("+") and r[5] == "non.ref" and at(r[3]) - at(r[2]) == 19:
            diffs = [(at(r[2]) + j, c) for j, c in enumerate(r[1]) if refd[at(r[2]) + j] != c]
            if len(diffs) == 1:
                snps[diffs[0]] = int(r[4])
    refseq = "".join(refd[x] for x in range(S, E)).encode()
    order = sorted(snps)
    hap = np.zeros((len(order), H), np.int8)
    nxt = 0
    for i, key in enumerate(order):
        hap[i, nxt:nxt + snps[key]] = 1
        nxt += snps[key]
    dhap = np.zeros((1, H), bool)
    dhap[0, nxt] = True
    sites = xo.Sites([p - S for p, _ in order], [refd[p] for p, _ in order], [[a] for _, a in order], hap)
    dels = xo.Dels([19723467 - S], [2], dhap)
    return rows, refseq, sites, dels, S, E


def make_consistent_graph_files(tmpdir, chrom="c", length=600, n_samples=24, seed=1, kinds="sidm", gz=False,
                                dense=True):
    """FASTA + phased VCF whose haplotypes are CONFLICT-FREE (what oracle/extract_bruteforce.py needs): records have
    disjoint REF spans, so a haplotype never carries two alleles over the same reference bases -- except that a
    single-base substitution record may be followed by an insertion / deletion record anchored on that same base, and a
    position may get a second substitution record carried only by haplotypes that are reference at the first.
    Only alleles the extraction graph models (kinds: s substitutions incl. multi-allelic, i insertions, d deletions,
    m equal-length multi-base substitutions, D one record whose alleles delete different stretches of its REF --
    several lengths behind one anchor and nested deletions with anchors further right, what an STR record normalises
    to --, O two records whose deletions overlap, the second carried only by haplotypes without the first, c complex
    alleles: REF and ALT of different lengths that share neither their first nor their last base, and S records with a
    symbolic ALT, which belong to no graph: `vg construct` without --handle-sv leaves them out whole).
    Returns (fasta, vcf)."""
    rng = np.random.default_rng(seed)
    ref = "".join(rng.choice(list("ACGT"), size=length, p=[0.3, 0.2, 0.2, 0.3]))
    fasta = os.path.join(tmpdir, f"cons{seed}.fa")
    with open(fasta, "w") as fh:
        fh.write(f">{chrom}\n")
        for i in range(0, length, 70):
            fh.write(ref[i:i + 70] + "\n")
    H = 2 * n_samples
    lines = []

    def genotypes(n_alleles, allowed=None):
        af = rng.random() ** 1.5
        g = np.where(rng.random(H) < af, rng.integers(1, n_alleles + 1, size=H), 0)
        if allowed is not None:
            g = np.where(allowed, g, 0)
        return g

    def emit(p, r, alts, g):
        cols = "\t".join(f"{g[2 * s]}|{g[2 * s + 1]}" for s in range(n_samples))
        lines.append(f"{chrom}\t{p + 1}\t.\t{r}\t{','.join(alts)}\t99\t.\t.\tGT\t{cols}")

    p = int(rng.integers(0, 4))
    while p < length - 12:
        r = ref[p]
        others = [b for b in "ACGT" if b != r]
        rng.shuffle(others)
        kind = rng.choice(list(kinds))
        end = p + 1                                   # first position behind this record's REF span
        if kind == "s":
            n_alt = int(rng.choice([1, 1, 1, 2, 3]))
            g = genotypes(n_alt)
            emit(p, r, others[:n_alt], g)
            u = rng.random()
            if u < 0.12 and n_alt < 3:                # a second record at the position, for the non-carriers
                g2 = genotypes(1, allowed=g == 0)
                emit(p, r, [others[n_alt] if rng.random() < 0.6 else others[0]], g2)
                g = np.where(g2 > 0, 1, g)
            elif u < 0.24 and "i" in kinds:           # an insertion anchored on the substituted base
                emit(p, r, [r + "".join(rng.choice(list("ACGT"), size=int(rng.integers(1, 5))))], genotypes(1))
            elif u < 0.36 and "d" in kinds:           # a deletion anchored on it
                ln = int(rng.integers(1, 5))
                emit(p, ref[p:p + 1 + ln], [r], genotypes(1))
                end = p + 1 + ln
        elif kind == "i":
            seqs = ["".join(rng.choice(list("ACGT"), size=int(rng.integers(1, 7)))) for _ in range(int(rng.choice([1, 1, 2])))]
            alts = [r + s for s in dict.fromkeys(seqs)]
            if rng.random() < 0.2:
                alts = [others[0]] + alts             # a substitution and insertions in one record
            emit(p, r, alts[:3], genotypes(len(alts[:3])))
        elif kind == "d":
            ln = int(rng.integers(1, 6))
            emit(p, ref[p:p + 1 + ln], [r], genotypes(1))
            end = p + 1 + ln
        elif kind == "D":
            ln = int(rng.integers(3, 7))                  # REF = anchor + ln bases
            refa = ref[p:p + 1 + ln]
            alts = []
            for j in rng.permutation(np.arange(1, ln + 1))[:3]:
                j = int(j)
                # remove j bases right behind the anchor (-> a deletion of j at p) or the last j (-> nested, anchor p + ln - j)
                alts.append(refa[0] + refa[1 + j:] if rng.random() < 0.5 else refa[:1 + ln - j])
            alts = [a for a in dict.fromkeys(alts) if a != refa]
            emit(p, refa, alts, genotypes(len(alts)))
            end = p + 1 + ln
        elif kind == "O":
            l1, shift, l2 = int(rng.integers(2, 6)), int(rng.integers(1, 3)), int(rng.integers(2, 6))
            g1 = genotypes(1)
            emit(p, ref[p:p + 1 + l1], [r], g1)
            p2 = p + min(shift, l1)                       # anchored on a base the first deletion removes
            emit(p2, ref[p2:p2 + 1 + l2], [ref[p2]], genotypes(1, allowed=g1 == 0))
            end = max(p + 1 + l1, p2 + 1 + l2)
        elif kind == "c":
            lr = int(rng.integers(1, 5))
            span = ref[p:p + lr]
            alts = []
            for _ in range(int(rng.choice([1, 1, 2]))):
                la = int(rng.choice([x for x in range(1, 7) if x != lr]))
                a = "".join(rng.choice(list("ACGT"), size=la))
                # neither end shared with REF: normalisation cannot reduce it to a substitution or a plain indel
                a = rng.choice([b for b in "ACGT" if b != span[0]]) + a[1:]
                if la > 1 and lr > 1:
                    a = a[:-1] + rng.choice([b for b in "ACGT" if b != span[-1]])
                alts.append(a)
            alts = list(dict.fromkeys(alts))
            emit(p, span, alts, genotypes(len(alts)))
            end = p + lr
        elif kind == "S":
            form = int(rng.integers(0, 4))
            if form == 0:
                emit(p, r, ["<DEL>"], genotypes(1))
            elif form == 1:
                emit(p, r, [others[0], "<CN0>"], genotypes(2))      # the base allele goes with the record
            elif form == 2:
                emit(p, ref[p:p + 3], [r, "*"], genotypes(2))
            else:
                emit(p, r, [r + "[" + chrom + ":5["], genotypes(1))
        else:
            ln = int(rng.integers(2, 5))
            span = ref[p:p + ln]
            sub = "".join(rng.choice([b for b in "ACGT" if b != c]) if rng.random() < 0.7 else c for c in span)
            if sub == span:
                sub = others[0] + span[1:]
            emit(p, span, [sub], genotypes(1))
            end = p + ln
        gap = int(rng.integers(0, 6)) if (dense and rng.random() < 0.7) else int(rng.integers(6, 40))
        if kind == "c":
            gap += 6              # up to five sites each: keeps the walks of a window enumerable in Python
        p = end + gap
    vcf = os.path.join(tmpdir, f"cons{seed}.vcf" + (".gz" if gz else ""))
    op = gzip.open if gz else open
    with op(vcf, "wt") as fh:
        fh.write("##fileformat=VCFv4.1\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" +
                 "\t".join(f"s{i}" for i in range(n_samples)) + "\n")
        fh.write("\n".join(lines) + "\n")
    return fasta, vcf


# ------------------------------------------------------------------------------------------------ the oracle's side
# (TEST INFRASTRUCTURE: everything below imports oracle/ and is used by the GPU tests as the EXPECTED side -- rows and
# tables that no HIP kernel produced.)
def variants_from_index(idx):
    """A GraphIndex's arrays as the oracle's Variants (oracle/extract_oracle.py) -- for hand-made graphs that never were
    a VCF.  Carriers come from the bitsets (none: no haplotypes)."""
    from oracle import extract_oracle as xo
    v = xo.Variants()
    H = int(idx.n_haplotypes) if idx.alt_bits is not None else 0
    v.n_haplotypes = H

    def carriers(i, k):
        if not H:
            return np.zeros(0, bool)
        words = np.ascontiguousarray(idx.alt_bits[i, k]).view(np.uint8)
        return np.unpackbits(words, bitorder="little")[:H].astype(bool)

    for i in range(len(idx.pos)):
        p = int(idx.pos[i])
        if idx.del_len[i] > 0:
            v.add(p, 2, length=int(idx.del_len[i]), carriers=[carriers(i, 0)])
        elif idx.ins_len[i] > 0:
            o = int(idx.ins_off[i])
            v.add(p, 1, seq=idx.ins_bases[o:o + int(idx.ins_len[i])].tobytes(), carriers=[carriers(i, 0)])
        else:
            na = int(idx.n_alts[i])
            v.add(p, 0, alts=[chr(int(c)) for c in idx.alt_bases[i, :na]], carriers=[carriers(i, k) for k in range(na)])
    return v


def motif_as_oracle_dict(motif):
    """the members oracle.compute_results reads, from a product Motif or a stand-in with the reference's members"""
    from oracle import oracle as orc
    W = int(motif.width)
    if hasattr(motif, "dense_score_matrix"):
        sm, bg = motif.dense_score_matrix(), motif.dense_bg()
    else:
        sm = np.ascontiguousarray(np.asarray(motif.score_matrix, dtype=np.int64).reshape(4, W))
        bg = np.array([float(motif.bg[n]) for n in "ACGT"], dtype=np.float64)
    return dict(score_matrix=sm, pmf=orc.comp_pval_mat(sm, bg), min_val=int(motif.min_val), scale=int(motif.scale),
                offset=float(motif.offset), width=W, motif_id=motif.motif_id, motif_name=motif.motif_name)


def oracle_table(tmpdir, chrom, ref, v, regions, motif, reuse_rows=False, **kw):
    """The report table the REFERENCE's pipeline gives for these regions, restated by the oracle end to end: the rows of
    `vg find -K W -E -H` from the walk enumerator (oracle/extract_oracle.py), written as the TSV files scan_graph leaves,
    read back, scored and filtered by oracle.compute_results (score_sequences.py:44-211, resultsTmp.py:241-314).
    kw: threshold, qval_t, no_qvalue, no_reverse, recomb.  `reuse_rows`: the files of the last call with these regions and
    this width are still in tmpdir (the enumeration is the slow part).  -> (DataFrame, rows scanned)
    The two sums of compute_score_seq (score_sequences.py:390-391) are taken the way GRAFIMO as shipped takes them -- numba's
    np.sum, a sequential loop (the oracle's sum_mode 1) -- not numpy's pairwise sums of the un-jitted function: the two differ
    in the last bit, which decides one thing only: at the LOWEST reachable score the sequential tail sum equals the total
    exactly (p = 1.0: not reported by `p < 1`, resultsTmp.py:303-307), the pairwise one gives 0.9999999999999999."""
    import pandas as pd
    from oracle import extract_oracle as xo
    from oracle import oracle as orc
    W = int(motif.width)
    d = os.path.join(str(tmpdir), "width_%d" % W)
    want = sorted(f"{chrom}_{int(s)}-{int(e)}.tsv" for s, e in regions)
    if not (reuse_rows and os.path.isdir(d) and sorted(os.listdir(d)) == want):
        os.makedirs(d, exist_ok=True)
        for f in os.listdir(d):
            os.remove(os.path.join(d, f))
        for s, e in regions:
            rows = xo.enumerate_region_variants(chrom, ref, v, int(s), int(e), W, with_counts=True)
            with open(os.path.join(d, f"{chrom}_{int(s)}-{int(e)}.tsv"), "w") as fh:
                for r in rows:
                    fh.write("\t".join(str(x) for x in r) + "\t1+,\n")
    res = orc.compute_results(motif_as_oracle_dict(motif), str(tmpdir), threshold=kw.get("threshold", 1.0),
                              qval_t=kw.get("qval_t", False), no_qvalue=kw.get("no_qvalue", False),
                              no_reverse=kw.get("no_reverse", False), recomb=kw.get("recomb", False), sum_mode=1)
    return pd.DataFrame({c: res[c] for c in res if not c.startswith("_")}), int(res["_scanned"])


def assert_table_equals_oracle(df, exp, what=""):
    """every column of every reported row; rows compared as a set keyed on everything that identifies one (pandas' sort
    on p-value leaves ties in no particular order, resultsTmp.py:312).  The score column -- scaled / scale + W * offset, the same
    two operations on both sides -- must be EQUAL.  p- and q-values at rtol 1e-12 (north_star asks 1e-6), not equal: the oracle
    takes the reference's two O(1000 W) sums per row (score_sequences.py:390-391), sequentially as numba does; the device
    looks the same quotient up in a tail table made by a blocked suffix sum -- another association order of the same
    additions, a few ulp apart for motifs whose pmf is not as benign as CTCF's (for which the TSV path's tests do find them
    equal, against the golden vectors)."""
    assert list(df.columns) == list(exp.columns), (what, list(df.columns), list(exp.columns))
    assert len(df) == len(exp), (what, len(df), len(exp))
    key = ["p-value", "sequence_name", "start", "stop", "strand", "matched_sequence", "haplotype_frequency"]
    a = df.sort_values(key).reset_index(drop=True)
    b = exp.sort_values(key).reset_index(drop=True)
    for c in exp.columns:
        if c == "score":
            assert np.array_equal(a[c].to_numpy(float), b[c].to_numpy(float)), (what, c)
        elif b[c].dtype.kind == "f":
            np.testing.assert_allclose(a[c].to_numpy(float), b[c].to_numpy(float), rtol=1e-12, atol=0, err_msg=str((what, c)))
        else:
            assert (a[c].astype(str) == b[c].astype(str)).all(), (what, c)


def write_region_tsvs_reference(index, rows, out_dir, labels=None, chrom=None):
    """TEST INFRASTRUCTURE: the Python row loop that wrote scan_graph's files in rounds 1-4 (4 us a row), kept as the
    expected side of the native writer (gfm_graph_write_tsvs, csrc/graph_tsv_writer.cpp): the two must agree byte for byte.
    out_dir/width_W/CHR_S-E.tsv (extract_regions.py:165-170,180), seven tab-separated columns per row like vg's; node
    paths through GraphIndex's node table and its Python walk enumerator (window_walks)."""
    from grafimo_amd.extract_regions import NODE_MAX
    W = rows.width
    cname = index.chrom if chrom is None else chrom
    d = os.path.join(out_dir, f"width_{W}")
    os.makedirs(d, exist_ok=True)
    km = rows.kmers.cpu().numpy()
    start, stop = rows.start.cpu().numpy(), rows.stop.cpu().numpy()
    strand, freq = rows.strand.cpu().numpy(), rows.freq.cpu().numpy()
    is_ref, region, walk = rows.is_ref.cpu().numpy(), rows.region.cpu().numpy(), rows.walk.cpu().numpy()
    paths = []
    bounds = np.searchsorted(region, np.arange(len(rows.regions) + 1), side="left")
    cuts, first, site_of, _ = index._node_table()
    L = len(index.ref)
    for r in range(len(rows.regions)):
        label = rows.region_label(r) if labels is None else labels[r]
        path = os.path.join(d, label.replace(":", "_") + ".tsv")
        S, E = int(rows.regions[r][0]), int(rows.regions[r][1])
        # node of every reference base of the region, reference alleles at the SNP sites: ONE search for the region
        # (per base and walk it was 17 us per row: two and a half minutes for the rows of ten thousand regions)
        x0, x1 = max(S, 0), min(max(E, S) + W + 1, L)
        xs = np.arange(x0, max(x1, x0), dtype=np.int64)
        jj = np.searchsorted(cuts, xs, side="right") - 1
        st_ = site_of[jj]
        node_ref = np.where(st_ >= 0, first[jj] + index.n_alts[np.maximum(st_, 0)], first[jj] + (xs - cuts[jj]) // NODE_MAX)
        lines = []
        cur_p, node_paths, plain = None, [], False
        i0 = i1 = 0
        w_i0, w_i1, w_touch = index.window_table(x0, max(x0, min(E, L)), W)
        for i in range(bounds[r], bounds[r + 1]):
            sg = chr(strand[i])
            p = int(start[i]) if sg == "+" else int(stop[i])
            if p != cur_p:                        # rows are window-major: a window's walks are prepared once
                cur_p = p
                plain = 0 <= p - x0 < len(w_touch) and not w_touch[p - x0]
                if plain:
                    i0, i1 = int(w_i0[p - x0]), int(w_i1[p - x0])
                    node_paths = {}
                else:
                    node_paths = [index.nodes_of(b_) for b_ in index.window_walks(p, W, rows.regions[r][1])]
            q = int(walk[i])
            if plain:
                nodes = node_paths.get(q)
                if nodes is None:                 # the reference path with the walk's alternate nodes put in
                    nd = node_ref[p - x0:p - x0 + W].copy()
                    qq = q
                    for k in range(i1 - i0 - 1, -1, -1):
                        n_all = 1 + int(index.n_alts[i0 + k])
                        a_ = qq % n_all
                        qq //= n_all
                        if a_:
                            xk = int(index.pos[i0 + k])
                            nd[xk - p] = first[int(np.searchsorted(cuts, xk, side="right")) - 1] + a_ - 1
                    nodes = nd[np.concatenate(([True], nd[1:] != nd[:-1]))].tolist()
                    node_paths[q] = nodes
            else:
                nodes = node_paths[q]
            if sg == "-":
                nodes = nodes[::-1]
            lines.append(f"{label}\t{km[i].tobytes().decode()}\t{cname}:{int(start[i])}{sg}\t{cname}:{int(stop[i])}{sg}\t"
                         f"{int(freq[i])}\t{'ref' if is_ref[i] else 'non.ref'}\t" + "".join(f"{n}{sg}," for n in nodes) + "\n")
        with open(path, "w") as fh:
            fh.writelines(lines)
        paths.append(path)
    return paths


def snp_graph_score_histogram(idx, regions, W, sm, L, min_val, forward_only=False):
    """TEST INFRASTRUCTURE (no HIP, no per-row Python): the score histogram of ALL rows `vg find -K W -E` would print for the
    regions of a graph of substitution sites only -- every window start p with p + W inside the region and the reference, every
    combination of the alleles of the sites in [p, p + W), both strands -- counted with numpy: the reference window's score by a
    sliding window over the region's bases, the walks by adding per-site score differences combination by combination
    (score_sequences.py:372-396: score = sum over j of sm[code(kmer[j]), j]; the reverse strand holds comp(base j) at W-1-j;
    an invalid base makes the whole k-mer score min_val).  Pinned against the walk enumerator + the C oracle on small graphs
    (tests/test_extract_host.py); used at BASELINE configs[1] scale where the enumerator cannot go (tests/test_gpu_fused.py).
    -> (hist int64 [L], rows)"""
    import itertools
    assert not (np.asarray(idx.del_len) > 0).any() and not (np.asarray(idx.ins_len) > 0).any()
    sm = np.asarray(sm, dtype=np.int64).reshape(4, W)
    code_of = np.full(256, -1, dtype=np.int64)
    for i, c in enumerate(b"ACGT"):
        code_of[c] = i
    ref = code_of[np.asarray(idx.ref, dtype=np.uint8)]
    pos = np.asarray(idx.pos, dtype=np.int64)
    n_alts = np.asarray(idx.n_alts, dtype=np.int64)
    alt = code_of[np.asarray(idx.alt_bases, dtype=np.uint8)]          # [V, 3]
    hist = np.zeros(L, dtype=np.int64)
    rows = 0
    ar = np.arange(W)
    # every window start of every region, in chunks: all regions at once (a Python step per REGION was 30 ms of numpy calls)
    lo = np.maximum(np.asarray([r[0] for r in regions], dtype=np.int64), 0)
    hi = np.minimum(np.asarray([r[1] for r in regions], dtype=np.int64), len(ref))
    n_win = np.maximum(hi - W - lo + 1, 0)
    starts = np.repeat(lo, n_win) + (np.arange(int(n_win.sum())) - np.repeat(np.cumsum(n_win) - n_win, n_win))
    for c0 in range(0, len(starts), 1 << 18):
        p = starts[c0:c0 + (1 << 18)]
        win = ref[p[:, None] + ar[None, :]]                                         # [n, W] codes
        bad0 = (win < 0).sum(1)
        safe = np.where(win < 0, 0, win)
        base_f = sm[safe, ar].sum(1)
        base_r = sm[3 - safe, W - 1 - ar].sum(1)
        i0 = np.searchsorted(pos, p, side="left")
        i1 = np.searchsorted(pos, p + W, side="left")
        k_of = i1 - i0
        for k in np.unique(k_of).tolist():
            w = np.flatnonzero(k_of == k)
            if k == 0:
                f, r, bad = base_f[w], base_r[w], bad0[w]
                s_f = np.where(bad > 0, min_val, f)
                s_r = np.where(bad > 0, min_val, r)
                hist += np.bincount(s_f, minlength=L)
                rows += len(w)
                if not forward_only:
                    hist += np.bincount(s_r, minlength=L)
                    rows += len(w)
                continue
            site = i0[w][:, None] + np.arange(k)[None, :]                           # [n, k] site indices
            j = pos[site] - p[w][:, None]                                           # position in the window
            cr = ref[pos[site]]
            na = n_alts[site]
            for combo in itertools.product(range(4), repeat=k):
                c = np.asarray(combo)
                ok = (c[None, :] <= na).all(1)
                if not ok.any():
                    continue
                f, r, bad = base_f[w][ok].copy(), base_r[w][ok].copy(), bad0[w][ok].copy()
                for t in range(k):
                    if c[t] == 0:
                        continue
                    ca = alt[site[ok, t], c[t] - 1]
                    jj, c0_ = j[ok, t], cr[ok, t]
                    c0s, cas = np.where(c0_ < 0, 0, c0_), np.where(ca < 0, 0, ca)
                    f += sm[cas, jj] - sm[c0s, jj]
                    r += sm[3 - cas, W - 1 - jj] - sm[3 - c0s, W - 1 - jj]
                    bad += (ca < 0).astype(np.int64) - (c0_ < 0).astype(np.int64)
                s_f = np.where(bad > 0, min_val, f)
                s_r = np.where(bad > 0, min_val, r)
                hist += np.bincount(s_f, minlength=L)
                rows += len(s_f)
                if not forward_only:
                    hist += np.bincount(s_r, minlength=L)
                    rows += len(s_r)
    return hist, rows
