"""vg's OWN node tables pin the extraction's graph layout.  The reference repository holds three vg artefacts -- the
`.vg` its test_vg_construct compares (tests/grafimo_run_test.py:15-30; built by constructVG.py:296-338) and the
tutorial's x.xg / y.xg (constructVG.py:343-402 on xy.fa + xy2.vcf.gz: one-base insertions and deletions).
tests/golden/make_golden.py decoded them (oracle/vg_graph.py) into tests/golden/vg_graphs.json -- and the x.gbwt / y.gbwt beside
the xg files into the node ids every haplotype threads; here the GraphIndex
built from the same FASTA + VCF must reproduce vg's node ids, node sequences, edges and reference path for the WHOLE
chromosome -- SNP sites (alternates numbered before the reference allele), the 32-base chopping, the cuts around a
deleted stretch, and the node of an insertion numbered right behind the reference node that ends with its anchor."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, REF_DATA


@pytest.fixture(scope="module")
def vg_graphs():
    with open(os.path.join(GOLDEN, "vg_graphs.json")) as fh:
        return json.load(fh)


def _index(g):
    from grafimo_amd.extract_regions import GraphIndex
    return GraphIndex.from_fasta_vcf(os.path.join(REF_DATA, g["fasta"]), os.path.join(REF_DATA, g["vcf"]), g["chrom"])


@pytest.mark.parametrize("name", ["expected_vg", "tutorial_x_xg", "tutorial_y_xg"])
def test_node_table_equals_vgs_own(vg_graphs, name, capsys):
    g = vg_graphs[name]
    idx = _index(g)
    nodes, edges, ref_path = idx.graph_nodes()
    want = {int(k): v for k, v in g["nodes"].items()}
    assert nodes == want
    assert edges == sorted(tuple(e) for e in g["edges"])
    assert ref_path == g["ref_path"]
    assert "".join(nodes[n] for n in ref_path) == idx.ref.tobytes().decode()
    assert max(len(s) for s in nodes.values()) <= 32
    if name == "expected_vg":
        assert g["reversing_edges"] == 0 and len(nodes) == 15
        # vg names the alleles of a record _alt_<hash>_0 (reference) / _1 (alternate): the alternate has the LOWER id
        for k, v in g["alt_paths"].items():
            mate = g["alt_paths"][k[:-1] + ("1" if k.endswith("0") else "0")]
            assert (v[0] > mate[0]) == k.endswith("_0")
    else:
        assert len(nodes) == 69 and len(edges) == 87 and (idx.ins_len > 0).sum() == 5 and (idx.del_len > 0).sum() == 4


def test_node_paths_of_the_written_rows_use_vgs_ids(vg_graphs):
    """GraphIndex.node_path (column 7 of the TSV rows write_region_tsvs produces) walks nodes of vg's table in an order
    vg's edges allow, for every window of the tutorial chromosome -- windows through inserted bases and over deletions
    included -- and spells the row's k-mer."""
    g = vg_graphs["tutorial_x_xg"]
    idx = _index(g)
    nodes = {int(k): v for k, v in g["nodes"].items()}
    edges = {tuple(e) for e in g["edges"]}
    W = 12
    checked = through_ins = over_del = 0
    for p in list(range(0, 40)) + list(range(170, 200)) + list(range(318, 336)) + list(range(938, 960)):
        for bases in idx.window_walks(p, W, None):
            path = idx.nodes_of(bases)
            assert all((a, b) in edges for a, b in zip(path[:-1], path[1:])), (p, path)
            kmer = "".join(chr(idx.ins_bases[idx.ins_off[b[1]] + b[2]]) if b[0] == "ins" else
                           (chr(idx.ref[b[0]]) if b[1] == 0 else chr(idx.alt_bases[_site_at(idx, b[0]), b[1] - 1])) for b in bases)
            assert kmer in "".join(nodes[n] for n in path)
            through_ins += any(b[0] == "ins" for b in bases)
            plain = [b[0] for b in bases if b[0] != "ins"]
            over_del += any(y - x > 1 for x, y in zip(plain[:-1], plain[1:]))
            checked += 1
    assert checked > 150 and through_ins > 20 and over_del > 10, (checked, through_ins, over_del)


def _site_at(idx, x):
    i = int(np.searchsorted(idx.pos, x, side="left"))
    while idx.del_len[i] or idx.ins_len[i]:
        i += 1
    assert idx.pos[i] == x
    return i


@pytest.mark.parametrize("name", ["tutorial_x_xg", "tutorial_y_xg"])
def test_haplotypes_thread_the_nodes_vgs_gbwt_says(vg_graphs, name):
    """The GBWT beside the tutorial's xg (x.gbwt / y.gbwt: what `vg find -H` counts haplotypes with, extract_regions.py:
    180,225) holds, per haplotype, the nodes it threads.  The product's GraphIndex -- its haplotype bitsets per alternate
    allele, its node table -- must put every haplotype of xy2.vcf.gz on exactly those nodes: substitutions, the one-base
    insertions (their own node behind the anchor's) and the deletions (the deleted nodes left out) included."""
    g = vg_graphs[name]
    idx = _index(g)
    want = g["haplotype_paths"]
    assert len(want) == idx.n_haplotypes == g["gbwt_sequences"] // 2
    carries = lambda site, a, h: bool((int(idx.alt_bits[site, a, h // 64]) >> (h % 64)) & 1)       # noqa: E731
    for h in range(idx.n_haplotypes):
        bases, gone_until = [], -1
        site = 0
        for x in range(len(idx.ref)):
            here = []
            while site < len(idx.pos) and idx.pos[site] == x:
                here.append(site)
                site += 1
            if x <= gone_until:                         # inside a deletion this haplotype carries
                assert not any(carries(s_, a, h) for s_ in here for a in range(int(idx.n_alts[s_])))
                continue
            allele = 0
            for s_ in here:                             # the substitution site first, then insertions, then the deletion
                if idx.del_len[s_] == 0 and idx.ins_len[s_] == 0:
                    allele = next((a + 1 for a in range(int(idx.n_alts[s_])) if carries(s_, a, h)), 0)
            bases.append((x, allele))
            for s_ in here:
                if idx.ins_len[s_] > 0 and carries(s_, 0, h):
                    bases.extend(("ins", s_, t) for t in range(int(idx.ins_len[s_])))
            for s_ in here:
                if idx.del_len[s_] > 0 and carries(s_, 0, h):
                    gone_until = max(gone_until, x + int(idx.del_len[s_]))
        assert idx.nodes_of(bases) == want[h], (name, h)
    assert any(len(p) != len(want[0]) for p in want) or name == "tutorial_x_xg"      # (y's second haplotype carries a deletion)


def test_decoders_on_hand_made_streams(tmp_path):
    """oracle/vg_graph.py on a stream written here: varints over 127, two groups, a gzip container."""
    import gzip
    from oracle import vg_graph as vg

    def varint(v):
        out = b""
        while True:
            out += bytes([(v & 0x7F) | (0x80 if v > 0x7F else 0)])
            v >>= 7
            if not v:
                return out

    def field(no, payload):
        return varint(no << 3 | 2) + varint(len(payload)) + payload

    def node(nid, seq):
        return field(1, field(1, seq.encode()) + varint(3 << 3) + varint(nid))

    def edge(a, b):
        return field(2, varint(1 << 3) + varint(a) + varint(2 << 3) + varint(b))

    big = "ACGT" * 40                                      # a 160-byte node: two-byte length varints
    g1 = node(1, "CA") + node(300, big) + edge(1, 300)
    g2 = node(301, "T") + edge(300, 301)
    stream = b"".join(varint(2) + varint(2) + b"VG" + varint(len(m)) + m for m in (g1, g2))
    p = tmp_path / "t.vg"
    p.write_bytes(gzip.compress(stream[:20]) + gzip.compress(stream[20:]))     # BGZF = concatenated members
    got = vg.read_vg(str(p))
    assert got["nodes"] == {1: "CA", 300: big, 301: "T"} and [e[:2] for e in got["edges"]] == [(1, 300), (300, 301)]
    assert vg.reference_path(got["nodes"], [(1, 300), (300, 301)], "CA" + big + "T") == [1, 300, 301]
