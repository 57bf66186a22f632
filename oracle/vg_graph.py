"""TEST INFRASTRUCTURE (only tests/ and tests/golden/make_golden.py import this; the product path never does).

Readers for the two kinds of vg artefact the REFERENCE REPOSITORY holds, so that vg's own node table can pin the
extraction's node numbering although no `vg` binary exists in this image:

  * read_vg   -- a `.vg` file: what `vg construct -C -R chrom -r REF -v VCF` writes (constructVG.py:296-338) and what
                 the reference's test_vg_construct compares by size (tests/grafimo_run_test.py:15-30;
                 tests/test_data/expected_results/expected.vg).  Container: BGZF (concatenated gzip members) around a
                 type-tagged stream of groups `[varint count][varint len, bytes]*`, the first message of a group the tag
                 "VG", the others protobuf `Graph` messages (vg.proto: Graph{1: Node, 2: Edge, 3: Path};
                 Node{1: sequence, 2: name, 3: id}; Edge{1: from, 2: to, 3: from_start, 4: to_end};
                 Path{1: name, 2: Mapping}; Mapping{1: Position{1: node_id, 2: offset, 4: is_reverse}, 2: Edit, 5: rank}).
  * read_xg   -- an `.xg` index: what `vg index -x` writes (constructVG.py:343-402) and what the tutorial ships
                 (tutorials/findmotif_tutorial/data/mygenome/{x,y}.xg, built from tutorials/buildvg_tutorial/data/xy.fa +
                 xy2.vcf.gz -- 38 records with one-base insertions and deletions).  Same tagged container (tag "XG",
                 uncompressed), payload = xg's sdsl serialisation: magic 0xF6F596A1 and a version (both big-endian
                 uint32), six uint64 (sequence length, node / edge / path count, min / max id), then sdsl int_vectors
                 (uint64 bit count, uint8 width, 64-bit words): the id vector, then the GRAPH VECTOR -- per node the
                 record [id, sequence start, length, #edges to, #edges from] followed by one entry per edge side
                 (zigzag of the other node's record offset relative to this one, << 1 | reversing bit) -- and, further on, the SEQUENCE
                 VECTOR (codes A 0, T 1, C 2, G 3, N 4).  Decoded: node ids, node sequences, edges.  NOT decoded (the
                 decode stops there): the rank / select supports, the path structures (XGPath: sdsl enc_vectors and
                 wavelet trees) -- so the reference path is recovered by matching node sequences against the FASTA.
  * read_gbwt -- the GBWT haplotype index beside an xg (x.gbwt; `vg index -G`, constructVG.py:343-402; what `vg find -H`
                 counts haplotypes with, extract_regions.py:180,225).  Tagged container (tag "GBWT"), payload = gbwt's
                 sdsl serialisation, version 4: header (tag 0x6B376B37, version, sequences, size, offset, alphabet
                 size, flags), the record array (record count; an sdsl sd_vector of the records' start bytes with its two
                 select supports; then the bytes), further on the document-array samples and the metadata (not read).
                 A record belongs to one oriented node (GBWT node = 2 * id + is_reverse; record r = node r + offset, record 0
                 the endmarker): ByteCode outdegree, (node delta, offset) per edge, then the run-length coded body: per
                 visit of the node, in order, the rank of the edge the visiting sequence leaves by.  Sequences are followed
                 from the endmarker by LF-mapping; sequence 2k is haplotype k forward, 2k + 1 its reverse.
"""
import gzip
import struct
from typing import Dict, List, Tuple


def _varint(b: bytes, i: int) -> Tuple[int, int]:
    v = s = 0
    while True:
        c = b[i]
        i += 1
        v |= (c & 0x7F) << s
        s += 7
        if not c & 0x80:
            return v, i


def _fields(b: bytes):
    """(field number, wire type, value) of one protobuf message; length-delimited values as bytes."""
    i, out = 0, []
    while i < len(b):
        key, i = _varint(b, i)
        f, wt = key >> 3, key & 7
        if wt == 0:
            v, i = _varint(b, i)
        elif wt == 2:
            ln, i = _varint(b, i)
            v = b[i:i + ln]
            i += ln
        elif wt == 1:
            v = b[i:i + 8]
            i += 8
        elif wt == 5:
            v = b[i:i + 4]
            i += 4
        else:
            raise ValueError(f"wire type {wt}")
        out.append((f, wt, v))
    return out


def _groups(data: bytes) -> List[List[bytes]]:
    i, out = 0, []
    while i < len(data):
        cnt, i = _varint(data, i)
        msgs = []
        for _ in range(cnt):
            ln, i = _varint(data, i)
            msgs.append(data[i:i + ln])
            i += ln
        out.append(msgs)
    return out


def read_vg(path: str) -> dict:
    """-> dict(nodes={id: sequence}, edges=[(from, to, from_start, to_end)], paths={name: [node ids in rank order]})"""
    raw = open(path, "rb").read()
    data = gzip.decompress(raw) if raw[:2] == b"\x1f\x8b" else raw
    nodes: Dict[int, str] = {}
    edges: List[Tuple[int, int, int, int]] = []
    paths: Dict[str, List[Tuple[int, int]]] = {}
    for grp in _groups(data):
        if not grp or grp[0] != b"VG":
            continue
        for msg in grp[1:]:
            for f, _, v in _fields(msg):
                if f == 1:
                    d = {ff: vv for ff, _, vv in _fields(v)}
                    nodes[int(d[3])] = d.get(1, b"").decode()
                elif f == 2:
                    d = {ff: vv for ff, _, vv in _fields(v)}
                    edges.append((int(d[1]), int(d[2]), int(d.get(3, 0)), int(d.get(4, 0))))
                elif f == 3:
                    name, steps = None, []
                    for ff, _, vv in _fields(v):
                        if ff == 1:
                            name = vv.decode()
                        elif ff == 2:
                            pos, rank = {}, 0
                            for f3, _, v3 in _fields(vv):
                                if f3 == 1:
                                    pos = {a: b for a, _, b in _fields(v3)}
                                elif f3 == 5:
                                    rank = int(v3)
                            steps.append((rank, int(pos.get(1, 0))))
                    paths.setdefault(name, []).extend(steps)
    return dict(nodes=nodes, edges=sorted(edges),
                paths={k: [n for _, n in sorted(v)] for k, v in paths.items()})


def _int_vector(raw: bytes, off: int):
    bits = struct.unpack_from("<Q", raw, off)[0]
    width = raw[off + 8]
    n_words = (bits + 63) // 64
    big = int.from_bytes(raw[off + 9:off + 9 + 8 * n_words], "little")
    n = bits // width if width else 0
    mask = (1 << width) - 1
    return [(big >> (i * width)) & mask for i in range(n)], off + 9 + 8 * n_words, width


XG_MAGIC = 0xF6F596A1
_XG_BASES = "ATCGN"


def read_xg(path: str) -> dict:
    """-> dict(version, seq_length, nodes={id: sequence}, edges=[(from, to)] (all forward: end of `from` to start of
    `to`), node_order=[ids in the index's order])."""
    raw = open(path, "rb").read()
    grp = _groups(raw)
    if not grp or grp[0][0] != b"XG":
        raise ValueError(f"{path}: not a tagged XG file")
    p = b"".join(grp[0][1:])
    magic, version = struct.unpack_from(">II", p, 0)
    if magic != XG_MAGIC:
        raise ValueError(f"{path}: bad XG magic {magic:#x}")
    seq_length, node_count, edge_count, path_count, min_id, max_id = struct.unpack_from("<6Q", p, 8)
    ids, off, _ = _int_vector(p, 56)
    if len(ids) != node_count:
        raise ValueError(f"{path}: id vector holds {len(ids)} entries for {node_count} nodes")
    g, off, _ = _int_vector(p, off)
    # the sequence vector: the first int_vector behind the graph vector that holds seq_length entries
    seq = None
    for at in range(off, len(p) - 9):
        bits = struct.unpack_from("<Q", p, at)[0]
        w = p[at + 8]
        if w in (2, 3) and bits == seq_length * w and at + 9 + 8 * ((bits + 63) // 64) <= len(p):
            seq, _, _ = _int_vector(p, at)
            break
    if seq is None:
        raise ValueError(f"{path}: sequence vector not found")
    bases = "".join(_XG_BASES[c] for c in seq)
    nodes, order, rec_at, recs = {}, [], {}, []
    i = 0
    while i < len(g):
        nid, start, length, n_to, n_from = g[i:i + 5]
        rec_at[i] = nid
        recs.append((i, nid, g[i + 5:i + 5 + n_to], g[i + 5 + n_to:i + 5 + n_to + n_from]))
        nodes[nid] = bases[start:start + length]
        order.append(nid)
        i += 5 + n_to + n_from
    if len(nodes) != node_count:
        raise ValueError(f"{path}: graph vector holds {len(nodes)} nodes, header says {node_count}")
    edges = set()

    def other(at, v):
        # an edge entry = zigzag(offset of the other node's record relative to this one) << 1 | reversing bit:
        # +7 -> 28, -7 -> 26 (read off the tutorial's files: node 1 -> nodes 2, 3 at +7, +14; node 2 <- node 1 at -7)
        if v & 1:
            raise ValueError(f"{path}: reversing edge (not expected from vg construct)")
        z = v >> 1
        return rec_at[at + ((z >> 1) if z % 2 == 0 else -((z + 1) >> 1))]

    for at, nid, to_side, from_side in recs:
        for v in from_side:          # the nodes this one leads to
            edges.add((nid, other(at, v)))
        for v in to_side:            # the nodes that lead here
            edges.add((other(at, v), nid))
    if len(edges) != edge_count:
        raise ValueError(f"{path}: decoded {len(edges)} edges, header says {edge_count}")
    return dict(version=int(version), seq_length=int(seq_length), path_count=int(path_count), min_id=int(min_id),
                max_id=int(max_id), nodes=nodes, edges=sorted(edges), node_order=order)


def reference_path(nodes: Dict[int, str], edges, ref: str) -> List[int]:
    """The node ids that spell `ref` from the first node on, following edges: at a branch the successor whose
    sequence continues the reference (ties -- an alternate allele equal to the reference cannot occur -- the LOWER
    id is not preferred: vg numbers alternates first, so among equal sequences there is none)."""
    succ: Dict[int, List[int]] = {}
    for e in edges:
        succ.setdefault(e[0], []).append(e[1])
    starts = [n for n in nodes if ref.startswith(nodes[n]) and not any(e[1] == n for e in edges)]
    if len(starts) != 1:
        raise ValueError(f"reference_path: {len(starts)} possible first nodes")
    path, at = [starts[0]], len(nodes[starts[0]])
    while at < len(ref):
        cands = [n for n in succ.get(path[-1], []) if ref.startswith(nodes[n], at)]
        # a deletion edge and the plain edge may both continue the reference only if the deleted stretch repeats;
        # prefer the successor that keeps the whole reference reachable: the one with the smallest id that is not an
        # alternate of a site (an alternate never equals the reference base there)
        if not cands:
            raise ValueError(f"reference_path: stuck at base {at} behind node {path[-1]}")
        best = None
        for n in sorted(cands):
            if _spells(nodes, succ, n, ref, at, 64):
                best = n
                break
        if best is None:
            raise ValueError(f"reference_path: no successor of node {path[-1]} spells the reference at base {at}")
        path.append(best)
        at += len(nodes[best])
    return path


def _spells(nodes, succ, n, ref, at, depth) -> bool:
    """does some walk from node n spell ref[at:] for at least `depth` more nodes (or to the end)?"""
    if not ref.startswith(nodes[n], at):
        return False
    at += len(nodes[n])
    if at >= len(ref) or depth == 0:
        return True
    return any(_spells(nodes, succ, m, ref, at, depth - 1) for m in succ.get(n, []))


def _sdsl_int_vector0(b: bytes, pos: int):
    """sdsl int_vector<0>: uint64 bit count, uint8 width, 64-bit words -> (values, next position)"""
    bits, = struct.unpack_from("<Q", b, pos)
    width = b[pos + 8]
    n_bytes = (bits + 63) // 64 * 8
    big = int.from_bytes(b[pos + 9:pos + 9 + n_bytes], "little")
    vals = [(big >> (i * width)) & ((1 << width) - 1) for i in range(bits // width)] if width else []
    return vals, pos + 9 + n_bytes


def _sdsl_bit_vector(b: bytes, pos: int):
    bits, = struct.unpack_from("<Q", b, pos)
    n_bytes = (bits + 63) // 64 * 8
    return bits, int.from_bytes(b[pos + 8:pos + 8 + n_bytes], "little"), pos + 8 + n_bytes


def _sdsl_select_mcl(b: bytes, pos: int) -> int:
    """skips an sdsl select_support_mcl: count, superblock vector, mini_or_long bits, one block vector per 4096 arguments"""
    arg_cnt, = struct.unpack_from("<Q", b, pos)
    pos += 8
    if arg_cnt:
        _, pos = _sdsl_int_vector0(b, pos)
        _, _, pos = _sdsl_bit_vector(b, pos)
        for _ in range((arg_cnt + 4095) >> 12):
            _, pos = _sdsl_int_vector0(b, pos)
    return pos


def _bytecode(buf: bytes, i: int) -> Tuple[int, int]:
    v = shift = 0
    while True:
        c = buf[i]
        i += 1
        v |= (c & 0x7F) << shift
        shift += 7
        if not c & 0x80:
            return v, i


def read_gbwt(path: str) -> dict:
    """-> dict(version, sequences, offset, alphabet_size, bidirectional, paths): paths[j] = the oriented nodes
    [(node id, is_reverse), ...] sequence j visits, in order."""
    raw = open(path, "rb").read()
    groups = _groups(raw)
    if not groups or groups[0][0] != b"GBWT":
        raise ValueError(f"{path}: not a type-tagged GBWT stream")
    b = b"".join(groups[0][1:])
    tag, version = struct.unpack_from("<II", b, 0)
    if tag != 0x6B376B37:
        raise ValueError(f"{path}: GBWT header tag {tag:#x}")
    sequences, size, offset, alphabet_size, flags = struct.unpack_from("<5Q", b, 8)
    if version != 4:
        raise ValueError(f"{path}: GBWT version {version}: only version 4 is decoded")
    pos = 48
    records, = struct.unpack_from("<Q", b, pos)
    pos += 8
    data_len, = struct.unpack_from("<Q", b, pos)                      # sd_vector: size, wl, low, high, two select supports
    wl = b[pos + 8]
    low, pos = _sdsl_int_vector0(b, pos + 9)
    hbits, high, pos = _sdsl_bit_vector(b, pos)
    pos = _sdsl_select_mcl(b, pos)
    pos = _sdsl_select_mcl(b, pos)
    data = b[pos:pos + data_len]
    starts, k = [], 0
    for p in range(hbits):
        if (high >> p) & 1:
            starts.append(((p - k) << wl) | low[k])
            k += 1
    if len(starts) != records or len(data) != data_len:
        raise ValueError(f"{path}: record index does not add up")
    recs = []
    for r in range(records):
        buf = data[starts[r]:starts[r + 1] if r + 1 < records else data_len]
        sigma, i = _bytecode(buf, 0)
        edges, node = [], 0
        for _ in range(sigma):
            d, i = _bytecode(buf, i)
            o, i = _bytecode(buf, i)
            node += d
            edges.append((node, o))
        body = []                                                     # rank of the outgoing edge per visit
        while i < len(buf):
            if sigma >= 255:
                rank, i = _bytecode(buf, i)
                extra, i = _bytecode(buf, i)
                run = extra + 1
            else:
                per_byte = 256 // sigma                               # run lengths one byte can hold
                c = buf[i]
                i += 1
                rank, run = c % sigma, c // sigma + 1
                if run == per_byte:                                   # the longest: more follows
                    extra, i = _bytecode(buf, i)
                    run += extra
            body.extend([rank] * run)
        recs.append((edges, body))

    def record_of(node):
        return 0 if node == 0 else node - offset

    paths = []
    for j in range(sequences):
        node, at, path = 0, j, []
        while True:
            edges, body = recs[record_of(node)]
            rank = body[at]
            nxt, first = edges[rank]
            at = first + sum(1 for x in body[:at] if x == rank)
            node = nxt
            if node == 0:
                break
            path.append((node >> 1, node & 1))
            if len(path) > size:
                raise ValueError(f"{path}: a sequence does not end")
        paths.append(path)
    return dict(version=version, sequences=sequences, offset=offset, alphabet_size=alphabet_size,
                bidirectional=bool(flags & 1), paths=paths)
