"""vg's own index files as the graph input of scan_graph: the XG (`vg index -x`: nodes, edges, the reference path) and the GBWT
beside it (`vg index -G`: the haplotype threads) -- what a GRAFIMO user HAS, and what the reference hands to
`vg find -p REGION -x XG -H GBWT -K W -E` (extract_regions.py:172-180, 217-225; built by constructVG.py:343-402).  Read here on
the host, once per chromosome, and turned into the GraphIndex the extraction kernels work on (index_from_vg); scan_graph does
that by itself when it finds `chrN.xg` + `chrN.gbwt` and no `chrN.gfmidx.npz`, and saves the index for the next run.

What is decoded, and what it is pinned by.  The formats are vg's (sdsl-lite serialisations inside vg's type-tagged message
stream); no vg binary and no vg source exist in this image, so the layouts below were read off the files the reference
repository ships (tutorials/findmotif_tutorial/data/mygenome/{x,y}.xg + .gbwt: XG version 15, GBWT version 4, built from
tutorials/buildvg_tutorial/data/xy.fa + xy2.vcf.gz) and every decode is CHECKED rather than trusted: a path must be a walk
along the graph's edges from record start to record start, the edge and node counts must equal the header's, the GBWT's
record index must add up.  tests/test_vg_files.py holds the pin: the GraphIndex from x.xg + x.gbwt equals, array by array,
the one from xy.fa + xy2.vcf.gz (the route pinned against vg's own `vg find` rows), for both tutorial chromosomes.  Other
versions of the two formats are refused by name, not guessed at.

  XG (version 15), payload after the tag "XG":
    u32 magic 0xF6F596A1, u32 version (both big-endian); u64 x 6: sequence length, nodes, edges, paths, min id, max id;
    r_iv   int_vector<0>   node ids by rank
    g_iv   int_vector<0>   per node the record [id, start in s_iv, length, #edges to, #edges from] and one entry per edge
                           side: zigzag(offset of the other node's record relative to this one) << 1 | reversing bit
    g_bv   bit_vector      1 at every record start of g_iv;  + rank_support_v, select_support_mcl
    s_iv   int_vector<0>   the bases (A 0, T 1, C 2, G 3, N 4);  s_bv bit_vector (node starts) + rank + select
    pn_iv  int_vector<0>   the path names, "#name$" each;  pn_csa (a compressed suffix array: NOT parsed -- skipped by
                           looking for the first byte offset behind it at which the path block below decodes and validates)
    u64 path count, then per path:  u64 min_handle;  enc_vector<elias_delta, 128> handles (handle = record offset << 1 |
                           is_reverse, stored minus min_handle);  rrr_vector<63> offsets (skipped structurally);  u8 is_circular
  GBWT (version 4), payload after the tag "GBWT":
    u32 tag 0x6B376B37, u32 version, u64 x 5: sequences, size, offset, alphabet size, flags (bit 0 = bidirectional);
    u64 records;  sd_vector of the records' first bytes (u64 size, u8 low width, low int_vector<0>, high bit_vector, two
    select supports);  the records' bytes.  Record r belongs to GBWT node r + offset (record 0: the endmarker), GBWT node =
    2 * id + is_reverse: ByteCode outdegree, (node delta, offset) per outgoing edge, then runs (edge rank, length) -- one
    entry per visit, in the order of the BWT.  In a bidirectional index sequence 2k is haplotype k forward.

Haplotypes per node WITHOUT following the threads one by one: the visits of a node are sorted by the node they came from, so
the sequence ids at node w are the concatenation, over its predecessors in node order, of the ids that left the predecessor
by the edge to w -- and the record says where each predecessor's block starts (the edge's offset).  One pass over the forward
nodes in id order (`vg construct` numbers nodes along the reference: every edge goes up) moves every id once per node it
visits, as array slices.
"""
import os
import struct
import sys
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

XG_MAGIC = 0xF6F596A1
XG_VERSIONS = (15,)                 # what the reference repository's files pin
GBWT_TAG = 0x6B376B37
GBWT_VERSIONS = (4,)
_XG_BASES = np.frombuffer(b"ATCGNNNN", dtype=np.uint8)
ENC_DENS = 128                      # sdsl::enc_vector<>'s sample density


class VGFormatError(ValueError):
    """the file is not what this reader decodes (wrong container, another version, a structure that does not add up)"""


def _varint(b, i: int) -> Tuple[int, int]:
    v = s = 0
    while True:
        c = b[i]
        i += 1
        v |= (c & 0x7F) << s
        s += 7
        if not c & 0x80:
            return v, i


def tagged_payload(raw: bytes, tag: bytes, path: str = "") -> bytes:
    """vg's type-tagged stream: groups `[varint count][varint length, bytes] * count`, the first message of a group its
    tag, the others the payload in order.  -> the payload of all groups with that tag, joined."""
    out, i, n = [], 0, len(raw)
    mv = memoryview(raw)
    try:
        while i < n:
            cnt, i = _varint(raw, i)
            for k in range(cnt):
                ln, i = _varint(raw, i)
                if i + ln > n:
                    raise IndexError
                if k == 0:
                    if bytes(mv[i:i + ln]) != tag:
                        raise VGFormatError(f"{path}: a group tagged {bytes(mv[i:i + ln])[:16]!r}, not {tag!r}")
                else:
                    out.append(mv[i:i + ln])
                i += ln
    except IndexError:
        raise VGFormatError(f"{path}: not a type-tagged vg stream (truncated, or another container)") from None
    if not out:
        raise VGFormatError(f"{path}: no {tag.decode()} payload")
    return b"".join(out)


def _unpack(words: np.ndarray, width: int, n: int) -> np.ndarray:
    """n entries of `width` bits from little-endian 64-bit words (sdsl::int_vector's layout) -> uint64 [n]"""
    if n == 0 or width == 0:
        return np.zeros(n, dtype=np.uint64)
    if width == 64:
        return words[:n].astype(np.uint64)
    pos = np.arange(n, dtype=np.uint64) * np.uint64(width)
    wi = (pos >> np.uint64(6)).astype(np.int64)
    sh = pos & np.uint64(63)
    w = np.concatenate([words.astype(np.uint64), np.zeros(1, dtype=np.uint64)])
    lo = w[wi] >> sh
    hi = np.where(sh > 0, w[wi + 1] << ((np.uint64(64) - sh) & np.uint64(63)), np.uint64(0))
    return (lo | hi) & np.uint64((1 << width) - 1)


def _ones(words: np.ndarray, bits: int) -> np.ndarray:
    """positions of the set bits of an sdsl::bit_vector"""
    if bits == 0:
        return np.zeros(0, dtype=np.int64)
    b = np.unpackbits(np.ascontiguousarray(words).view(np.uint8), bitorder="little")[:bits]
    return np.flatnonzero(b).astype(np.int64)


class _Reader:
    """a cursor over an sdsl serialisation"""

    def __init__(self, buf: bytes, path: str, at: int = 0):
        self.b, self.path, self.o = buf, path, at

    def _need(self, n: int):
        if n < 0 or self.o + n > len(self.b):
            raise VGFormatError(f"{self.path}: a structure runs past the end of the file (offset {self.o}, {n} bytes)")

    def u64(self) -> int:
        self._need(8)
        v, = struct.unpack_from("<Q", self.b, self.o)
        self.o += 8
        return v

    def u8(self) -> int:
        self._need(1)
        v = self.b[self.o]
        self.o += 1
        return v

    def words(self, bits: int) -> np.ndarray:
        n = (bits + 63) >> 6
        self._need(8 * n)
        a = np.frombuffer(self.b, dtype="<u8", count=n, offset=self.o)
        self.o += 8 * n
        return a

    def int_vector0(self) -> Tuple[np.ndarray, int, int]:
        """sdsl::int_vector<0>: u64 bits, u8 width, words -> (words, width, entries)"""
        bits, width = self.u64(), self.u8()
        if width > 64:
            raise VGFormatError(f"{self.path}: int_vector of width {width} at offset {self.o - 9}")
        return self.words(bits), width, (bits // width if width else 0)

    def values0(self) -> np.ndarray:
        w, width, n = self.int_vector0()
        return _unpack(w, width, n)

    def bit_vector(self) -> Tuple[np.ndarray, int]:
        bits = self.u64()
        return self.words(bits), bits

    def skip_rank_v(self):
        self.words(self.u64())                                     # int_vector<64> of the basic blocks

    def skip_select_mcl(self):
        """sdsl::select_support_mcl: u64 arguments; if any: the superblock vector, the mini-or-long bits, one vector per
        4 096 arguments"""
        args = self.u64()
        if args:
            self.int_vector0()
            self.bit_vector()
            for _ in range((args + 4095) >> 12):
                self.int_vector0()

    def skip_rrr63(self):
        """sdsl::rrr_vector<63>: u64 size, block types, block numbers (a bit_vector), their pointers, rank samples, the
        inversion bits; its rank / select supports serialise nothing"""
        self.u64()
        self.int_vector0()
        self.bit_vector()
        self.int_vector0()
        self.int_vector0()
        self.bit_vector()


def _window(w: np.ndarray, pos: np.ndarray) -> np.ndarray:
    """the 64 bits that start at bit `pos` of the little-endian words w (w padded by one word)"""
    wi = (pos >> np.uint64(6)).astype(np.int64)
    sh = pos & np.uint64(63)
    lo = w[wi] >> sh
    hi = np.where(sh > 0, w[wi + 1] << ((np.uint64(64) - sh) & np.uint64(63)), np.uint64(0))
    return lo | hi


def _elias_delta_all(zwords: np.ndarray, zbits: int, samples: np.ndarray, size: int) -> np.ndarray:
    """sdsl::enc_vector<coder::elias_delta, 128>: entry i = the sample value of its block + the sum of the deltas since.
    samples = [value, bit pointer] per block (+ a closing pair).  A delta x >= 1 is written, least significant bit first, as
    k zeros and a one (k = bits(bits(x)) - 1), the low k bits of bits(x), the low bits(x) - 1 bits of x (x = 2^64 stands
    for 0).  All blocks are decoded in step: entry j of every block at once."""
    n_blocks = (size + ENC_DENS - 1) // ENC_DENS
    w = np.concatenate([zwords.astype(np.uint64), np.zeros(2, dtype=np.uint64)])
    val = samples[0:2 * n_blocks:2].astype(np.uint64).copy()
    pos = samples[1:2 * n_blocks:2].astype(np.uint64).copy()
    if len(pos) and int(pos.max()) > zbits:
        raise VGFormatError("Elias-delta stream: a block starts behind its end")
    out = np.empty((ENC_DENS, n_blocks), dtype=np.uint64)
    out[0] = val
    one = np.uint64(1)
    limit = np.uint64(zbits + 64)
    for j in range(1, min(ENC_DENS, size)):
        live = np.arange(n_blocks) * ENC_DENS + j < size
        pos = np.where(live, pos, np.uint64(0))
        win = _window(w, pos)
        low = win & (~win + one)                                  # the lowest set bit: 2^k
        if ((low == 0) | (low > np.uint64(64)))[live].any():
            raise VGFormatError("Elias-delta stream: no code here")
        k = np.zeros(n_blocks, dtype=np.uint64)
        for b in range(1, 7):
            k[low == np.uint64(1 << b)] = b
        pos = pos + k + one
        ln = (_window(w, pos) & ((one << k) - one)) + (one << k)   # bits(x); k == 0: 1
        pos = pos + k
        body = ln - one                                            # 0 .. 64
        full = body >= np.uint64(64)
        sh = np.where(full, np.uint64(0), body)
        d = np.where(full, _window(w, pos), (_window(w, pos) & ((one << sh) - one)) + (one << sh))
        pos = pos + body
        if (pos[live] > limit).any():
            raise VGFormatError("Elias-delta stream runs out")
        val = val + np.where(live, d, np.uint64(0))               # (wraps: the differences are taken modulo 2^64)
        out[j] = val
    return out.T.reshape(-1)[:size].copy()


class XG:
    """nodes (ids, sequences), edges (indices into the node arrays, all end -> start) and the embedded paths of one XG"""

    def __init__(self, path: str):
        self.path = path
        raw = open(path, "rb").read()
        p = raw if raw[:4] == struct.pack(">I", XG_MAGIC) else tagged_payload(raw, b"XG", path)
        if len(p) < 56:
            raise VGFormatError(f"{path}: too short for an XG")
        magic, self.version = struct.unpack_from(">II", p, 0)
        if magic != XG_MAGIC:
            raise VGFormatError(f"{path}: bad XG magic {magic:#x}")
        if self.version not in XG_VERSIONS:
            raise VGFormatError(f"{path}: XG version {self.version}; this reader decodes version(s) "
                                f"{', '.join(map(str, XG_VERSIONS))} (build the index from FASTA + VCF instead: "
                                f"GraphIndex.from_fasta_vcf)")
        r = _Reader(p, path, 8)
        self.seq_length, n_nodes, n_edges, n_paths, self.min_id, self.max_id = (r.u64() for _ in range(6))
        rank_ids = r.values0()
        if len(rank_ids) != n_nodes:
            raise VGFormatError(f"{path}: id vector holds {len(rank_ids)} entries for {n_nodes} nodes")
        g = r.values0().astype(np.int64)
        gbv, gbits = r.bit_vector()
        if gbits != len(g):
            raise VGFormatError(f"{path}: graph bit vector of {gbits} bits beside a graph vector of {len(g)} entries")
        r.skip_rank_v()
        r.skip_select_mcl()
        sw, swidth, sn = r.int_vector0()
        if sn != self.seq_length or swidth > 3:
            raise VGFormatError(f"{path}: sequence vector of {sn} x {swidth} bits for {self.seq_length} bases")
        self.bases = _XG_BASES[_unpack(sw, swidth, sn).astype(np.int64)]
        r.bit_vector()                                            # s_bv: node starts (the records say the same)
        r.skip_rank_v()
        r.skip_select_mcl()
        names = bytes(r.values0().astype(np.uint8))
        # ---- the node records
        rec = _ones(gbv, gbits)
        if len(rec) != n_nodes or (n_nodes and (rec[0] != 0 or rec[-1] + 5 > len(g))):
            raise VGFormatError(f"{path}: {len(rec)} node records for {n_nodes} nodes")
        self.rec_off = rec
        self.ids = g[rec]
        self.start = g[rec + 1]
        self.length = g[rec + 2]
        n_to, n_from = g[rec + 3], g[rec + 4]
        ends = rec + 5 + n_to + n_from
        if n_nodes and (not np.array_equal(ends[:-1], rec[1:]) or ends[-1] != len(g)):
            raise VGFormatError(f"{path}: the node records do not tile the graph vector")
        if n_nodes and ((self.start + self.length).max() > self.seq_length or self.length.min() < 1):
            raise VGFormatError(f"{path}: a node's sequence lies outside the sequence vector")
        if int(n_from.sum()) != n_edges or int(n_to.sum()) != n_edges:
            raise VGFormatError(f"{path}: {int(n_from.sum())} / {int(n_to.sum())} edge entries, the header says {n_edges} edges")
        src = np.repeat(np.arange(n_nodes, dtype=np.int64), n_from)
        inner = np.arange(len(src), dtype=np.int64) - np.repeat(np.cumsum(n_from) - n_from, n_from)
        ent = g[np.repeat(rec + 5 + n_to, n_from) + inner]
        if (ent & 1).any():
            raise VGFormatError(f"{path}: a reversing edge -- not a graph `vg construct` writes")
        zz = ent >> 1
        delta = np.where(zz & 1, -((zz + 1) >> 1), zz >> 1)
        tgt = np.repeat(rec, n_from) + delta
        dst = np.searchsorted(rec, tgt)
        if len(dst) and ((dst >= n_nodes).any() or not np.array_equal(rec[np.minimum(dst, n_nodes - 1)], tgt)):
            raise VGFormatError(f"{path}: an edge does not lead to a node record")
        if len(src) != n_edges:
            raise VGFormatError(f"{path}: {len(src)} edges decoded, the header says {n_edges}")
        self.edge_from, self.edge_to = src, dst
        # ---- the paths: behind the names' suffix array, which is skipped by finding where the block decodes
        self.path_names = [s.decode() for s in names.replace(b"#", b"").split(b"$") if s]
        if len(self.path_names) != n_paths:
            raise VGFormatError(f"{path}: {len(self.path_names)} path names for {n_paths} paths")
        self.paths: Dict[str, np.ndarray] = {}
        self._edge_key = None
        if n_paths:
            self._read_paths(p, r.o, n_paths)

    def _path_at(self, p: bytes, at: int) -> Tuple[np.ndarray, int]:
        """the XGPath serialised at `at` -> (node indices of its steps, offset behind it); VGFormatError if there is none"""
        r = _Reader(p, self.path, at)
        min_handle = r.u64()
        size = r.u64()
        if size == 0 or size > 64 * len(self.ids) + 64 or (min_handle >> 1) > int(self.rec_off[-1]):
            raise VGFormatError("no path here")
        zw, _, _ = r.int_vector0()
        zbits = len(zw) * 64
        samples = r.values0()
        if len(samples) != 2 * ((size + ENC_DENS - 1) // ENC_DENS + 1):
            raise VGFormatError("no path here")
        handles = _elias_delta_all(zw, zbits, samples, size) + np.uint64(min_handle)
        if (handles & np.uint64(1)).any():
            raise VGFormatError(f"{self.path}: a path visits a node in reverse -- not a graph `vg construct` writes")
        off = (handles >> np.uint64(1)).astype(np.int64)
        idx = np.searchsorted(self.rec_off, off)
        if (idx >= len(self.rec_off)).any() or not np.array_equal(self.rec_off[idx], off):
            raise VGFormatError("no path here")
        if self._edge_key is None:                                # (once: candidates that get this far are few, edges are many)
            self._edge_key = np.sort(self.edge_from * len(self.ids) + self.edge_to)
        want = idx[:-1] * len(self.ids) + idx[1:]
        at_ = np.searchsorted(self._edge_key, want)
        if len(want) and ((at_ >= len(self._edge_key)).any()
                          or not np.array_equal(self._edge_key[np.minimum(at_, len(self._edge_key) - 1)], want)):
            raise VGFormatError("no path here")                   # (two steps in a row that no edge joins)
        r.skip_rrr63()
        r.u8()                                                    # is_circular
        return idx, r.o

    def _read_paths(self, p: bytes, lo: int, n_paths: int):
        pat = struct.pack("<Q", n_paths)
        at = lo
        while True:
            at = p.find(pat, at)
            if at < 0:
                raise VGFormatError(f"{self.path}: the path block was not found (the reference path is what region "
                                    f"coordinates are counted along: nothing can be extracted without it)")
            try:
                o = at + 8
                got = []
                for _ in range(n_paths):
                    idx, o = self._path_at(p, o)
                    got.append(idx)
                break
            except (VGFormatError, struct.error, IndexError):
                at += 1
        for name, idx in zip(self.path_names, got):
            self.paths[name] = idx

    def sequence_of(self, node: int) -> bytes:
        return self.bases[int(self.start[node]):int(self.start[node] + self.length[node])].tobytes()


def _bytecode(buf, i: int) -> Tuple[int, int]:
    v = shift = 0
    while True:
        c = buf[i]
        i += 1
        v |= (c & 0x7F) << shift
        shift += 7
        if not c & 0x80:
            return v, i


class GBWT:
    """the records of a GBWT (version 4, as `vg index -G` writes it): per oriented node its outgoing edges and, per visit
    in BWT order, the edge the visiting sequence leaves by"""

    def __init__(self, path: str):
        self.path = path
        raw = open(path, "rb").read()
        b = raw if raw[:4] == struct.pack("<I", GBWT_TAG) else tagged_payload(raw, b"GBWT", path)
        if len(b) < 56:
            raise VGFormatError(f"{path}: too short for a GBWT")
        tag, self.version = struct.unpack_from("<II", b, 0)
        if tag != GBWT_TAG:
            raise VGFormatError(f"{path}: GBWT header tag {tag:#x}")
        if self.version not in GBWT_VERSIONS:
            raise VGFormatError(f"{path}: GBWT version {self.version}; this reader decodes version(s) "
                                f"{', '.join(map(str, GBWT_VERSIONS))} (build the index from FASTA + VCF instead: "
                                f"GraphIndex.from_fasta_vcf)")
        self.sequences, self.size, self.offset, self.alphabet_size, flags = struct.unpack_from("<5Q", b, 8)
        self.bidirectional = bool(flags & 1)
        r = _Reader(b, path, 48)
        self.records = r.u64()
        data_len, wl = r.u64(), r.u8()
        low = r.values0().astype(np.int64)
        hw, hbits = r.bit_vector()
        r.skip_select_mcl()
        r.skip_select_mcl()
        ones = _ones(hw, hbits)
        if len(ones) != self.records or len(low) != self.records:
            raise VGFormatError(f"{path}: the record index holds {len(ones)} starts for {self.records} records")
        self.starts = ((ones - np.arange(len(ones), dtype=np.int64)) << wl) | low
        r._need(data_len)
        self.data = memoryview(b)[r.o:r.o + data_len]
        self.data_len = data_len
        if len(self.starts) and (np.diff(self.starts) < 0).any() or (len(self.starts) and self.starts[-1] > data_len):
            raise VGFormatError(f"{path}: the record index does not add up")

    def record(self, node: int) -> Tuple[List[Tuple[int, int]], List[int], List[int], int]:
        """GBWT node (0 = endmarker) -> ([(successor node, offset in its record)], edge rank per run, run lengths, visits)"""
        rix = 0 if node == 0 else node - self.offset
        if not 0 <= rix < self.records:
            return [], [], [], 0
        lo = int(self.starts[rix])
        hi = int(self.starts[rix + 1]) if rix + 1 < self.records else self.data_len
        buf = self.data[lo:hi]
        n = len(buf)
        sigma, i = _bytecode(buf, 0) if n else (0, 0)
        edges, to = [], 0
        for _ in range(sigma):
            d, i = _bytecode(buf, i)
            o, i = _bytecode(buf, i)
            to += d
            edges.append((to, o))
        ranks, runs, visits = [], [], 0
        if sigma >= 255:
            while i < n:
                rk, i = _bytecode(buf, i)
                ex, i = _bytecode(buf, i)
                ranks.append(rk)
                runs.append(ex + 1)
                visits += ex + 1
        elif sigma:
            per_byte = 256 // sigma                                # run lengths one byte can hold
            if n - i > 24:                                         # many runs: all of them at once if none is continued
                c = np.frombuffer(buf, dtype=np.uint8, offset=i).astype(np.int64)
                rn = c // sigma + 1
                if int(rn.max()) < per_byte:
                    return edges, (c % sigma).tolist(), rn.tolist(), int(rn.sum())
            while i < n:
                c = buf[i]
                i += 1
                rk, run = c % sigma, c // sigma + 1
                if run >= per_byte:                                # the longest: more follows
                    ex, i = _bytecode(buf, i)
                    run += ex
                ranks.append(rk)
                runs.append(run)
                visits += run
        return edges, ranks, runs, visits

    def first_nodes(self) -> np.ndarray:
        """per sequence the GBWT node it starts with (the endmarker's record: one visit per sequence, in order)"""
        e, ranks, runs, _ = self.record(0)
        body = np.repeat(np.asarray(ranks, dtype=np.int64), np.asarray(runs, dtype=np.int64))
        if len(body) != self.sequences:
            raise VGFormatError(f"{self.path}: the endmarker holds {len(body)} visits for {self.sequences} sequences")
        return np.array([w for w, _ in e], dtype=np.int64)[body] if len(e) else np.zeros(0, np.int64)

    def haplotype_sets(self, nodes: Sequence[int], edges: Sequence[Tuple[int, int]]):
        """-> ({node id: sequence ids that visit it forward}, {(from id, to id): sequence ids that take that edge}) for the
        node ids / edges asked for, by one pass over the forward nodes in id order (see the module's head)."""
        want_n = set(int(x) for x in nodes)
        want_e: Dict[int, set] = {}
        for u, w in edges:
            want_e.setdefault(int(u), set()).add(int(w))
        got_n: Dict[int, np.ndarray] = {}
        got_e: Dict[Tuple[int, int], np.ndarray] = {}
        ids: Dict[int, np.ndarray] = {}              # GBWT node -> sequence ids of its visits (filled by its predecessors)
        recs: Dict[int, tuple] = {}

        def rec_of(v):
            if v not in recs:
                recs[v] = self.record(v)
            return recs[v]

        def hand_on(v, mine):
            e, ranks, runs, visits = rec_of(v)
            recs.pop(v, None)
            if not e:
                return
            if visits != len(mine):
                raise VGFormatError(f"{self.path}: node {v >> 1}: {visits} visits in its record, {len(mine)} arrive")
            body = None if len(e) == 1 else np.repeat(np.asarray(ranks, dtype=np.int64), np.asarray(runs, dtype=np.int64))
            fwd = v and not (v & 1)
            for k, (w, off) in enumerate(e):
                part = mine if body is None else mine[body == k]
                if fwd and not (w & 1) and (v >> 1) in want_e and (w >> 1) in want_e[v >> 1]:
                    got_e[(v >> 1, w >> 1)] = part
                if w == 0 and fwd and len(part):
                    self.last_nodes.add(v >> 1)                  # (forward sequences end here)
                if w == 0 or (w & 1):                            # the sequence ends / a reverse node: not followed
                    continue
                if v and w <= v:
                    raise VGFormatError(f"{self.path}: an edge from node {v >> 1} back to node {w >> 1}: not a graph "
                                        f"whose ids go up along every walk (`vg construct` numbers them so)")
                have = ids.get(w)
                if have is None:
                    total = rec_of(w)[3]
                    if total > self.sequences:
                        raise VGFormatError(f"{self.path}: node {w >> 1}: {total} visits by {self.sequences} sequences")
                    if off == 0 and len(part) == total:          # the only predecessor: its block IS the node's visits
                        ids[w] = part
                        continue
                    have = ids[w] = np.full(total, -1, dtype=np.int64)
                    patched.add(w)
                if off + len(part) > len(have) or w not in patched:
                    raise VGFormatError(f"{self.path}: edge {v >> 1} -> {w >> 1} overruns the record of node {w >> 1}")
                have[off:off + len(part)] = part

        patched = set()                                          # nodes whose visits are put together from several blocks
        self.last_nodes = set()
        hand_on(0, np.arange(self.sequences, dtype=np.int64))
        while ids:
            v = min(ids)                                         # (few nodes are open at a time: a bubble's)
            mine = ids.pop(v)
            if v in patched:
                patched.discard(v)
                if (mine < 0).any():
                    raise VGFormatError(f"{self.path}: node {v >> 1} has visits no predecessor accounts for")
            if (v >> 1) in want_n:
                got_n[v >> 1] = mine
            hand_on(v, mine)
        return got_n, got_e


MAX_ALTS = 3


def graph_to_index(chrom: str, node_ids: np.ndarray, node_seqs, edge_from: np.ndarray, edge_to: np.ndarray,
                   ref_steps: np.ndarray, carriers, n_hap: int, where: str = "graph"):
    """Nodes (ids, sequences), edges (indices into the node arrays) and the reference path (node indices) of a graph as
    `vg construct` makes it of a reference and a VCF -> GraphIndex.  `carriers(alt_node_ids, gap_edges)` -> ({node id:
    haplotypes that visit it}, {(from id, to id): haplotypes that take the edge}) (None: no haplotypes).

    The path's nodes spell the reference.  Every other node -- or chain of nodes: a long allele chopped up -- is an alternate
    allele that replaces the reference bases between the reference positions it hangs on: a substitution where it replaces
    as many bases as it has (one site per mismatching position), an insertion where it replaces none, otherwise the
    substitutions of the common length and an insertion / deletion of the rest behind them; an edge that skips reference
    bases is a deletion (also one that leaves or enters an alternate allele: its carriers have both).  Sites are put
    together by the rules of the VCF reader (csrc/vcf_ingest.cpp: one substitution site per position with up to three
    alternates, insertions and deletions sites of their own behind their anchor, order substitution < insertion <
    deletion; equal alleles merge their carriers), so that both routes give the same index for the same data."""
    from .extract_regions import GraphIndex
    n = len(node_ids)
    ref_steps = np.asarray(ref_steps, dtype=np.int64)
    step_of = np.full(n, -1, dtype=np.int64)
    step_of[ref_steps] = np.arange(len(ref_steps))
    if (step_of[ref_steps] != np.arange(len(ref_steps))).any():
        raise VGFormatError(f"{where}: path {chrom} visits a node twice")
    if isinstance(node_seqs, tuple):                              # (bases, start, length): a node's bases = bases[start:start+length]
        bases, s_at, length = (np.asarray(a) for a in node_seqs)
        length = length.astype(np.int64)
    else:
        length = np.array([len(q) for q in node_seqs], dtype=np.int64)
        s_at = np.cumsum(length) - length
        bases = np.frombuffer(b"".join(node_seqs), dtype=np.uint8)
    seq_of = lambda v: bases[int(s_at[v]):int(s_at[v] + length[v])].tobytes()      # noqa: E731
    lens = length[ref_steps]
    r_start = np.zeros(n, dtype=np.int64)
    r_start[ref_steps] = np.cumsum(lens) - lens
    ref_len = int(lens.sum())
    ref = bases[np.repeat(s_at[ref_steps] - r_start[ref_steps], lens) + np.arange(ref_len)] if ref_len else np.zeros(0, np.uint8)
    ref = np.ascontiguousarray(ref, dtype=np.uint8)
    on_ref = step_of >= 0
    order = np.argsort(edge_from, kind="stable")
    et = edge_to[order]
    succ_at = np.searchsorted(edge_from[order], np.arange(n + 1))
    order_p = np.argsort(edge_to, kind="stable")
    pf = edge_from[order_p]
    pred_at = np.searchsorted(edge_to[order_p], np.arange(n + 1))
    succs = lambda v: et[succ_at[v]:succ_at[v + 1]]                # noqa: E731
    preds = lambda v: pf[pred_at[v]:pred_at[v + 1]]                # noqa: E731
    # ---- coordinates: start[v] = the reference offset a walk stands at when it enters v, end[v] = when it leaves v
    start = np.where(on_ref, r_start, -1)
    end = np.where(on_ref, r_start + length, -1)
    alt_nodes = np.flatnonzero(~on_ref).tolist()
    chain_head = np.arange(n, dtype=np.int64)                     # an allele longer than a node: its nodes in a row
    for v in alt_nodes:                                           # (ids go up along every walk: a head comes before its tail)
        p = preds(v)
        if len(p) == 1 and not on_ref[p[0]] and len(succs(int(p[0]))) == 1:
            chain_head[v] = chain_head[int(p[0])]
    members: Dict[int, List[int]] = {}
    for v in alt_nodes:
        members.setdefault(int(chain_head[v]), []).append(v)
    a_start: Dict[int, int] = {}
    a_end: Dict[int, int] = {}
    for h, mem in members.items():
        # a deletion that ends where this allele starts adds the node in front of it to the predecessors (and likewise
        # behind): the allele starts behind the LAST reference predecessor and ends at the FIRST reference successor
        p = preds(h)
        rp = p[on_ref[p]]
        if len(rp):
            a_start[h] = int(end[rp].max())
        elif len(p) == 0:
            a_start[h] = 0
        s = succs(mem[-1])
        rs = s[on_ref[s]]
        if len(rs):
            a_end[h] = int(start[rs].min())
        elif len(s) == 0:
            a_end[h] = ref_len
    for _ in range(8):          # an allele that hangs on alternate alleles only takes its ends from them
        missing = [h for h in members if h not in a_start or h not in a_end]
        if not missing:
            break
        for h in missing:
            if h not in a_start:
                known = [a_end[int(chain_head[q])] for q in preds(h).tolist() if int(chain_head[q]) in a_end and not on_ref[q]]
                if known:
                    a_start[h] = max(known)
            if h not in a_end:
                known = [a_start[int(chain_head[q])] for q in succs(members[h][-1]).tolist()
                         if int(chain_head[q]) in a_start and not on_ref[q]]
                if known:
                    a_end[h] = min(known)
    for h, mem in members.items():
        if h not in a_start or h not in a_end or a_end[h] < a_start[h]:
            raise VGFormatError(f"{where}: node {int(node_ids[h])} is no allele between two reference positions -- not a "
                                f"graph `vg construct` writes")
        for v in mem:                       # entered at the allele's start, left at its end; an edge inside the chain skips nothing
            start[v] = a_start[h] if v == h else a_end[h]
            end[v] = a_end[h]
    ef_l, et_l = edge_from.tolist(), edge_to.tolist()
    gap = []
    for u, w in zip(ef_l, et_l):
        if not on_ref[u] and not on_ref[w] and chain_head[u] == chain_head[w]:
            continue
        if end[u] > start[w]:
            raise VGFormatError(f"{where}: the edge {int(node_ids[u])} -> {int(node_ids[w])} goes backwards along the reference")
        if end[u] < start[w]:
            gap.append((u, w))
    # ---- carriers
    hw = (n_hap + 63) // 64
    node_sets, edge_sets = ({}, {})
    if carriers is not None and n_hap:
        node_sets, edge_sets = carriers([int(node_ids[h]) for h in members],
                                        [(int(node_ids[u]), int(node_ids[w])) for u, w in gap])

    # ---- atoms: (position, kind 0 substitution / 1 insertion / 2 deletion, payload, carriers, node id: file order in a tie)
    atoms = []
    skipped = 0
    for h, mem in members.items():
        s, e = a_start[h], a_end[h]
        seq = b"".join(seq_of(v) for v in mem)
        who = node_sets.get(int(node_ids[h]))
        lr, la = e - s, len(seq)
        m = min(lr, la)
        key = int(node_ids[h])
        if any(c not in b"ACGT" for c in seq) or (la > m and s + m - 1 < 0) or (lr > m and s + m - 1 < 0):
            skipped += 1            # not a base string / an insertion or deletion in front of the first base: no anchor
            continue
        for j in range(m):
            if seq[j] != ref[s + j]:
                atoms.append((s + j, 0, seq[j], who, key))
        if la > m:
            atoms.append((s + m - 1, 1, seq[m:], who, key))
        elif lr > m:
            atoms.append((s + m - 1, 2, lr - m, who, key))
    for (u, w) in gap:
        s, e = int(end[u]), int(start[w])
        if s - 1 < 0:
            skipped += 1
            continue
        atoms.append((s - 1, 2, e - s, edge_sets.get((int(node_ids[u]), int(node_ids[w]))), int(node_ids[w])))
    atoms.sort(key=lambda a: (a[0], a[1], a[4]))
    pos, dl, il, io, na, ab, ins = [], [], [], [], [], [], bytearray()
    placed = []                              # (site, slot, carriers): ORed into ONE bit array once the sites are counted

    def add_site(p, d, i_len, i_off):
        pos.append(p)
        dl.append(d)
        il.append(i_len)
        io.append(i_off)
        na.append(1)
        ab.append([0] * MAX_ALTS)

    def same_anchor(p):                      # the sites already made at this anchor, latest first
        for x in range(len(pos) - 1, -1, -1):
            if pos[x] != p:
                return
            yield x

    for (p, kind, payload, who, _) in atoms:
        if kind == 0:
            if pos and pos[-1] == p and dl[-1] == 0 and il[-1] == 0:
                slot = next((x for x in range(na[-1]) if ab[-1][x] == payload), -1)
                if slot < 0:
                    if na[-1] >= MAX_ALTS:
                        skipped += 1
                        continue
                    slot = na[-1]
                    na[-1] += 1
                    ab[-1][slot] = payload
                placed.append((len(pos) - 1, slot, who))
            else:
                add_site(p, 0, 0, 0)
                ab[-1][0] = payload
                placed.append((len(pos) - 1, 0, who))
        elif kind == 1:
            dup = next((x for x in same_anchor(p) if il[x] == len(payload) and bytes(ins[io[x]:io[x] + il[x]]) == payload), -1)
            if dup < 0:
                add_site(p, 0, len(payload), len(ins))
                ins.extend(payload)
                dup = len(pos) - 1
            placed.append((dup, 0, who))
        else:
            dup = next((x for x in same_anchor(p) if dl[x] == payload), -1)
            if dup < 0:
                add_site(p, payload, 0, 0)
                dup = len(pos) - 1
            placed.append((dup, 0, who))
    V = len(pos)
    alt_bits = None
    if V and hw:
        alt_bits = np.zeros((V, MAX_ALTS, hw), dtype=np.uint64)
        for site, slot, who in placed:
            if who is not None and len(who):
                h = np.unique(np.asarray(who, dtype=np.int64))
                np.bitwise_or.at(alt_bits[site, slot], h >> 6, np.uint64(1) << (h & 63).astype(np.uint64))
    return GraphIndex(chrom, ref, np.asarray(pos, np.int32), np.asarray(na, np.uint8),
                      np.asarray(ab, np.uint8).reshape(V, MAX_ALTS), alt_bits, n_hap, skipped,
                      del_len=np.asarray(dl, np.int32), ins_len=np.asarray(il, np.int32), ins_off=np.asarray(io, np.int32),
                      ins_bases=np.frombuffer(bytes(ins), dtype=np.uint8))


def index_from_vg(xg_path: str, gbwt_path: Optional[str] = None, chrom: Optional[str] = None, path_name: Optional[str] = None):
    """XG (+ the GBWT beside it) -> GraphIndex of the embedded path `path_name` (default: the path named `chrom`, or the
    only path there is).  Without a GBWT the index carries no haplotypes.  Raises VGFormatError for what is not decoded."""
    try:
        return _index_from_vg(xg_path, gbwt_path, chrom, path_name)
    except VGFormatError:
        raise
    except (IndexError, KeyError, struct.error, OverflowError, MemoryError, ValueError, ZeroDivisionError) as e:
        # a damaged file that got past the structural checks: still the reader's error, not a stray exception
        raise VGFormatError(f"{xg_path}{' / ' + gbwt_path if gbwt_path else ''}: not decodable ({type(e).__name__}: {e})") from e


def _index_from_vg(xg_path, gbwt_path, chrom, path_name):
    xg = XG(xg_path)
    if not xg.paths:
        raise VGFormatError(f"{xg_path}: no embedded path: region coordinates have nothing to refer to")
    name = path_name if path_name is not None else chrom
    if name is None or name not in xg.paths:
        if len(xg.paths) == 1 and path_name is None:
            name = next(iter(xg.paths))     # one chromosome per XG (constructVG.py:296-402): its path, whatever the file is called
        else:
            raise VGFormatError(f"{xg_path}: no path named {name!r} (paths: {', '.join(xg.paths)})")
    ids, ef, et, steps = xg.ids, xg.edge_from, xg.edge_to, xg.paths[name]
    keep = np.arange(len(ids))
    if len(xg.paths) > 1:
        # a whole-genome XG: the chromosome is the connected component its path lies in
        from scipy.sparse import coo_matrix
        from scipy.sparse.csgraph import connected_components
        n = len(ids)
        _, label = connected_components(coo_matrix((np.ones(len(ef), np.int8), (ef, et)), shape=(n, n)), directed=False)
        keep = np.flatnonzero(label == label[steps[0]])
        new_at = np.full(n, -1, dtype=np.int64)
        new_at[keep] = np.arange(len(keep))
        if (new_at[steps] < 0).any():
            raise VGFormatError(f"{xg_path}: path {name} is not connected")
        sel = new_at[ef] >= 0
        ids, ef, et, steps = ids[keep], new_at[ef[sel]], new_at[et[sel]], new_at[steps]
    n_hap, carriers = 0, None
    if gbwt_path:
        # The haplotypes of this chromosome: the forward sequences that start in its component (a whole-genome GBWT holds
        # one sequence per haplotype and chromosome), numbered in their order.  `vg find -H` counts the sequences a k-mer
        # lies on, and so do the kernels -- with "carries no alternate allele of the site" standing for the reference
        # allele, which is only right for sequences that run through the WHOLE chromosome: one that starts or ends inside
        # it (a haplotype broken at a phase break) is refused rather than counted where it does not go.
        gb = GBWT(gbwt_path)
        firsts = gb.first_nodes()
        fwd = np.flatnonzero((firsts & 1) == 0)
        fwd = fwd[np.isin(firsts[fwd] >> 1, ids)]
        n_hap = len(fwd)
        sources = set(ids[np.setdiff1d(np.arange(len(ids)), et)].tolist())
        sinks = set(ids[np.setdiff1d(np.arange(len(ids)), ef)].tolist())
        inside = sorted(set((firsts[fwd] >> 1).tolist()) - sources)
        if inside:
            raise VGFormatError(f"{gbwt_path}: a haplotype starts at node {inside[0]}, inside the chromosome: haplotypes in "
                                f"pieces (phase breaks) are not modelled")

        def carriers(nodes, edges):
            ns, es = gb.haplotype_sets(nodes, edges)
            inside = sorted(gb.last_nodes & set(ids.tolist()) - sinks)
            if inside:
                raise VGFormatError(f"{gbwt_path}: a haplotype ends at node {inside[0]}, inside the chromosome: haplotypes "
                                    f"in pieces (phase breaks) are not modelled")

            def dense(v):
                k = np.searchsorted(fwd, v)
                if len(v) and ((k >= len(fwd)).any() or not np.array_equal(fwd[np.minimum(k, len(fwd) - 1)], v)):
                    raise VGFormatError(f"{gbwt_path}: a sequence that starts on another chromosome visits this one")
                return k

            return {k: dense(v) for k, v in ns.items()}, {k: dense(v) for k, v in es.items()}

    return graph_to_index(chrom if chrom is not None else name, ids, (xg.bases, xg.start[keep], xg.length[keep]), ef, et, steps,
                          carriers, n_hap, where=xg_path)
