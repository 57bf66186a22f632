"""TEST INFRASTRUCTURE: writers of the two vg index formats as grafimo_amd/vg_files.py documents them (XG version 15, GBWT
version 4), for graphs `vg` itself never wrote here -- there is no vg binary in this image.  What they can show: that the
readers handle more than the two tiny files the reference ships (several paths, hundreds of haplotypes, runs longer than a
byte holds, N bases, several message groups, junk where the unparsed suffix array sits).  What they cannot show: that vg
writes these bytes -- that is pinned by the reference's own files only (tests/test_vg_files.py, first three tests).
Structures the readers skip (rank / select supports, the path names' suffix array, a path's rrr_vector) are written as
stand-ins of the right SHAPE with arbitrary content."""
import struct
from typing import Dict, List, Sequence, Tuple

import numpy as np


def u64(v: int) -> bytes:
    return struct.pack("<Q", v)


def varint(v: int) -> bytes:
    out = b""
    while True:
        out += bytes([(v & 0x7F) | (0x80 if v > 0x7F else 0)])
        v >>= 7
        if not v:
            return out


def pack_fields(values, widths) -> bytes:
    """fields of widths[i] <= 64 bits, one behind the other, least significant bit first -> whole 64-bit words"""
    values = np.asarray(values, dtype=np.uint64)
    widths = np.asarray(widths, dtype=np.uint64)
    total = int(widths.sum())
    words = np.zeros((total + 63) // 64 + 1, dtype=np.uint64)
    if len(values):
        pos = np.cumsum(widths) - widths
        wi = (pos >> np.uint64(6)).astype(np.int64)
        sh = pos & np.uint64(63)
        np.bitwise_or.at(words, wi, values << sh)
        spill = sh + widths > np.uint64(64)
        np.bitwise_or.at(words, wi[spill] + 1, values[spill] >> (np.uint64(64) - sh[spill]))
    return words[:(total + 63) // 64].tobytes()


def int_vector0(values: Sequence[int], width: int = 0) -> bytes:
    values = np.asarray([int(v) for v in values] if not isinstance(values, np.ndarray) else values, dtype=np.uint64)
    width = width or max(1, int(values.max(initial=0)).bit_length())
    assert width == 64 or not len(values) or int(values.max()) < (1 << width)
    return u64(len(values) * width) + bytes([width]) + pack_fields(values, np.full(len(values), width))


def bit_vector(bits: Sequence[int]) -> bytes:
    b = np.asarray(bits, dtype=bool).astype(np.uint8)
    packed = np.packbits(b, bitorder="little")
    return u64(len(b)) + packed.tobytes() + bytes((-len(packed)) % 8)


def rank_v(n_bits: int) -> bytes:
    words = 2 * ((n_bits >> 9) + 1)
    return u64(64 * words) + bytes(8 * words)


def select_mcl(n_ones: int) -> bytes:
    if n_ones == 0:
        return u64(0)
    sb = (n_ones + 4095) >> 12
    out = u64(n_ones) + int_vector0(list(range(sb)), 7) + bit_vector([1] * sb)
    for i in range(sb):
        out += int_vector0([3, 1, 4, 1, 5], 11)
    return out


def enc_vector(values: Sequence[int], dens: int = 128) -> bytes:
    """sdsl::enc_vector<coder::elias_delta, dens>: per entry that is no sample the Elias-delta code of its difference to the
    entry before (modulo 2^64; 2^64 for 0): k zeros and a one (k = bits(bits(d)) - 1), the low k bits of bits(d), the low
    bits(d) - 1 bits of d -- least significant bit first"""
    fields_v, fields_w, samples, pos = [], [], [], 0
    for i, v in enumerate(values):
        if i % dens == 0:
            samples += [v, pos]
            continue
        d = (v - values[i - 1]) & ((1 << 64) - 1) or (1 << 64)
        len_1 = d.bit_length() - 1
        k = (len_1 + 1).bit_length() - 1
        fields_v.append(1 << k)
        fields_w.append(k + 1)
        if k:
            fields_v += [(len_1 + 1) & ((1 << k) - 1), d & ((1 << len_1) - 1)]
            fields_w += [k, len_1]
        pos += 2 * k + 1 + (len_1 if k else 0)
    samples += [0, pos + 1]
    return u64(len(values)) + u64(pos) + bytes([1]) + pack_fields(fields_v, fields_w) + int_vector0(samples)


def rrr63(n_bits: int) -> bytes:
    blocks = (n_bits + 62) // 63
    return (u64(n_bits) + int_vector0([i % 60 for i in range(blocks)], 6) + bit_vector([i % 3 == 0 for i in range(5 * blocks)]) +
            int_vector0([7] * (blocks // 32 + 1)) + int_vector0([0, blocks], 0) + bit_vector([0] * (blocks // 32 + 1)))


def tagged(tag: bytes, payload: bytes, msg_size: int = 1 << 30, msgs_per_group: int = 1 << 30) -> bytes:
    msgs = [payload[i:i + msg_size] for i in range(0, len(payload), msg_size)] or [b""]
    out = b""
    for g in range(0, len(msgs), msgs_per_group):
        grp = msgs[g:g + msgs_per_group]
        out += varint(len(grp) + 1) + varint(len(tag)) + tag
        for m in grp:
            out += varint(len(m)) + m
    return out


XG_CODE = {ord("A"): 0, ord("T"): 1, ord("C"): 2, ord("G"): 3, ord("N"): 4}


def xg_bytes(nodes: Dict[int, bytes], edges: Sequence[Tuple[int, int]], paths: Dict[str, List[int]], version: int = 15,
             junk: bytes = b"", msg_size: int = 1 << 30, msgs_per_group: int = 1 << 30) -> bytes:
    ids = sorted(nodes)
    to_of: Dict[int, List[int]] = {i: [] for i in ids}
    from_of: Dict[int, List[int]] = {i: [] for i in ids}
    for a, b in edges:
        from_of[a].append(b)
        to_of[b].append(a)
    rec_at, at, seq_at, s = {}, 0, {}, 0
    for i in ids:
        rec_at[i] = at
        at += 5 + len(to_of[i]) + len(from_of[i])
        seq_at[i] = s
        s += len(nodes[i])

    def entry(me: int, other: int) -> int:
        d = rec_at[other] - rec_at[me]
        return ((2 * d) if d >= 0 else (-2 * d - 1)) << 1

    g, marks = [], []
    for i in ids:
        rec = [i, seq_at[i], len(nodes[i]), len(to_of[i]), len(from_of[i])]
        rec += [entry(i, o) for o in to_of[i]] + [entry(i, o) for o in from_of[i]]
        g += rec
        marks += [1] + [0] * (len(rec) - 1)
    bases = b"".join(nodes[i] for i in ids)
    s_bv = [0] * (len(bases) + 1)
    for i in ids:
        s_bv[seq_at[i]] = 1
    s_bv[len(bases)] = 1
    names = "".join(f"#{n}$" for n in paths).encode()
    out = struct.pack(">II", 0xF6F596A1, version)
    out += b"".join(u64(v) for v in (len(bases), len(ids), len(edges), len(paths), ids[0], ids[-1]))
    out += int_vector0(ids) + int_vector0(g) + bit_vector(marks) + rank_v(len(marks)) + select_mcl(len(ids))
    out += int_vector0([XG_CODE[c] for c in bases], 3 if b"N" in bases else 2)
    out += bit_vector(s_bv) + rank_v(len(s_bv)) + select_mcl(len(ids) + 1)
    out += int_vector0(list(names), 7)
    out += junk                                                   # where pn_csa, pn_bv (+ supports) and pi_iv sit
    out += u64(len(paths))
    for name, steps in paths.items():
        handles = [rec_at[i] << 1 for i in steps]
        mn = min(handles)
        out += u64(mn) + enc_vector([h - mn for h in handles]) + rrr63(sum(len(nodes[i]) for i in steps)) + b"\x00"
    out += bit_vector([1] * len(ids)) + junk[:97]                 # (what follows the paths is not read)
    return tagged(b"XG", out, msg_size, msgs_per_group)


def bytecode(v: int) -> bytes:
    return varint(v)


def gbwt_bytes(walks: Sequence[Sequence[int]], bidirectional: bool = True, version: int = 4) -> bytes:
    """walks: per haplotype the node ids it visits, forward"""
    seqs = []
    for w in walks:
        seqs.append([2 * i for i in w])
        if bidirectional:
            seqs.append([2 * i + 1 for i in reversed(w)])
    all_nodes = sorted({v for s in seqs for v in s})
    offset = all_nodes[0] - 1
    alphabet = all_nodes[-1] + 1
    n_records = alphabet - offset
    # visits of a node in BWT order: by the reverse of what the sequence did before (its endmarker last: sequence id)
    visits: Dict[int, List[Tuple[tuple, int, int]]] = {}
    for j, s in enumerate(seqs):
        for i, v in enumerate(s):
            key = tuple(reversed(s[:i])) + (0, j)
            visits.setdefault(v, []).append((key, s[i - 1] if i else 0, s[i + 1] if i + 1 < len(s) else 0))
    for v in visits:
        visits[v].sort()
    from_counts: Dict[int, Dict[int, int]] = {}                   # w -> {predecessor: visits that came from it}
    for v, vs in visits.items():
        for _, pred, _ in vs:
            from_counts.setdefault(v, {}).setdefault(pred, 0)
            from_counts[v][pred] += 1

    def offset_of(v: int, w: int) -> int:
        return 0 if w == 0 else sum(c for p, c in from_counts[w].items() if p < v)

    def record(v: int, succ: List[int]) -> bytes:
        outs = sorted(set(succ))
        sigma = len(outs)
        rec, prev = bytecode(sigma), 0
        for w in outs:
            rec += bytecode(w - prev) + bytecode(offset_of(v, w))
            prev = w
        ranks = [outs.index(w) for w in succ]
        i = 0
        while i < len(ranks):
            j = i
            while j < len(ranks) and ranks[j] == ranks[i]:
                j += 1
            r, run = ranks[i], j - i
            if sigma >= 255:
                rec += bytecode(r) + bytecode(run - 1)
            else:
                per_byte = 256 // sigma
                if run < per_byte:
                    rec += bytes([r + sigma * (run - 1)])
                else:
                    rec += bytes([r + sigma * (per_byte - 1)]) + bytecode(run - per_byte)
            i = j
        return rec

    recs = [record(0, [s[0] for s in seqs])]
    for v in range(offset + 1, alphabet):
        recs.append(record(v, [nx for _, _, nx in visits[v]]) if v in visits else bytecode(0))
    assert len(recs) == n_records
    starts, data = [], b""
    for r in recs:
        starts.append(len(data))
        data += r
    wl = max(1, (len(data) // max(1, n_records)).bit_length())
    high = [0] * ((len(data) >> wl) + n_records + 1)
    for k, st in enumerate(starts):
        high[(st >> wl) + k] = 1
    total = sum(len(s) + 1 for s in seqs)
    out = struct.pack("<II", 0x6B376B37, version)
    out += b"".join(u64(v) for v in (len(seqs), total, offset, alphabet, 1 if bidirectional else 0))
    out += u64(n_records)
    out += u64(len(data)) + bytes([wl]) + int_vector0([st & ((1 << wl) - 1) for st in starts], wl) + bit_vector(high)
    out += select_mcl(n_records) + select_mcl(len(high) - n_records)
    out += data + b"\x00" * 40                                    # (the document-array samples and the metadata: not read)
    return tagged(b"GBWT", out)
