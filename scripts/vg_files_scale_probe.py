"""How long does grafimo_amd/vg_files.py take on a graph of realistic width?  A chain of N bubbles (anchor node -> alternate |
reference node -> next anchor), H haplotypes that take the alternate with probability AF each: the GBWT of it is written
directly (the visits of a node are known in closed form: what came from the alternate, then what came from the reference node),
the XG by tests/vg_encode.py's writer.  Times: XG(), GBWT(), haplotype_sets, index_from_vg; checks the haplotype bitsets.
Not a test (tests/test_vg_files.py holds the pins); host only.

    python scripts/vg_files_scale_probe.py [N=100000] [H=5096] [AF=0.02]
"""
import os
import struct
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import vg_encode as ve                      # noqa: E402
from grafimo_amd import vg_files as vf      # noqa: E402


def runs_bytes(ranks: np.ndarray, sigma: int) -> bytes:
    if len(ranks) == 0:
        return b""
    cut = np.flatnonzero(np.diff(ranks)) + 1
    starts = np.concatenate([[0], cut])
    lens = np.diff(np.concatenate([starts, [len(ranks)]]))
    out = bytearray()
    per_byte = 256 // sigma
    for r, run in zip(ranks[starts].tolist(), lens.tolist()):
        if run < per_byte:
            out.append(r + sigma * (run - 1))
        else:
            out.append(r + sigma * (per_byte - 1))
            out += ve.bytecode(run - per_byte)
    return bytes(out)


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    H = int(sys.argv[2]) if len(sys.argv) > 2 else 5096
    AF = float(sys.argv[3]) if len(sys.argv) > 3 else 0.02
    rng = np.random.default_rng(1)
    t0 = time.time()
    # ---- the graph: anchor 3i+1 (8 bases), alternate 3i+2, reference 3i+3 (one base each), last anchor 3N+1
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    nodes, edges, path = {}, [], []
    for i in range(N):
        a, alt, ref = 3 * i + 1, 3 * i + 2, 3 * i + 3
        nodes[a] = acgt[rng.integers(0, 4, 8)].tobytes()
        rb = int(rng.integers(0, 4))
        nodes[ref] = bytes([acgt[rb]])
        nodes[alt] = bytes([acgt[(rb + 1 + int(rng.integers(0, 3))) % 4]])
        edges += [(a, alt), (a, ref), (alt, a + 3), (ref, a + 3)]
        path += [a, ref]
    nodes[3 * N + 1] = b"ACGTACGT"
    path.append(3 * N + 1)
    tmp = tempfile.mkdtemp(prefix="gfm_vgscale_")
    xg_path, gbwt_path = os.path.join(tmp, "c.xg"), os.path.join(tmp, "c.gbwt")
    open(xg_path, "wb").write(ve.xg_bytes(nodes, edges, {"c": path}))
    print(f"XG written: {len(nodes)} nodes, {os.path.getsize(xg_path) / 1e6:.1f} MB in {time.time() - t0:.1f}s", flush=True)
    # ---- the GBWT (not bidirectional: H forward sequences), record by record
    t0 = time.time()
    af = rng.random(N) ** 3 * AF * 4
    order = np.arange(H, dtype=np.int64)
    carriers = []
    offset = 2 * 1 - 1                     # GBWT node of id 1 is 2: record 1
    recs = [ve.bytecode(1) + ve.bytecode(2) + ve.bytecode(0) + runs_bytes(np.zeros(H, np.int64), 1)]
    node_rec = {}
    for i in range(N):
        a, alt, ref = 3 * i + 1, 3 * i + 2, 3 * i + 3
        take = rng.random(H) < af[i]                     # per HAPLOTYPE
        t_ord = take[order]                              # per visit of the anchor, in BWT order
        alt_v, ref_v = order[t_ord], order[~t_ord]
        carriers.append(np.sort(alt_v))
        # anchor: edges to 2*alt (offset 0) and 2*ref (offset 0)
        node_rec[2 * a] = (ve.bytecode(2) + ve.bytecode(2 * alt) + ve.bytecode(0) + ve.bytecode(2 * ref - 2 * alt) + ve.bytecode(0) +
                           runs_bytes(np.where(t_ord, 0, 1), 2))
        nxt = 2 * (a + 3)
        node_rec[2 * alt] = ve.bytecode(1) + ve.bytecode(nxt) + ve.bytecode(0) + runs_bytes(np.zeros(len(alt_v), np.int64), 1)
        node_rec[2 * ref] = ve.bytecode(1) + ve.bytecode(nxt) + ve.bytecode(len(alt_v)) + runs_bytes(np.zeros(len(ref_v), np.int64), 1)
        order = np.concatenate([alt_v, ref_v])
    last = 2 * (3 * N + 1)
    node_rec[last] = ve.bytecode(1) + ve.bytecode(0) + ve.bytecode(0) + runs_bytes(np.zeros(H, np.int64), 1)
    alphabet = last + 1
    for v in range(offset + 1, alphabet):
        recs.append(node_rec.get(v, ve.bytecode(0)))
    starts = np.cumsum([0] + [len(r) for r in recs[:-1]])
    data = b"".join(recs)
    wl = max(1, (len(data) // len(recs)).bit_length())
    high = np.zeros((len(data) >> wl) + len(recs) + 1, dtype=np.uint8)
    high[(starts >> wl) + np.arange(len(recs))] = 1
    hi_words = np.packbits(high, bitorder="little")
    hi_words = np.concatenate([hi_words, np.zeros((-len(hi_words)) % 8, np.uint8)]).tobytes()
    low = (starts & ((1 << wl) - 1)).astype(np.uint64)
    big = np.zeros((len(low) * wl + 63) // 64 + 1, dtype=np.uint64)
    pos = np.arange(len(low), dtype=np.uint64) * np.uint64(wl)
    np.bitwise_or.at(big, (pos >> np.uint64(6)).astype(np.int64), low << (pos & np.uint64(63)))
    spill = (pos & np.uint64(63)) + np.uint64(wl) > np.uint64(64)
    np.bitwise_or.at(big, (pos[spill] >> np.uint64(6)).astype(np.int64) + 1,
                     low[spill] >> (np.uint64(64) - (pos[spill] & np.uint64(63))))
    low_iv = ve.u64(len(low) * wl) + bytes([wl]) + big[:(len(low) * wl + 63) // 64].tobytes()
    out = struct.pack("<II", 0x6B376B37, 4) + b"".join(ve.u64(v) for v in (H, H * (2 * N + 2), offset, alphabet, 0))
    out += ve.u64(len(recs)) + ve.u64(len(data)) + bytes([wl]) + low_iv + ve.u64(len(high)) + hi_words
    out += ve.select_mcl(len(recs)) + ve.select_mcl(len(high) - len(recs)) + data + bytes(40)
    open(gbwt_path, "wb").write(ve.tagged(b"GBWT", out))
    print(f"GBWT written: {len(recs)} records, {os.path.getsize(gbwt_path) / 1e6:.1f} MB in {time.time() - t0:.1f}s", flush=True)
    # ---- the readers
    t0 = time.time()
    xg = vf.XG(xg_path)
    t_xg = time.time() - t0
    t0 = time.time()
    gb = vf.GBWT(gbwt_path)
    t_gb = time.time() - t0
    t0 = time.time()
    ns, es = gb.haplotype_sets([3 * i + 2 for i in range(N)], [])
    t_sets = time.time() - t0
    bad = sum(1 for i in range(N) if not np.array_equal(np.sort(ns.get(3 * i + 2, np.zeros(0, np.int64))), carriers[i]))
    t0 = time.time()
    idx = vf.index_from_vg(xg_path, gbwt_path, "c")
    t_all = time.time() - t0
    print(f"N {N} bubbles, H {H} haplotypes: XG() {t_xg:.2f}s, GBWT() {t_gb:.2f}s, haplotype_sets {t_sets:.2f}s "
          f"({t_sets / (3 * N) * 1e6:.1f} us per node), index_from_vg {t_all:.2f}s; sites {len(idx.pos)}, wrong carrier sets {bad}")
    want = np.array([len(c) for c in carriers])
    got = np.array([sum(bin(int(w)).count("1") for w in idx.alt_bits[s, 0]) for s in range(min(len(idx.pos), 2000))])
    print("carrier counts of the first sites equal:", np.array_equal(got, want[:len(got)]), "; n_haplotypes", idx.n_haplotypes)
    import shutil
    shutil.rmtree(tmp)


if __name__ == "__main__":
    main()
