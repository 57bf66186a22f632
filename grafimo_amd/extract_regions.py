"""K-mer extraction on the GPU -- the `vg find -p CHR:S-E -x XG -H GBWT -K W -E` step of
extract_regions.py:180,225,326 for variation graphs that are a linear reference plus the SNP records
of a phased VCF (what `grafimo buildvg` feeds `vg construct`, constructVG.py:332).

    index = GraphIndex.from_fasta_vcf("chr22.fa", "chr22.vcf.gz", "22")
    graph = DeviceGraph(index)
    rows  = graph.extract([(19723256, 19723526), ...], width=19)    # device-resident rows
    df    = compute_results_from_graph(motif, graph, regions, args)  # extraction -> scoring, no TSV
    write_region_tsvs(index, rows, "out")                            # or the files GRAFIMO expects

Row semantics are those of vg's output as far as the reference's golden file pins them (32 rows of
tests/test_data/expected_results/expected_seqs.tsv, reproduced exactly incl. node paths); haplotype
counts follow "phased haplotypes of the VCF that carry every allele of the walk".  VCF records that
are not single-base substitutions (indels, MNPs) are not part of the graph: `GraphIndex.skipped`
counts them.  There is no CPU fallback: extraction needs libgrafimo_hip.so and a GPU.
"""
import ctypes
import gzip
import os
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import pandas as pd

from . import _native as nv
from .device import DeviceMotif, _stream_ptr, _torch
from .motif import Motif
from .resultsTmp import build_frame
from .utils import exception_handler

NODE_MAX = 32   # vg construct -m default: an invariant stretch is chopped into nodes of <= 32 bases
MAX_ALTS = 3


def _read_fasta_record(path: str, chrom: str) -> np.ndarray:
    parts, on = [], False
    with open(path, "rb") as fh:
        for line in fh:
            if line.startswith(b">"):
                if on:
                    break
                on = line[1:].split()[0].decode() == chrom
            elif on:
                parts.append(line.strip().upper())
    if not parts:
        raise ValueError(f"chromosome {chrom} not found in {path}")
    return np.frombuffer(b"".join(parts), dtype=np.uint8).copy()


class GraphIndex:
    """Host-side description of one chromosome's graph: reference bases, SNP sites and, per
    alternate allele, the bitset of haplotypes that carry it."""

    def __init__(self, chrom: str, ref: np.ndarray, pos, n_alts, alt_bases, alt_bits, n_haplotypes: int,
                 skipped: int = 0):
        self.chrom = chrom
        self.ref = np.ascontiguousarray(ref, dtype=np.uint8)
        self.pos = np.ascontiguousarray(pos, dtype=np.int32)
        self.n_alts = np.ascontiguousarray(n_alts, dtype=np.uint8)
        self.alt_bases = np.ascontiguousarray(alt_bases, dtype=np.uint8).reshape(len(self.pos), MAX_ALTS)
        self.n_haplotypes = int(n_haplotypes)
        self.hw = (self.n_haplotypes + 63) // 64
        self.alt_bits = None
        if alt_bits is not None and self.n_haplotypes:
            self.alt_bits = np.ascontiguousarray(alt_bits, dtype=np.uint64).reshape(len(self.pos), MAX_ALTS, self.hw)
        self.skipped = int(skipped)
        self._nodes = None

    @classmethod
    def from_fasta_vcf(cls, fasta: str, vcf: str, chrom: str, with_haplotypes: bool = True) -> "GraphIndex":
        ref = _read_fasta_record(fasta, chrom)
        pos, alts, gts, skipped = [], [], [], 0
        op = gzip.open if vcf.endswith(".gz") else open
        with op(vcf, "rt") as fh:
            for line in fh:
                if line[0] == "#":
                    continue
                f = line.rstrip("\n").split("\t")
                if f[0] != chrom:
                    continue
                r, a = f[3].upper(), f[4].upper().split(",")
                if len(r) != 1 or len(a) > MAX_ALTS or any(len(x) != 1 or x not in "ACGT" for x in a):
                    skipped += 1            # indel / MNP / symbolic allele: not part of this graph
                    continue
                p = int(f[1]) - 1
                if pos and p <= pos[-1]:
                    if p == pos[-1]:
                        skipped += 1        # second record at one position
                        continue
                    raise ValueError(f"{vcf}: records of {chrom} are not sorted by position")
                pos.append(p)
                alts.append(a)
                if with_haplotypes:
                    row = []
                    for s in f[9:]:
                        gt = s.split(":", 1)[0].replace("/", "|").split("|")
                        if len(gt) == 1:
                            gt = gt * 2
                        row += [int(x) if x.isdigit() else 0 for x in gt[:2]]
                    gts.append(row)
        V = len(pos)
        n_alts = np.array([len(a) for a in alts], dtype=np.uint8)
        alt_bases = np.zeros((V, MAX_ALTS), dtype=np.uint8)
        for i, a in enumerate(alts):
            alt_bases[i, :len(a)] = [ord(x) for x in a]
        H = len(gts[0]) if gts else 0
        bits = None
        if H:
            g = np.asarray(gts, dtype=np.int8)                       # [V, H]
            hw = (H + 63) // 64
            bits = np.zeros((V, MAX_ALTS, hw), dtype=np.uint64)
            padded = np.zeros((V, hw * 64), dtype=bool)
            for a in range(MAX_ALTS):
                padded[:, :H] = g == a + 1
                # bit h of word h // 64: little-endian bit order inside little-endian 64-bit words
                bits[:, a, :] = np.packbits(padded, axis=1, bitorder="little").view(np.uint64)
        return cls(chrom, ref, pos, n_alts, alt_bases, bits, H, skipped)

    # ---- node ids of `vg construct` on this graph (column 7 of the TSV; not used by GRAFIMO's scoring)
    def _node_table(self):
        if self._nodes is None:
            seg_start, seg_end, seg_id, site_ids = [], [], [], []
            nid, cur = 1, 0
            for p, na in zip(self.pos.tolist(), self.n_alts.tolist()):
                while cur < p:
                    e = min(cur + NODE_MAX, p)
                    seg_start.append(cur); seg_end.append(e); seg_id.append(nid)
                    nid += 1
                    cur = e
                alt_ids = list(range(nid, nid + na))     # alternate alleles first, then the reference allele
                nid += na
                site_ids.append([nid] + alt_ids)
                nid += 1
                cur = p + 1
            while cur < len(self.ref):
                e = min(cur + NODE_MAX, len(self.ref))
                seg_start.append(cur); seg_end.append(e); seg_id.append(nid)
                nid += 1
                cur = e
            self._nodes = (np.asarray(seg_start, dtype=np.int64), seg_end, seg_id, site_ids)
        return self._nodes

    def walk_alleles(self, p: int, width: int, walk: int) -> Tuple[int, List[int]]:
        """(first site, allele per site) of walk number `walk` of window p (last site fastest)."""
        i0 = int(np.searchsorted(self.pos, p, side="left"))
        i1 = int(np.searchsorted(self.pos, p + width, side="left"))
        alleles = [0] * (i1 - i0)
        for k in range(i1 - i0 - 1, -1, -1):
            n = 1 + int(self.n_alts[i0 + k])
            alleles[k] = walk % n
            walk //= n
        return i0, alleles

    def node_path(self, p: int, width: int, walk: int) -> List[int]:
        seg_start, seg_end, seg_id, site_ids = self._node_table()
        i0, alleles = self.walk_alleles(p, width, walk)
        out, cur, k = [], p, 0
        while cur < p + width:
            if k < len(alleles) and int(self.pos[i0 + k]) == cur:
                out.append(site_ids[i0 + k][alleles[k]])
                cur += 1
                k += 1
            else:
                j = int(np.searchsorted(seg_start, cur, side="right")) - 1
                out.append(seg_id[j])
                cur = seg_end[j]
        return out


class ExtractedKmers:
    """Device-resident rows of one extraction (torch tensors on the graph's device)."""

    def __init__(self, chrom, regions, width, kmers, start, stop, strand, freq, is_ref, region, walk):
        self.chrom, self.regions, self.width = chrom, list(regions), int(width)
        self.kmers, self.start, self.stop, self.strand = kmers, start, stop, strand
        self.freq, self.is_ref, self.region, self.walk = freq, is_ref, region, walk

    def __len__(self):
        return int(self.kmers.shape[0])

    def region_label(self, r: int) -> str:
        s, e = self.regions[r]
        return f"{self.chrom}:{s}-{e}"


class DeviceGraph:
    """gfm_graph_* handle of one chromosome on the current GPU."""

    def __init__(self, index: GraphIndex, device=None):
        torch = _torch()
        self.index = index
        self.device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            nv.check(nv.lib().gfm_graph_create(
                nv.ptr(index.ref), len(index.ref), len(index.pos), nv.ptr(index.pos), nv.ptr(index.n_alts),
                nv.ptr(index.alt_bases), nv.ptr(index.alt_bits) if index.alt_bits is not None else None,
                index.n_haplotypes if index.alt_bits is not None else 0, ctypes.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            nv.lib().gfm_graph_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def extract(self, regions: Sequence[Tuple[int, int]], width: int, stream=None) -> ExtractedKmers:
        """All rows of `vg find -p chrom:S-E -K width -E -H` for the given (S, E) regions."""
        torch = _torch()
        starts = np.ascontiguousarray([r[0] for r in regions], dtype=np.int64)
        stops = np.ascontiguousarray([r[1] for r in regions], dtype=np.int64)
        nw, nr = ctypes.c_int64(), ctypes.c_int64()
        with torch.cuda.device(self.device):
            nv.check(nv.lib().gfm_graph_plan(self._h, len(regions), nv.ptr(starts), nv.ptr(stops), int(width),
                                             ctypes.byref(nw), ctypes.byref(nr)))
            n = int(nr.value)
            dev = self.device
            kmers = torch.empty((n, width), dtype=torch.uint8, device=dev)
            start = torch.empty(n, dtype=torch.int64, device=dev)
            stop = torch.empty(n, dtype=torch.int64, device=dev)
            strand = torch.empty(n, dtype=torch.uint8, device=dev)
            freq = torch.empty(n, dtype=torch.int64, device=dev)
            is_ref = torch.empty(n, dtype=torch.uint8, device=dev)
            region = torch.empty(n, dtype=torch.int32, device=dev)
            walk = torch.empty(n, dtype=torch.int32, device=dev)
            if n:
                nv.check(nv.lib().gfm_graph_emit(self._h, kmers.data_ptr(), start.data_ptr(), stop.data_ptr(),
                                                 strand.data_ptr(), freq.data_ptr(), is_ref.data_ptr(),
                                                 region.data_ptr(), walk.data_ptr(), _stream_ptr(stream)))
        return ExtractedKmers(self.index.chrom, regions, width, kmers, start, stop, strand, freq, is_ref,
                              region, walk)


def write_region_tsvs(index: GraphIndex, rows: ExtractedKmers, out_dir: str) -> List[str]:
    """The files scan_graph leaves for compute_results: out_dir/width_W/CHR_S-E.tsv
    (extract_regions.py:165-170,180), seven tab-separated columns per row like vg's."""
    W = rows.width
    d = os.path.join(out_dir, f"width_{W}")
    os.makedirs(d, exist_ok=True)
    km = rows.kmers.cpu().numpy()
    start, stop = rows.start.cpu().numpy(), rows.stop.cpu().numpy()
    strand, freq = rows.strand.cpu().numpy(), rows.freq.cpu().numpy()
    is_ref, region, walk = rows.is_ref.cpu().numpy(), rows.region.cpu().numpy(), rows.walk.cpu().numpy()
    paths = []
    bounds = np.searchsorted(region, np.arange(len(rows.regions) + 1), side="left")
    for r in range(len(rows.regions)):
        label = rows.region_label(r)
        path = os.path.join(d, label.replace(":", "_") + ".tsv")
        with open(path, "w") as fh:
            for i in range(bounds[r], bounds[r + 1]):
                sg = chr(strand[i])
                p = int(start[i]) if sg == "+" else int(stop[i])
                nodes = index.node_path(p, W, int(walk[i]))
                if sg == "-":
                    nodes = nodes[::-1]
                fh.write("\t".join([
                    label, km[i].tobytes().decode(), f"{index.chrom}:{int(start[i])}{sg}",
                    f"{index.chrom}:{int(stop[i])}{sg}", str(int(freq[i])), "ref" if is_ref[i] else "non.ref",
                    "".join(f"{n}{sg}," for n in nodes)]) + "\n")
        paths.append(path)
    return paths


def read_bed_regions(bedfile: str) -> Dict[str, List[Tuple[int, int]]]:
    """chromosome -> [(start, stop)] in file order (the first three BED columns, extract_regions.py:422-426)."""
    out: Dict[str, List[Tuple[int, int]]] = {}
    with open(bedfile) as fh:
        for line in fh:
            f = line.split()
            if len(f) < 3 or line.startswith(("#", "track", "browser")):
                continue
            out.setdefault(f[0], []).append((int(f[1]), int(f[2])))
    return out


def compute_results_from_graph(motif: Motif, graph, regions, debug: bool, args_obj, group=None,
                               always_collective: bool = False) -> Optional[pd.DataFrame]:
    """extract_regions.scan_graph + score_sequences.compute_results in one device pipeline: the rows go
    from the extraction kernel to the score kernel in HBM; only the hits and their metadata come back.
    `graph` / `regions`: one DeviceGraph with its [(S, E)] list, or lists of both (one entry per
    chromosome) -- the q-values are computed over the rows of all of them, like the reference does over
    all TSV files of a motif.
    Under torch.distributed (one process per GPU, every rank calls this with the same arguments and its
    own replica of the graphs) the regions are split over the ranks, the score histogram is all-reduced
    so that q-values stay global, and rank 0 returns the merged table (the others None)."""
    from .scan import KmerScanner
    from .score_sequences import print_scoring_msg
    torch = _torch()
    graphs = list(graph) if isinstance(graph, (list, tuple)) else [graph]
    region_lists = [list(r) for r in regions] if isinstance(graph, (list, tuple)) else [list(regions)]
    dist = torch.distributed
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank(group) if world > 1 or (dist.is_available() and dist.is_initialized()) else 0
    if world > 1:                                    # contiguous shard of the flattened (graph, region) list
        from .distributed import shard_bounds
        flat = [(gi, r) for gi, regs in enumerate(region_lists) for r in regs]
        lo_, hi_ = shard_bounds(len(flat), world, rank)
        region_lists = [[r for gi2, r in flat[lo_:hi_] if gi2 == gi] for gi in range(len(graphs))]
    threshold = float(args_obj.threshold)
    no_qvalue, qval_t = bool(args_obj.noqvalue), bool(args_obj.qvalueT)
    no_reverse, recomb = bool(args_obj.noreverse), bool(args_obj.recomb)
    if rank == 0:
        print_scoring_msg(motif, no_reverse, debug)
    W = motif.width
    parts = [g.extract(r, W) for g, r in zip(graphs, region_lists)]
    labels: List[str] = []
    region_ids = []
    for part in parts:                               # region ids of all chromosomes in one label table
        region_ids.append(part.region + len(labels))
        labels += [part.region_label(r) for r in range(len(part.regions))]
    cat = lambda name: torch.cat([getattr(p_, name) for p_ in parts]) if len(parts) > 1 else getattr(parts[0], name)
    all_kmers, region = cat("kmers"), (torch.cat(region_ids) if len(parts) > 1 else region_ids[0])
    n_all = int(all_kmers.shape[0])
    keep = None
    kmers = all_kmers
    if no_reverse:                                   # '-' rows are skipped before scoring (score_sequences.py:281)
        keep = torch.arange(0, n_all, 2, device=kmers.device)
        kmers = kmers[keep].contiguous()
    n = int(kmers.shape[0])
    n_global = n
    if world > 1 or always_collective:
        tot = torch.tensor([n], dtype=torch.int64, device=kmers.device)
        dist.all_reduce(tot, group=group)
        n_global = int(tot.item())
    if n_global == 0:
        errmsg = "No result retrieved. Unable to proceed.\n"
        errmsg += "\nAre you using the correct VGs and searching on the right chromosomes?\n"
        exception_handler(ValueError, errmsg, debug)
    dm = DeviceMotif.from_motif(motif)
    try:
        if not no_qvalue and rank == 0:
            print("\nComputing q-values...\n")
        # with a process group the scanner all-reduces the histogram before the q-table; hits stay local
        # (their metadata lives on this rank) and travel as finished table rows below
        sc = KmerScanner(dm, max(n, 1), device=kmers.device, side_stream=False, group=group,
                         always_collective=always_collective)
        res = sc.collect(sc.enqueue(kmers, threshold, on_qvalue=qval_t, want_qvalues=not no_qvalue),
                         want_qvalues=not no_qvalue)
        lo, pv = dm.annotate(res["scaled"])
    finally:
        dm.close()
    if rank == 0:
        print(f"Scanned sequences:\t{n_global}")
        print(f"Scanned nucleotides:\t{n_global * W}")
    hit = torch.from_numpy(res["rows"]).to(kmers.device)
    src = keep[hit] if keep is not None else hit
    take = lambda t: t[src].cpu().numpy()
    df = build_frame(
        motif,
        seqnames=[labels[int(r)] for r in take(region)],
        starts=take(cat("start")), stops=take(cat("stop")),
        strands=[chr(c) for c in take(cat("strand"))],
        scores=lo, pvalues=pv,
        qvalues=None if no_qvalue else res["qtable"][res["scaled"]],
        seqs=[bytes(k).decode() for k in take(all_kmers)],
        frequencies=take(cat("freq")),
        references=["ref" if r else "non.ref" for r in take(cat("is_ref"))],
        threshold=None, recomb=recomb,
    )
    if world > 1:
        from .distributed import gather_frames
        return gather_frames(df, group)
    return df
