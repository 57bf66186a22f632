"""Sharded motif scan across the GPUs of one node: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on ROCm).

Regions (TSV files) are independent (the reference already splits files over worker processes,
score_sequences.py:123), so rows shard with no data-path exchange except the one the statistics
need:

  * all-reduce(sum) of the per-motif score histogram  -- BH ranks and n are global; 1000*W+1
    int64 bins (152 KB at W=19): latency-bound on 7 x ~153 GB/s xGMI links, so one flat
    all-reduce per motif, no bucketing;
  * gather of the hit rows to rank 0 for the report (hits are ~1e-4..1e-2 of the rows).

With --no-qvalue and a p-value threshold no collective is needed before the gather.

The collective logic is backend-agnostic (``ScanBackend``): the product backend is the HIP path
(``HipBackend``); the CPU test-suite drives the same orchestration over gloo with a stand-in
backend, which is how the N > 1 path is covered without GPUs.
"""
import glob
import os
from typing import List, Optional, Sequence

import numpy as np
import pandas as pd

from .motif import Motif
from .resultsTmp import build_frame
from .utils import exception_handler

HIT_SCORE_BITS = 20


# ------------------------------------------------------------------------------ sharding
def shard_bounds(n_items: int, world: int, rank: int):
    """Contiguous balanced split [lo, hi) of n_items over `world` ranks (np.array_split rule:
    the first n % world shards get one extra item)."""
    base, extra = divmod(int(n_items), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_files(files: Sequence[str], world: int, rank: int, by_size: bool = True) -> List[str]:
    """Contiguous range of the (sorted) file list for this rank, balanced by file size so that
    every GPU scores about the same number of rows; each region's TSV stays on one device."""
    files = list(files)
    if not by_size or not files:
        lo, hi = shard_bounds(len(files), world, rank)
        return files[lo:hi]
    sizes = np.array([max(os.path.getsize(f), 1) for f in files], dtype=np.float64)
    cum = np.cumsum(sizes)
    total = cum[-1]
    # file i goes to the rank whose share of the total its midpoint falls into
    owner = np.minimum(((cum - sizes / 2) / total * world).astype(int), world - 1)
    return [f for f, o in zip(files, owner) if o == rank]


# ------------------------------------------------------------------------------ backends
class ScanBackend:
    """What the sharded orchestration needs from a scorer.  All arrays are numpy on return;
    `hist` is exchanged as a torch tensor on `self.device` so that the collective runs where
    the data lives."""
    device = None
    L = 0

    def score(self, kmers: np.ndarray):
        """-> (scaled int32[n], hist torch.int64[L] on self.device)"""
        raise NotImplementedError

    def tables(self, hist, threshold: float, on_qvalue: bool):
        """global hist -> (qtable f64[L] numpy, cutoff int, n_rows int)"""
        raise NotImplementedError

    def pvalue_cutoff(self, threshold: float) -> int:
        raise NotImplementedError

    def annotate(self, scaled: np.ndarray):
        """-> (logodds f64, pvalue f64)"""
        raise NotImplementedError

    # ---- the file-level form (compute_results_sharded): one streamed pass over this rank's TSV files
    def begin(self, motifs, files, width, no_reverse, threads, threshold, on_qvalue, want_qvalues):
        """Parse + score `files` for `motifs` (one width) -> a shard scan: .n rows scored on this rank, .device,
        .hist (torch int64 [M, L] on .device: every motif's score histogram, complete), and .finish() -> (list of
        per-motif hit tables -- rows (ids local to this rank's files, ascending), scaled, logodds, pvalue,
        qvalue or None, kmers, start, stop, strand, freq, is_ref, name_id -- and the list of REGION names name_id
        indexes) computed from .hist AS IT IS THEN (the caller all-reduces it in between)."""
        raise NotImplementedError


class HipBackend(ScanBackend):
    """The product backend: DeviceMotif on the current GPU."""

    def __init__(self, motif: Motif, device=None):
        import torch
        from .device import DeviceMotif
        self.torch = torch
        self.device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.dm = DeviceMotif.from_motif(motif)
        self.L = self.dm.L
        self._scores = None

    def score(self, kmers):
        torch = self.torch
        n = int(kmers.shape[0])
        d_k = torch.from_numpy(np.ascontiguousarray(kmers)).to(self.device) if n else \
            torch.empty((0, self.dm.width), dtype=torch.uint8, device=self.device)
        self._scores = torch.empty(n, dtype=torch.int32, device=self.device)
        hist = torch.zeros(self.L, dtype=torch.int64, device=self.device)
        self.dm.score(d_k, self._scores, hist=hist)
        return None, hist   # scores stay on the device; hits are selected there

    @staticmethod
    def score_many(backends, kmers):
        """Same-width motifs over ONE upload and ONE read of the k-mers per group of up to three
        motifs (gfm_score_kmers_multi).  -> hist torch.int64[M, L]; scores stay on the device."""
        from .device import score_multi
        first = backends[0]
        torch = first.torch
        n = int(kmers.shape[0])
        d_k = torch.from_numpy(np.ascontiguousarray(kmers)).to(first.device) if n else \
            torch.empty((0, first.dm.width), dtype=torch.uint8, device=first.device)
        hist = torch.zeros((len(backends), first.L), dtype=torch.int64, device=first.device)
        for b in backends:
            b._scores = torch.empty(n, dtype=torch.int32, device=first.device)
        score_multi([b.dm for b in backends], d_k, [b._scores for b in backends],
                    hists=[hist[j] for j in range(len(backends))])
        return hist

    @staticmethod
    def begin(motifs, files, width, no_reverse, threads, threshold, on_qvalue, want_qvalues):
        """The product's shard scan: gfm_scan_tsv_begin / _finish (score_sequences.StreamScan) over this rank's files,
        the histograms in a torch tensor so that the all-reduce runs where they live."""
        import torch
        from .device import DeviceMotif
        from .score_sequences import StreamScan
        dev = torch.device("cuda", torch.cuda.current_device())
        dms = [DeviceMotif.from_motif(m) for m in motifs]

        class HipShardScan:
            device = dev

            def __init__(self):
                # empty, not zeros: the library zeroes the buffers on ITS score stream before it adds into them; a fill
                # on torch's current stream is ordered against neither and could land behind the first atomics
                self.hist = torch.empty((len(dms), dms[0].L), dtype=torch.int64, device=dev)
                torch.cuda.current_stream(dev).synchronize()     # (the allocator may hand out memory with work pending)
                try:
                    self.scan = StreamScan(dms, files, no_reverse, threads, threshold, on_qvalue, want_qvalues,
                                           hists=[self.hist[j] for j in range(len(dms))], defer=True)
                except Exception:
                    for d in dms:
                        d.close()
                    raise
                self.n = self.scan.n

            def finish(self):
                try:
                    torch.cuda.synchronize(dev)          # the caller's all-reduce of self.hist is complete
                    self.scan.finish()
                    return self.scan.hits, self.scan.names
                finally:
                    for d in dms:
                        d.close()

        return HipShardScan()

    def tables(self, hist, threshold, on_qvalue):
        torch = self.torch
        q = torch.empty(self.L, dtype=torch.float64, device=self.device)
        cut = torch.zeros(1, dtype=torch.int32, device=self.device)
        nrows = torch.zeros(1, dtype=torch.int64, device=self.device)
        self.dm.qvalue_table(hist, threshold, on_qvalue, q, cut, nrows)
        return q.cpu().numpy(), int(cut.item()), int(nrows.item())

    def pvalue_cutoff(self, threshold):
        return self.dm.pvalue_cutoff(threshold)

    def select(self, cutoff: int, row_base: int):
        """rows with score >= cutoff -> (global rows int64, scaled int32), ascending by row"""
        torch = self.torch
        n = int(self._scores.numel())
        cap = max(n, 1)
        hits = torch.zeros(cap + 1, dtype=torch.int64, device=self.device)
        d_cut = torch.tensor([cutoff], dtype=torch.int32, device=self.device)
        self.dm.select_hits(self._scores, d_cut, hits[1:], hits[:1], row_base=row_base, reset_hits=True)
        k = int(hits[0].item())
        packed = np.sort(hits[1:1 + k].cpu().numpy())
        return packed >> HIT_SCORE_BITS, (packed & ((1 << HIT_SCORE_BITS) - 1)).astype(np.int32)

    def annotate(self, scaled):
        return self.dm.annotate(scaled)

    def close(self):
        self.dm.close()


# ------------------------------------------------------------------------------ orchestration
def _dist():
    import torch.distributed as dist
    return dist


def sharded_scan(backend: ScanBackend, kmers: np.ndarray, threshold: float, on_qvalue: bool,
                 want_qvalues: bool, group=None, select=None):
    """One motif over this rank's rows.  Returns dict(rows (global ids), scaled, logodds,
    pvalue[, qvalue], n_scored (global), row_base).  Collectives: all_gather of the row counts
    (global row ids), all_reduce of the histogram when q-values are wanted."""
    if on_qvalue and not want_qvalues:
        raise ValueError("q-value threshold without q-values")
    import torch
    dist = _dist()
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    n_local = int(kmers.shape[0])
    counts = torch.zeros(world, dtype=torch.int64, device=backend.device)
    counts[rank] = n_local
    if world > 1:
        dist.all_reduce(counts, group=group)
    counts = counts.cpu().numpy()
    row_base = int(counts[:rank].sum())
    n_global = int(counts.sum())

    scaled_all, hist = backend.score(kmers)
    qtable = None
    if want_qvalues:
        if world > 1:
            dist.all_reduce(hist, group=group)          # the one data-path exchange
        qtable, cutoff_q, n_hist = backend.tables(hist, threshold, on_qvalue)
        assert n_hist == n_global, (n_hist, n_global)
    cutoff = cutoff_q if on_qvalue else backend.pvalue_cutoff(threshold)
    if select is not None:
        rows, scaled = select(scaled_all, cutoff, row_base)
    else:
        rows, scaled = backend.select(cutoff, row_base)
    lo, pv = backend.annotate(scaled)
    out = dict(rows=rows, scaled=scaled, logodds=lo, pvalue=pv, n_scored=n_global, row_base=row_base,
               shard_bases=np.concatenate([[0], np.cumsum(counts)[:-1]]))
    if want_qvalues:
        out["qvalue"] = qtable[scaled]
    return out


def sharded_scan_same_width(backends: Sequence[ScanBackend], kmers: np.ndarray, threshold: float,
                            on_qvalue: bool, want_qvalues: bool, group=None, score_many=None):
    """Several motifs of ONE width over this rank's rows (BASELINE config 5 on N GPUs).  The k-mers
    are scored once for the whole set (`score_many`, default: the backend class's own -- for
    HipBackend the batched launch that shares one k-mer read between up to three motifs) and the M
    histograms cross the ranks as ONE all-reduce of an [M, L] tensor: xGMI collectives of this
    size are latency-bound, so M motifs cost one latency instead of M.  Returns one dict per motif
    like sharded_scan."""
    if on_qvalue and not want_qvalues:
        raise ValueError("q-value threshold without q-values")
    import torch
    dist = _dist()
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if len({b.L for b in backends}) != 1:
        raise ValueError("sharded_scan_same_width: the motifs must share one width")
    dev = backends[0].device
    counts = torch.zeros(world, dtype=torch.int64, device=dev)
    counts[rank] = int(kmers.shape[0])
    if world > 1:
        dist.all_reduce(counts, group=group)
    counts = counts.cpu().numpy()
    row_base, n_global = int(counts[:rank].sum()), int(counts.sum())

    if score_many is None:
        score_many = getattr(type(backends[0]), "score_many", None)
    scaled_all = [None] * len(backends)
    if score_many is not None:
        hist = score_many(backends, kmers)
    else:
        parts = [b.score(kmers) for b in backends]
        scaled_all = [p[0] for p in parts]
        hist = torch.stack([p[1] for p in parts])
    if want_qvalues and world > 1:
        dist.all_reduce(hist, group=group)              # one exchange for the whole set
    out = []
    for j, b in enumerate(backends):
        qtable = None
        if want_qvalues:
            qtable, cutoff_q, n_hist = b.tables(hist[j], threshold, on_qvalue)
            assert n_hist == n_global, (n_hist, n_global)
        cutoff = cutoff_q if on_qvalue else b.pvalue_cutoff(threshold)
        if scaled_all[j] is not None:
            rows, scaled = b.select_host(scaled_all[j], cutoff, row_base)
        else:
            rows, scaled = b.select(cutoff, row_base)
        lo, pv = b.annotate(scaled)
        res = dict(rows=rows, scaled=scaled, logodds=lo, pvalue=pv, n_scored=n_global, row_base=row_base)
        if want_qvalues:
            res["qvalue"] = qtable[scaled]
        out.append(res)
    return out


def gather_columns(cols, device, group=None):
    """Per-rank column arrays (same names, dtypes and trailing shapes on every rank; any number of rows) ->
    on rank 0 the columns of all ranks concatenated in rank order, elsewhere None.  The rows travel as ONE
    padded byte matrix through torch.distributed.gather (RCCL on the GPUs; all_gather where a backend has no
    gather) after one all_gather of the row counts -- tensors over the interconnect, not pickled DataFrames
    through the store (a p < 1e-2 scan of 1e9 rows has 1e7 hit rows)."""
    import torch
    dist = _dist()
    names = list(cols)
    arrs = [np.ascontiguousarray(cols[n]) for n in names]
    k = int(arrs[0].shape[0]) if arrs else 0
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return {n: a for n, a in zip(names, arrs)}
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    widths = [int(a.dtype.itemsize * int(np.prod(a.shape[1:], dtype=np.int64))) for a in arrs]
    rec = np.empty((k, sum(widths)), dtype=np.uint8)
    at = 0
    for a, w in zip(arrs, widths):
        rec[:, at:at + w] = a.reshape(k, -1).view(np.uint8).reshape(k, w) if k else np.empty((0, w), np.uint8)
        at += w
    counts = torch.zeros(world, dtype=torch.int64, device=device)
    counts[rank] = k
    dist.all_reduce(counts, group=group)
    counts = counts.cpu().numpy()
    kmax = int(counts.max())
    if kmax == 0:
        return {n: a for n, a in zip(names, arrs)} if rank == 0 else None
    pad = torch.zeros((kmax, rec.shape[1]), dtype=torch.uint8, device=device)
    if k:
        pad[:k] = torch.from_numpy(rec).to(device)
    bucket = [torch.empty_like(pad) for _ in range(world)] if rank == 0 else None
    try:
        dist.gather(pad, bucket, dst=0, group=group)
    except (RuntimeError, NotImplementedError):
        bucket = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(bucket, pad, group=group)
    if rank != 0:
        return None
    merged = np.concatenate([bucket[r][:int(counts[r])].cpu().numpy() for r in range(world)], axis=0)
    out, at = {}, 0
    for n, a, w in zip(names, arrs, widths):
        col = np.ascontiguousarray(merged[:, at:at + w]).view(a.dtype).reshape((merged.shape[0],) + a.shape[1:])
        out[n] = col
        at += w
    return out


def gather_names(names: Sequence[str], device, group=None):
    """Every rank's list of strings -> on rank 0 one list per rank (None elsewhere); same transport."""
    blob = np.frombuffer("\n".join(names).encode(), dtype=np.uint8)
    got = gather_columns({"b": blob}, device, group)
    dist = _dist()
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return [list(names)]
    import torch
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    lens = torch.zeros(world, dtype=torch.int64, device=device)
    lens[rank] = len(blob)
    dist.all_reduce(lens, group=group)
    if rank != 0:
        return None
    lens = lens.cpu().numpy()
    raw, at, out = got["b"].tobytes(), 0, []
    for r in range(world):
        part = raw[at:at + int(lens[r])].decode()
        out.append(part.split("\n") if part else [])
        at += int(lens[r])
    return out


def _parse_threads(cores: int, world: int) -> int:
    """Parse threads of one rank: the ranks of a node share its cores (eight ranks x 96 threads would fight over
    256 of them), so every rank takes its share."""
    share = max(1, (os.cpu_count() or 1) // max(1, world))
    return max(1, min(int(cores) if cores and cores > 0 else share, share))


def _scan_width_sharded(motifs, files, width, args_obj, group, backend, debug, reduce=None):
    """One streamed pass per rank over its shard of `files` for `motifs` (one width); histogram all-reduce between
    the scoring phase and the tables; hit rows to rank 0 as packed columns.  -> (per-motif column dicts on rank 0
    / None elsewhere, the merged REGION names on rank 0, rows scored over all ranks).  `reduce`: applied to every rank's
    column dict before it travels (top_hits.compute_top_regions_sharded keeps one hit per region there)."""
    import torch
    dist = _dist()
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    threshold = float(args_obj.threshold)
    no_qvalue, qval_t = bool(args_obj.noqvalue), bool(args_obj.qvalueT)
    no_reverse = bool(args_obj.noreverse)
    mine = shard_files(files, world, rank)
    factory = backend if backend is not None else HipBackend
    from .score_sequences import ScanRetry
    for attempt in range(3):
        scan = factory.begin(motifs, mine, width, no_reverse, _parse_threads(int(args_obj.cores), world), threshold, qval_t,
                             not no_qvalue)
        dev = scan.device
        counts = torch.zeros(world, dtype=torch.int64, device=dev)
        counts[rank] = scan.n
        if world > 1:
            dist.all_reduce(counts, group=group)
        counts = counts.cpu().numpy()
        row_base, n_global = int(counts[:rank].sum()), int(counts.sum())
        if not no_qvalue and world > 1:
            dist.all_reduce(scan.hist, group=group)          # the one data-path exchange: [M, L] in one collective
        # The scan stores no scores: a rank whose hit list was too short has had it grown and must run both phases again --
        # and because the first phase ends in a collective, every rank repeats it or none does.
        again, err = 0, None
        try:
            hits, names = scan.finish()
        except ScanRetry as e:
            again, err = 1, e
        if world > 1:
            flag = torch.tensor([again], dtype=torch.int64, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
            again = int(flag.item())
        if not again:
            break
        if attempt == 2:
            raise err if err is not None else RuntimeError("another rank's hit list kept overflowing")
    if n_global == 0:
        errmsg = "No result retrieved. Unable to proceed.\n"
        errmsg += "\nAre you using the correct VGs and searching on the right chromosomes?\n"
        exception_handler(ValueError, errmsg, debug)
    bases = np.concatenate([[0], np.cumsum(counts)[:-1]]).astype(np.int64)
    name_lists = gather_names(names, dev, group)
    out = []
    for h in hits:
        # hit rows travel to rank 0 as packed columns (one padded tensor gather per motif)
        cols = dict(rows=np.asarray(h.rows, dtype=np.int64) + row_base, scaled=np.asarray(h.scaled, dtype=np.int32),
                    logodds=np.asarray(h.logodds, dtype=np.float64), pvalue=np.asarray(h.pvalue, dtype=np.float64),
                    start=np.asarray(h.start, dtype=np.int64), stop=np.asarray(h.stop, dtype=np.int64),
                    strand=np.asarray(h.strand, dtype=np.uint8), freq=np.asarray(h.freq, dtype=np.int64),
                    is_ref=np.asarray(h.is_ref, dtype=np.uint8), name_id=np.asarray(h.name_id, dtype=np.int32),
                    kmers=np.asarray(h.kmers, dtype=np.uint8).reshape(len(h.rows), width))
        if not no_qvalue:
            cols["qvalue"] = np.asarray(h.qvalue, dtype=np.float64)
        if reduce is not None:
            cols = reduce(cols)
        got = gather_columns(cols, dev, group)
        if rank == 0:
            if world > 1:
                # rows ascend with the rank (contiguous shards): rank of a hit = how many shard bases lie at or below it
                owner = np.searchsorted(bases, got["rows"], side="right") - 1
                shift = np.cumsum([0] + [len(lst) for lst in name_lists])[:-1]
                got["name_ix"] = got["name_id"].astype(np.int64) + shift[owner]
            else:
                got["name_ix"] = got["name_id"].astype(np.int64)
        out.append(got if rank == 0 else None)
    all_names = [n for lst in name_lists for n in lst] if rank == 0 else None
    return out, all_names, n_global


def _frame_from_columns(motif, got, all_names, no_qvalue, recomb):
    names_arr = np.array(all_names, dtype=object)
    return build_frame(
        motif,
        seqnames=list(names_arr[got["name_ix"]]) if len(got["name_ix"]) else [],
        starts=got["start"], stops=got["stop"],
        strands=[chr(c) for c in got["strand"]],
        scores=got["logodds"], pvalues=got["pvalue"],
        qvalues=None if no_qvalue else got["qvalue"],
        seqs=[bytes(k).decode() for k in got["kmers"]],
        frequencies=got["freq"],
        references=["ref" if r else "non.ref" for r in got["is_ref"]],
        threshold=None, recomb=recomb,
    )


def compute_results_sharded(motif: Motif, sequence_loc: str, debug: bool, args_obj, group=None,
                            backend: Optional[ScanBackend] = None) -> Optional[pd.DataFrame]:
    """compute_results (score_sequences.py:44-211) with the TSV files sharded over the ranks of `group`: every rank
    runs the streamed scan (parse threads -> pinned chunks -> score kernel per chunk) over its own files, the score
    histogram is all-reduced between the scoring phase and the q-table, the hit rows are gathered.  Every rank calls
    it; rank 0 gets the report table, the others None.  `backend`: the test seam (an object with ScanBackend.begin);
    None = the HIP path."""
    from .score_sequences import print_scoring_msg
    dist = _dist()
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    no_qvalue, qval_t = bool(args_obj.noqvalue), bool(args_obj.qvalueT)
    if qval_t and no_qvalue:
        raise ValueError("q-value threshold without q-values")
    if backend is None:        # scan_graph's manifest instead of rows: regions AND graph shard over the ranks
        from .extract_regions import compute_results_from_manifest, read_manifest
        manifest = read_manifest(sequence_loc)
        if manifest is not None:
            return compute_results_from_manifest(motif, manifest, debug, args_obj, group=group)
    if rank == 0:
        print_scoring_msg(motif, bool(args_obj.noreverse), debug)
    width = motif.width
    files = sorted(glob.glob(os.path.join(sequence_loc, f"width_{width}", "*.tsv")))
    got, all_names, n_global = _scan_width_sharded([motif], files, width, args_obj, group, backend, debug)
    out = None
    if rank == 0:
        out = _frame_from_columns(motif, got[0], all_names, no_qvalue, bool(args_obj.recomb))
        print(f"Scanned sequences:\t{n_global}")
        print(f"Scanned nucleotides:\t{n_global * width}")
    return out


def compute_results_many_sharded(motifs: Sequence[Motif], sequence_loc: str, debug: bool, args_obj, group=None,
                                 backend: Optional[ScanBackend] = None):
    """score_sequences.compute_results_many under torch.distributed: per width ONE streamed pass per rank for all
    motifs of that width, their M histograms all-reduced as one [M, L] tensor.  Rank 0 gets the tables in the order
    of `motifs`, the others a list of None."""
    from .score_sequences import print_scoring_msg
    dist = _dist()
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    no_qvalue = bool(args_obj.noqvalue)
    if bool(args_obj.qvalueT) and no_qvalue:
        raise ValueError("q-value threshold without q-values")
    if backend is None:        # scan_graph's manifest instead of rows
        from .extract_regions import compute_results_many_from_manifest, read_manifest
        manifest = read_manifest(sequence_loc)
        if manifest is not None:
            return compute_results_many_from_manifest(list(motifs), manifest, debug, args_obj, group=group)
    out = [None] * len(motifs)
    by_width = {}
    for i, m in enumerate(motifs):
        by_width.setdefault(m.width, []).append(i)
    for width, idxs in by_width.items():
        files = sorted(glob.glob(os.path.join(sequence_loc, f"width_{width}", "*.tsv")))
        got, all_names, n_global = _scan_width_sharded([motifs[i] for i in idxs], files, width, args_obj, group, backend,
                                                       debug)
        if rank == 0:
            for i, g in zip(idxs, got):
                print_scoring_msg(motifs[i], bool(args_obj.noreverse), debug)
                print(f"Scanned sequences:\t{n_global}")
                print(f"Scanned nucleotides:\t{n_global * width}")
                out[i] = _frame_from_columns(motifs[i], g, all_names, no_qvalue, bool(args_obj.recomb))
    return out
