"""Device pipeline of one motif scan: score -> histogram -> [all-reduce] -> q-table ->
threshold -> hit list, the numeric core of compute_results (score_sequences.py:44-211).

One process drives one GPU.  With a torch.distributed process group the rows are sharded
across ranks (regions are independent: SURVEY.md section 8e) and the only exchange on the data
path is the all-reduce of the per-motif score histogram (BH ranks are global); hit rows are
gathered to rank 0 for the report.  The collective + table work of one batch runs on a side
stream so that it overlaps the score kernel of the next batch.
"""
from typing import Optional

import numpy as np

from . import _native as _nv
from .device import DeviceMotif, _torch

HIT_SCORE_BITS = 20  # GFM_HIT_SCORE_BITS: hit entry = (row << 20) | scaled score


class ScanSlot:
    """One set of device buffers for a batch of up to n_rows k-mers."""

    def __init__(self, dm: DeviceMotif, n_rows: int, hit_capacity: int, device, scores=None):
        torch = _torch()
        self.n_rows = int(n_rows)
        self.hit_capacity = int(hit_capacity)
        self.scores = scores if scores is not None else torch.empty(self.n_rows, dtype=torch.int32, device=device)
        self.hist = torch.zeros(dm.L, dtype=torch.int64, device=device)
        self.qtable = torch.empty(dm.L, dtype=torch.float64, device=device)
        self.cutoff = torch.zeros(1, dtype=torch.int32, device=device)
        self.nrows = torch.zeros(1, dtype=torch.int64, device=device)
        # [0] = hit count, [1:] = hit rows (one buffer so that one gather moves both)
        self.hits = torch.zeros(self.hit_capacity + 1, dtype=torch.int64, device=device)
        self.best = None      # per-region best (score, row) keys of gfm_region_best (allocated by KmerScanner.set_regions)
        self.cand = None      # [count | candidate entries] of a q-value threshold scan (allocated on first use)
        self.p_cand_count = self.p_cand_rows = None
        self.done = torch.cuda.Event()
        self.tail_done = torch.cuda.Event()
        self.used = False
        self.gathered = None
        # raw addresses for the ctypes fast path
        self.p_scores = self.scores.data_ptr()
        self.p_hist = self.hist.data_ptr()
        self.p_qtable = self.qtable.data_ptr()
        self.p_cutoff = self.cutoff.data_ptr()
        self.p_nrows = self.nrows.data_ptr()
        self.p_hit_count = self.hits.data_ptr()
        self.p_hit_rows = self.hits.data_ptr() + 8

    def candidates(self):
        if self.cand is None:
            self.cand = _torch().zeros_like(self.hits)
            self.p_cand_count = self.cand.data_ptr()
            self.p_cand_rows = self.cand.data_ptr() + 8
        return self.cand

    @property
    def hit_count(self):
        return self.hits[:1]

    @property
    def hit_rows(self):
        return self.hits[1:]


class KmerScanner:
    def __init__(self, dm: DeviceMotif, n_rows: int, hit_capacity: Optional[int] = None,
                 device=None, group=None, n_slots: int = 3, side_stream: bool = True,
                 always_collective: bool = False, gather_group=None, host_paced: Optional[bool] = None,
                 candidates: bool = True, score_buffers: Optional[int] = None):
        torch = _torch()
        self.dm = dm
        self.device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.group = group
        self.world = 1
        self.rank = 0
        if group is not None or (torch.distributed.is_available() and torch.distributed.is_initialized()):
            self.world = torch.distributed.get_world_size(group)
            self.rank = torch.distributed.get_rank(group)
        # always_collective: issue the collectives even in a world of one (exercises the RCCL
        # calls on a single-GPU box)
        self.collective = self.world > 1 or (always_collective and torch.distributed.is_initialized())
        self._gather_ok = True
        # The hit gather is issued from its own stream: the tail stream then goes on with the post kernel
        # and q-table of the next batch instead of idling through the gather.  Collectives of ONE group
        # still execute in issue order (gather(k) before all-reduce(k+1), the same on every rank), so
        # the communication chain per batch is all-reduce + gather while the tail chain is post +
        # all-reduce + q-table.  A second process group (`gather_group`: its own RCCL communicator) would
        # also let gather(k) run beside all-reduce(k+1) -- kernels of two communicators overlapping on one
        # device, which RCCL does not promise to survive: leave it None unless that has been tried on the
        # target node (a one-rank test with it hung once in about forty runs).
        self.gather_group = gather_group if gather_group is not None else group
        self._gather_stream = None
        self.tail_done = None
        if self.collective and side_stream:
            self._gather_stream = torch.cuda.Stream(device=self.device, priority=-1)
        cap = int(hit_capacity) if hit_capacity is not None else int(n_rows)
        # `score_buffers` < n_slots: the int32 [n_rows] score arrays form a shorter ring than the slots (slot k writes array
        # k % score_buffers): a batch's `slot.scores` then stays valid for score_buffers - 1 further enqueues only, while its
        # histogram, q-table and hit list stay valid for n_slots - 1.  What it is for: a fourth slot gives the host a third step of
        # slack against a tail that runs long, but a fourth 80 MB score array pushes the write-through stores' working set
        # (3 x 80 MB fits the 256 MB Infinity Cache, 4 x 80 MB does not: the score kernel 83 -> 87 us, profiles/r06_step_gap.txt).
        nb = n_slots if score_buffers is None else max(1, min(int(score_buffers), n_slots))
        ring = [torch.empty(int(n_rows), dtype=torch.int32, device=self.device) for _ in range(nb)]
        self.slots = [ScanSlot(dm, n_rows, cap, self.device, scores=ring[k % nb]) for k in range(n_slots)]
        self._score_ring = nb
        # high priority: tail kernels are tiny and sit on the critical path of slot reuse
        self.side = torch.cuda.Stream(device=self.device, priority=-1) if side_stream else None
        self._side_p = self.side.cuda_stream if self.side is not None else None
        self._lib = _nv.lib()
        self._cutoffs = {}
        # Slot reuse.  Batch k takes slot k % n_slots, whose previous batch (k - n_slots) must have been
        # consumed by its tail.  With two slots the main stream waits for that on the device (an event wait:
        # a barrier packet in front of the score kernel, several microseconds of an idle GPU per batch).  With
        # three or more slots the HOST waits instead (host_paced): the device still holds n_slots - 1 queued
        # batches meanwhile, and the main stream then carries nothing but score kernels, back to back.
        self.host_paced = (n_slots >= 3) if host_paced is None else bool(host_paced)
        if self.host_paced and n_slots < 2:
            raise ValueError("host-paced slot reuse needs at least two slots")
        # either way batch k - GFM_WORKSPACE_RING is done when batch k is enqueued (n_slots <= the ring, 4): the library's
        # workspace ring is free again without its own event wait.  (The rings belong to the DeviceMotif, not to this
        # scanner: the library honours the flag only while the ring's worth of calls before this one on the handle came from
        # the same stream pair, and orders the reuse itself when another scanner or a batched call was in between.)
        self._reuse_flag = _nv.GFM_FLAG_CALLER_ORDERS_REUSE if n_slots <= _nv.GFM_WORKSPACE_RING else 0
        # q-value threshold: select from the p < t candidates the score kernel collects (enqueue) instead of
        # reading every score again; False = the separate pass over the scores (measurement aid)
        self.candidates = bool(candidates)
        # measurement aid (bench.py tail_ms): close the tail timing of a timed call (gfm_profile_mark_tail) behind the
        # last thing the step enqueues -- one more ctypes call per step, so only on request
        self.profile_tail = False
        self._turn = 0
        # entries of a slot's hit buffer ([count | hits...]) that a gather moves: the whole buffer until
        # size_gather() has seen how many hits a batch really holds
        self.gather_len = cap + 1
        # per-region best hit (set_regions): region ids of a batch's rows, optional haplotype counts, and whether the
        # per-step gather moves the n_regions keys INSTEAD of the hit entries (top_only)
        self._regions = None
        self.top_only = False

    def set_regions(self, region, n_regions: int, freq=None, top_only: bool = False):
        """Per-region best hit beside the hit list (north_star: the top-hit reduction; top_hits.py): every enqueue() then
        also leaves, in slot.best, for each of `n_regions` regions the key of its best row among those that pass the
        batch's threshold (and have freq > 0 if `freq` is given: the rows reported without --recomb).  `region`: int32
        [n_rows] on the device, region id per row of the batches to come (the same layout for every batch).
        top_only: with a process group the per-step gather moves these n_regions keys per rank INSTEAD of every hit
        entry -- what a caller that asks for the top regions only needs (res_writer.py:153-157)."""
        torch = _torch()
        assert region.dtype == torch.int32 and region.is_cuda
        self._regions = (region, int(n_regions), freq)
        self.top_only = bool(top_only)
        for s in self.slots:
            s.best = torch.zeros(int(n_regions), dtype=torch.int64, device=self.device)
            s.gathered = None

    def size_gather(self, margin: float = 1.25, granule: int = 512) -> int:
        """Cut the per-step hit gather down to what is hit: the largest hit count of the finished batches
        over ALL ranks (one MAX all-reduce; call it between batches, e.g. after a warm-up), times `margin`,
        rounded up to `granule` entries (4 KiB).  The fixed-size gather of the whole hit buffer moved
        (world - 1) x capacity x 8 bytes into rank 0 per step -- 17.5 MB at 8 ranks and 2e7-row shards
        against ~1.5 MB of hits.  A later batch with more hits than this raises OverflowError in collect()."""
        torch = _torch()
        self.finish()
        torch.cuda.synchronize(self.device)
        m = torch.stack([s.hits[0] for s in self.slots]).max().reshape(1)
        if self.collective:
            torch.distributed.all_reduce(m, op=torch.distributed.ReduceOp.MAX, group=self.group)
        want = int(int(m.item()) * margin) + 1
        want = -(-want // granule) * granule
        self.gather_len = min(self.slots[0].hit_capacity, want) + 1
        for s in self.slots:
            s.gathered = None
        return self.gather_len

    # ------------------------------------------------------------------ one batch
    def enqueue(self, d_kmers, threshold: float, on_qvalue: bool = False, want_qvalues: bool = True,
                row_base: int = 0, gather_hits: bool = False) -> ScanSlot:
        """Enqueue the whole path for one batch; returns the slot holding its outputs.
        With two slots nothing here synchronises with the host; with three or more (host_paced) the host
        waits for the batch n_slots back, which the device finished long ago unless the host runs more than
        n_slots - 1 batches ahead.  Kept lean on purpose: a step is ~100 us of GPU
        time, so the host side (ctypes calls with cached raw pointers, no torch dispatch unless a
        collective is needed) must stay well below that or the GPU starves."""
        if on_qvalue and not want_qvalues:      # the reference asserts this in ResultTmp.to_df
            raise ValueError("q-value threshold without q-values")
        torch = _torch()
        lib = self._lib
        dm = self.dm
        slot = self.slots[self._turn % len(self.slots)]
        self._turn += 1
        main = torch.cuda.current_stream(self.device)
        main_p = main.cuda_stream
        tail = self.side if self.side is not None else main
        tail_p = self._side_p if self.side is not None else main_p
        if slot.used:                        # the slot's previous batch has been consumed
            if self.host_paced:
                slot.done.synchronize()
            else:
                main.wait_event(slot.done)
        slot.used = True
        if self._score_ring < len(self.slots) and (on_qvalue or self._regions is not None):
            # a shorter score ring, and this batch's tail READS the scores (selection on q, per-region best hit): the batch that
            # wrote this score array last must be through its tail before the score kernel overwrites it
            prev = self.slots[(self._turn - 1 - self._score_ring) % len(self.slots)]
            if prev.used and self._turn > self._score_ring:
                prev.done.synchronize()
        # no zeroing on the critical path: the q-value kernel hands the histogram back cleared
        # (GFM_FLAG_CLEAR_HIST) and the hit list restarts through GFM_FLAG_RESET_HITS
        n = int(d_kmers.shape[0])
        kp = d_kmers.data_ptr() if n else None
        h = dm.handle
        # the score kernel goes to the main stream; the library puts its post kernel (histogram
        # slabs -> slot.hist, residual hits -> slot.hits) on the tail stream behind an event
        use_cand = on_qvalue and self.candidates
        if not on_qvalue:
            key = float(threshold)
            cut = self._cutoffs.get(key)
            if cut is None:                   # host lookup in p_table, known before scoring
                cut = self._cutoffs[key] = dm.pvalue_cutoff(key)
            _nv.check(lib.gfm_score_kmers(h, kp, n, slot.p_scores, slot.p_hist if want_qvalues else None,
                                          cut, int(row_base), slot.p_hit_rows, slot.hit_capacity,
                                          slot.p_hit_count, _nv.GFM_FLAG_RESET_HITS | self._reuse_flag, main_p, tail_p))
        elif not use_cand:
            _nv.check(lib.gfm_score_kmers(h, kp, n, slot.p_scores, slot.p_hist, _nv.GFM_NO_SELECT, 0,
                                          None, 0, None, self._reuse_flag, main_p, tail_p))
        else:
            # q-value threshold: the cutoff needs the global histogram, but q >= p, so the rows with q < t are
            # among those with p < t -- the score kernel collects THOSE on the fly (candidates), and the
            # selection behind the q-table filters the candidate list instead of reading every score again
            # (1e8 rows: 400 MB on the tail stream, beside the next score kernel).  A candidate list that
            # overflows is noticed on the device and the scores are read after all (gfm_select_hits_from).
            key = float(threshold)
            cut = self._cutoffs.get(key)
            if cut is None:
                cut = self._cutoffs[key] = dm.pvalue_cutoff(key)
            slot.candidates()
            _nv.check(lib.gfm_score_kmers(h, kp, n, slot.p_scores, slot.p_hist, cut, int(row_base),
                                          slot.p_cand_rows, slot.hit_capacity, slot.p_cand_count,
                                          _nv.GFM_FLAG_RESET_HITS | self._reuse_flag, main_p, tail_p))
        if want_qvalues:
            if self.collective:
                with torch.cuda.stream(tail):
                    torch.distributed.all_reduce(slot.hist, group=self.group)
            _nv.check(lib.gfm_qvalue_table(h, slot.p_hist, float(threshold), int(bool(on_qvalue)),
                                           slot.p_qtable, slot.p_cutoff, slot.p_nrows,
                                           _nv.GFM_FLAG_CLEAR_HIST, tail_p))
        if on_qvalue and not use_cand:
            _nv.check(lib.gfm_select_hits(h, slot.p_scores, n, slot.p_cutoff, int(row_base),
                                          slot.p_hit_rows, slot.hit_capacity, slot.p_hit_count,
                                          _nv.GFM_FLAG_RESET_HITS, tail_p))
        elif on_qvalue:
            _nv.check(lib.gfm_select_hits_from(h, slot.p_scores, n, slot.p_cutoff, int(row_base),
                                               slot.p_cand_rows, slot.hit_capacity, slot.p_cand_count,
                                               slot.p_hit_rows, slot.hit_capacity, slot.p_hit_count, tail_p))
        if self._regions is not None:
            region, n_regions, freq = self._regions
            with torch.cuda.stream(tail):
                slot.best.zero_()
            # rows under the batch's threshold: the p-value cutoff is known on the host, the q-value cutoff lives in slot.cutoff
            _nv.check(lib.gfm_region_best(slot.p_scores, n, region.data_ptr(), n_regions,
                                          freq.data_ptr() if freq is not None else None,
                                          0 if on_qvalue else self._cutoffs[float(threshold)],
                                          slot.p_cutoff if on_qvalue else None, int(row_base), slot.best.data_ptr(), tail_p))
        if gather_hits and self.collective:
            gs = self._gather_stream
            if gs is not None:                 # gather on its own stream, behind this step's tail
                slot.tail_done.record(tail)
                gs.wait_event(slot.tail_done)
                with torch.cuda.stream(gs):
                    self._gather(slot)
                if self.profile_tail:
                    lib.gfm_profile_mark_tail(h, gs.cuda_stream)
                slot.done.record(gs)
                return slot
            with torch.cuda.stream(tail):
                self._gather(slot)
        if self.profile_tail:
            lib.gfm_profile_mark_tail(h, tail_p)
        slot.done.record(tail)
        return slot

    def _gather(self, slot: ScanSlot):
        """hit buffers ([count | entries], fixed size) of all ranks -> rank 0.  gather moves
        (world-1) buffers into rank 0 only; if the backend lacks it, all_gather is the fallback."""
        torch = _torch()
        dist = torch.distributed
        # [count | the first gather_len - 1 hit entries] -- or, for a caller that wants the top regions only, the n_regions
        # best-hit keys: the top-hit-only gather
        part = slot.best if (self.top_only and slot.best is not None) else slot.hits[:self.gather_len]
        if slot.gathered is None and (self.rank == 0 or not self._gather_ok):
            slot.gathered = [torch.empty_like(part) for _ in range(self.world)]
        if self._gather_ok:
            try:
                dist.gather(part, slot.gathered if self.rank == 0 else None, dst=0,
                            group=self.gather_group)
                return
            except (RuntimeError, NotImplementedError):
                self._gather_ok = False
                if slot.gathered is None:
                    slot.gathered = [torch.empty_like(part) for _ in range(self.world)]
        dist.all_gather(slot.gathered, part, group=self.gather_group)

    def finish(self):
        """Make the caller's stream wait for all side-stream work."""
        torch = _torch()
        main = torch.cuda.current_stream(self.device)
        for s in self.slots:
            main.wait_event(s.done)

    # ------------------------------------------------------------------ results of one slot
    def collect(self, slot: ScanSlot, want_qvalues: bool = True):
        """D2H of one finished batch: hit rows ascending with scaled score, log-odds,
        p-value and q-value.  On rank 0 of a sharded scan the gathered hits of all ranks
        are merged (rows are global ids through row_base)."""
        torch = _torch()
        slot.done.synchronize()
        if self.top_only and slot.best is not None:      # per-region keys: the ranks' tables merged by maximum (regions are sharded,
            keys = slot.best                             # so at most one rank holds a key for a region)
            if slot.gathered is not None and self.rank == 0:
                keys = torch.stack(slot.gathered).max(dim=0).values
            out = {"best": keys.cpu().numpy(), "n_scored": int(slot.nrows.item()) if want_qvalues else None}
            if want_qvalues:
                out["qtable"] = slot.qtable.cpu().numpy()
            return out
        bufs = [slot.hits]
        if slot.gathered is not None and self.rank == 0:
            bufs = slot.gathered
        packed_all = []
        for b in bufs:
            k = int(b[0].item())
            if k > int(b.numel()) - 1:
                raise OverflowError(f"{k} hits exceed the {int(b.numel()) - 1} entries the slot holds / gathers")
            packed_all.append(b[1:1 + k].cpu().numpy())
        packed = np.sort(np.concatenate(packed_all)) if packed_all else np.empty(0, np.int64)
        rows = packed >> HIT_SCORE_BITS
        scaled = (packed & ((1 << HIT_SCORE_BITS) - 1)).astype(np.int32)
        out = {"rows": rows, "scaled": scaled, "n_scored": int(slot.nrows.item()) if want_qvalues else None}
        if want_qvalues:
            out["qtable"] = slot.qtable.cpu().numpy()
        if slot.best is not None:
            out["best"] = slot.best.cpu().numpy()
        return out


class SameWidthScanner:
    """Resident buffers for repeated batched scans of several motifs of ONE width (BASELINE config 5):
    the k-mer matrix is read once per group of up to three motifs (gfm_score_kmers_multi); with a process
    group the M histograms cross the ranks as ONE all-reduce of an [M, L] tensor; q-tables (and the
    separate selection of --qvalueT) run per motif.  Nothing here synchronises with the host."""

    def __init__(self, dms, n_rows: int, hit_capacity: int, device, group=None, always_collective: bool = False,
                 side_stream: bool = False):
        torch = _torch()
        self.dms, self.device, self.group = list(dms), device, group
        # side_stream: the per-motif tail (all-reduce, q-table kernels, the separate selection of --qvalueT) goes
        # to a side stream behind ONE event per enqueue and runs beside the score kernels of the next enqueue.
        # Off by default: measured at fifty motifs (BASELINE config 5, one MI355X) the chain of small
        # latency-bound kernels, 3-4x slower each while a score kernel saturates HBM, became the critical
        # path (step 13.5 ms against 10.6 ms in stream order) and the score kernels lost 7 % to it.
        # Kept as an option for N > 1, where the tail holds a collective (not measured: no multi-GPU box).
        self.side = torch.cuda.Stream(device=device, priority=-1) if side_stream else None
        self.scored = torch.cuda.Event()
        self.done = torch.cuda.Event()
        self.used = False
        M, L = len(self.dms), self.dms[0].L
        if any(d.L != L for d in self.dms):
            raise ValueError("SameWidthScanner: the motifs must share one width")
        dist = torch.distributed
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.collective = self.world > 1 or (always_collective and dist.is_initialized())
        self.scores = [torch.empty(n_rows, dtype=torch.int32, device=device) for _ in range(M)]
        self.hist = torch.zeros((M, L), dtype=torch.int64, device=device)
        self.qtable = torch.empty((M, L), dtype=torch.float64, device=device)
        self.cutoff = torch.zeros(M, dtype=torch.int32, device=device)
        self.nrows = torch.zeros(M, dtype=torch.int64, device=device)
        self.hits = torch.zeros((M, int(hit_capacity) + 1), dtype=torch.int64, device=device)
        self.cand = None      # candidate lists of a q-value threshold scan (allocated on first use)
        self._cuts = {}

    def enqueue(self, d_kmers, threshold: float, on_qvalue: bool = False, want_qvalues: bool = True,
                row_base: int = 0):
        from .device import qvalue_table_multi, score_multi
        torch = _torch()
        if on_qvalue and not want_qvalues:
            raise ValueError("q-value threshold without q-values")
        M = len(self.dms)
        hists = [self.hist[j] for j in range(M)] if want_qvalues else None
        main = torch.cuda.current_stream(self.device)
        tail = self.side if self.side is not None else main
        if self.used and tail is not main:
            main.wait_event(self.done)           # the previous enqueue's tail has let go of the buffers
        self.used = True
        if not on_qvalue:
            cuts = self._cuts.get(float(threshold))
            if cuts is None:
                cuts = self._cuts[float(threshold)] = [d.pvalue_cutoff(threshold) for d in self.dms]
            score_multi(self.dms, d_kmers, self.scores, hists=hists, cutoffs=cuts, row_base=row_base,
                        hit_rows=[self.hits[j, 1:] for j in range(M)], hit_counts=[self.hits[j, :1] for j in range(M)],
                        reset_hits=True, stream=main)
        else:       # q-value threshold: collect the p < t candidates while scoring (see KmerScanner.enqueue)
            cuts = self._cuts.get(float(threshold))
            if cuts is None:
                cuts = self._cuts[float(threshold)] = [d.pvalue_cutoff(threshold) for d in self.dms]
            if self.cand is None:
                self.cand = torch.zeros_like(self.hits)
            score_multi(self.dms, d_kmers, self.scores, hists=hists, cutoffs=cuts, row_base=row_base,
                        hit_rows=[self.cand[j, 1:] for j in range(M)], hit_counts=[self.cand[j, :1] for j in range(M)],
                        reset_hits=True, stream=main)
        if not (want_qvalues or on_qvalue):
            return
        if tail is not main:
            self.scored.record(main)
            tail.wait_event(self.scored)
        if want_qvalues:
            if self.collective:
                with torch.cuda.stream(tail):
                    torch.distributed.all_reduce(self.hist, group=self.group)   # one exchange for the whole set
            qvalue_table_multi(self.dms, [self.hist[j] for j in range(M)], threshold, on_qvalue,
                               [self.qtable[j] for j in range(M)], [self.cutoff[j:j + 1] for j in range(M)],
                               [self.nrows[j:j + 1] for j in range(M)], stream=tail, clear_hist=True)
        if on_qvalue:
            n = int(d_kmers.shape[0])
            for j, d in enumerate(self.dms):
                d.select_hits_from(self.scores[j][:n], self.cutoff[j:j + 1], self.cand[j, 1:], self.cand[j, :1],
                                   self.hits[j, 1:], self.hits[j, :1], row_base=row_base, stream=tail)
        if tail is not main:
            self.done.record(tail)

    def finish(self):
        """Make the caller's stream wait for the side-stream work of the last enqueue."""
        if self.used and self.side is not None:
            _torch().cuda.current_stream(self.device).wait_event(self.done)


def scan_same_width(motifs, d_kmers, threshold: float, on_qvalue: bool = False,
                    want_qvalues: bool = True, row_base: int = 0, hit_capacity: Optional[int] = None):
    """Several motifs of ONE width over one device-resident k-mer matrix (BASELINE config 5): the
    batched launch reads the k-mers once per group of up to three motifs; q-tables and selection
    then run per motif.  Returns one dict per motif like KmerScanner.collect()."""
    if on_qvalue and not want_qvalues:
        raise ValueError("q-value threshold without q-values")
    torch = _torch()
    from .device import qvalue_table_multi, score_multi
    n = int(d_kmers.shape[0])
    dev = d_kmers.device
    cap = int(hit_capacity) if hit_capacity is not None else n
    scores = [torch.empty(n, dtype=torch.int32, device=dev) for _ in motifs]
    hists = [torch.zeros(m.L, dtype=torch.int64, device=dev) for m in motifs] if want_qvalues else None
    hits = [torch.zeros(cap + 1, dtype=torch.int64, device=dev) for _ in motifs]
    qtabs = [torch.empty(m.L, dtype=torch.float64, device=dev) for m in motifs] if want_qvalues else None
    cut_d = [torch.zeros(1, dtype=torch.int32, device=dev) for _ in motifs]
    nrows = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in motifs]
    if not on_qvalue:
        cuts = [m.pvalue_cutoff(threshold) for m in motifs]
        score_multi(motifs, d_kmers, scores, hists=hists, cutoffs=cuts, row_base=row_base,
                    hit_rows=[h[1:] for h in hits], hit_counts=[h[:1] for h in hits], reset_hits=True)
    else:       # q-value threshold: the p < t candidates are collected while scoring (q >= p)
        cuts = [m.pvalue_cutoff(threshold) for m in motifs]
        cand = [torch.zeros(cap + 1, dtype=torch.int64, device=dev) for _ in motifs]
        score_multi(motifs, d_kmers, scores, hists=hists, cutoffs=cuts, row_base=row_base,
                    hit_rows=[c[1:] for c in cand], hit_counts=[c[:1] for c in cand], reset_hits=True)
    out = []
    if want_qvalues:
        qvalue_table_multi(motifs, hists, threshold, on_qvalue, qtabs, cut_d, nrows)
    for j, m in enumerate(motifs):
        if on_qvalue:
            m.select_hits_from(scores[j], cut_d[j], cand[j][1:], cand[j][:1], hits[j][1:], hits[j][:1],
                               row_base=row_base)
    torch.cuda.synchronize(dev)
    for j, m in enumerate(motifs):
        k = int(hits[j][0].item())
        if k > cap:
            raise OverflowError(f"{k} hits exceed the capacity {cap}")
        packed = np.sort(hits[j][1:1 + k].cpu().numpy())
        res = {"rows": packed >> HIT_SCORE_BITS,
               "scaled": (packed & ((1 << HIT_SCORE_BITS) - 1)).astype(np.int32),
               "n_scored": int(nrows[j].item()) if want_qvalues else None,
               "scores": scores[j]}
        if want_qvalues:
            res["qtable"] = qtabs[j].cpu().numpy()
        out.append(res)
    return out
