"""Device-side objects: a motif resident in HBM and the scoring pipeline around it.

Everything numeric is done by libgrafimo_hip.so; torch is used only for device
buffers, streams and (in distributed.py) torch.distributed over RCCL.
"""
import ctypes
from collections import OrderedDict
from typing import Optional

import numpy as np

from . import _native as nv

RANGE = nv.RANGE


def _torch():
    import torch  # deferred: importing grafimo_amd must stay cheap and GPU-free
    return torch


_FP_WEIGHTS = {}


def _fingerprint(pmf) -> tuple:
    """A content key for a motif's score distribution (f64 [L], 152 KB at W = 19) in ~25 us: length, the XOR of its 64-bit
    words, and their sum weighted by position (odd weights, arithmetic mod 2^64: entries that trade places change it).
    (hash(bytes) of it was 50-100 us of every compute_results call for a Motif that carries its pval_matrix -- GRAFIMO's
    always does, motif_ops.py:1021-1022 -- a quarter of a 0.3 ms call.)"""
    words = np.ascontiguousarray(pmf, dtype=np.float64).view(np.uint64)
    n = len(words)
    if not n:
        return (0, 0, 0)
    w = _FP_WEIGHTS.get(n)
    if w is None:
        if len(_FP_WEIGHTS) > 64:
            _FP_WEIGHTS.clear()
        w = _FP_WEIGHTS[n] = np.arange(n, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(1) | np.uint64(1)
    return (n, int(np.bitwise_xor.reduce(words)), int((words * w).sum(dtype=np.uint64)))


class DeviceMotif:
    """Numeric content of a ``Motif`` on the current HIP device (gfm_motif_t).

    score_matrix int [4,W] rows A,C,G,T; bg f64[4]; pmf (``pval_matrix``) optional --
    when omitted the Staden DP of comp_pval_mat (motif_processing.pyx:552-603) runs on
    the device.
    """

    def __init__(self, score_matrix, bg, min_val, scale, offset, pmf: Optional[np.ndarray] = None):
        sm = np.ascontiguousarray(score_matrix, dtype=np.int64)
        if sm.ndim != 2 or sm.shape[0] != 4:
            raise ValueError("score_matrix must be [4, W]")
        bg = np.ascontiguousarray(bg, dtype=np.float64)
        self.width = int(sm.shape[1])
        self.L = RANGE * self.width + 1
        self.min_val = int(min_val)
        self.scale = int(scale)
        self.offset = float(offset)
        if pmf is not None:
            pmf = np.ascontiguousarray(pmf, dtype=np.float64)
            if pmf.shape != (self.L,):
                raise ValueError(f"pval_matrix must have {self.L} entries")
        h = ctypes.c_void_p()
        nv.check(nv.lib().gfm_motif_create(nv.ptr(sm), self.width, nv.ptr(bg), self.min_val,
                                           self.scale, self.offset, nv.ptr(pmf), ctypes.byref(h)))
        self._h = h
        lo, hi = ctypes.c_int32(), ctypes.c_int32()
        nv.check(nv.lib().gfm_motif_score_range(self._h, ctypes.byref(lo), ctypes.byref(hi)))
        self.score_lo, self.score_hi = lo.value, hi.value

    @classmethod
    def from_motif(cls, motif, use_motif_pmf=True):
        pmf = None
        if use_motif_pmf:
            try:
                pmf = motif.pval_matrix
            except AttributeError:     # not computed yet: the DP runs on the device
                pmf = None
        from .motif import dense_bg, dense_score_matrix
        return cls(dense_score_matrix(motif), dense_bg(motif), int(motif.min_val), int(motif.scale),
                   float(motif.offset), pmf)

    def close(self):
        if getattr(self, "_h", None):
            nv.lib().gfm_motif_destroy(self._h)
            self._h = None

    # ---- kept handles for the entry points that are called once per motif AND once more per chromosome / region set
    # (compute_results, compute_results_from_graph): creating the handle -- one device allocation, the pair tables, the
    # DP or the upload of its result, the tail table -- and destroying it again was 0.65 ms of a 3 ms call
    _kept = OrderedDict()          # key -> [DeviceMotif, leases]
    # handles nobody holds that are kept for the next call, least recently used out first.  64: a motif set is scored motif after
    # motif and again (grafimo.findmotif per chromosome set; BASELINE configs[4] has 50 PWMs) -- with 8, every call of a 50-motif
    # set created and destroyed 50 handles, 0.5 ms each, a quarter of the call.  A handle holds ~80 MB of scoring workspace;
    # GRAFIMO_MOTIF_CACHE sets another number, DeviceMotif.drop_kept() frees them.
    _KEEP = max(1, int(__import__("os").environ.get("GRAFIMO_MOTIF_CACHE", 64)))

    @classmethod
    def lease(cls, motif, use_motif_pmf=True) -> "DeviceMotif":
        """A handle for `motif` on the current device, a kept one if the same numbers were leased before.  Give it
        back with `release()`, not `close()`."""
        from .motif import dense_bg, dense_score_matrix
        pmf = getattr(motif, "pval_matrix", None) if use_motif_pmf else None
        sm, bg = dense_score_matrix(motif), dense_bg(motif)
        key = (_torch().cuda.current_device(), np.ascontiguousarray(sm, dtype=np.int64).tobytes(),
               np.ascontiguousarray(bg, dtype=np.float64).tobytes(), int(motif.min_val), int(motif.scale), float(motif.offset),
               None if pmf is None else _fingerprint(pmf))
        slot = cls._kept.get(key)
        if slot is not None and slot[0]._h:
            slot[1] += 1
            cls._kept.move_to_end(key)
            return slot[0]
        dm = cls(sm, bg, int(motif.min_val), int(motif.scale), float(motif.offset), pmf)
        dm._lease_key = key
        cls._kept[key] = [dm, 1]
        for old in [k for k, v in cls._kept.items() if v[1] == 0][:max(0, len(cls._kept) - cls._KEEP)]:
            cls._kept.pop(old)[0].close()
        return dm

    def release(self):
        slot = DeviceMotif._kept.get(getattr(self, "_lease_key", None))
        if slot is None or slot[0] is not self:
            self.close()
        else:
            slot[1] = max(0, slot[1] - 1)

    @classmethod
    def drop_kept(cls):
        """Destroy every kept handle nobody holds (tests; before a fork that will use the GPU itself)."""
        for k in [k for k, v in cls._kept.items() if v[1] == 0]:
            cls._kept.pop(k)[0].close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    # ---- tables
    def tables(self):
        """(pmf, p_table) as host f64[L] arrays."""
        pmf = np.empty(self.L, dtype=np.float64)
        pt = np.empty(self.L, dtype=np.float64)
        nv.check(nv.lib().gfm_motif_tables(self._h, nv.ptr(pmf), nv.ptr(pt)))
        return pmf, pt

    def ptable_host(self) -> np.ndarray:
        """p_table f64 [L] on the host (read once per handle): p-value of a scaled score = one lookup."""
        pt = getattr(self, "_ptable_host", None)
        if pt is None:
            pt = self._ptable_host = self.tables()[1]
        return pt

    def pvalue_cutoff(self, threshold: float) -> int:
        c = ctypes.c_int32()
        nv.check(nv.lib().gfm_motif_pvalue_cutoff(self._h, float(threshold), ctypes.byref(c)))
        return c.value

    def annotate(self, scaled_scores):
        """scaled int scores -> (log-odds f64, p-value f64) (score_sequences.py:390-393)."""
        s = np.ascontiguousarray(scaled_scores, dtype=np.int32)
        lo = np.empty(len(s), dtype=np.float64)
        pv = np.empty(len(s), dtype=np.float64)
        nv.check(nv.lib().gfm_motif_annotate(self._h, nv.ptr(s), len(s), nv.ptr(lo), nv.ptr(pv)))
        return lo, pv

    def fused_workspace(self, device):
        """int64 [2 L + 2] on `device`, kept with the handle: score histogram | q-table (viewed as f64) | cutoff, row count
        -- what one compute_results_from_graph call needs beside the graphs' own buffers (no allocation per call)."""
        w = getattr(self, "_fused_ws", None)
        if w is None or w.device != device:
            w = self._fused_ws = _torch().zeros(2 * self.L + 2, dtype=_torch().int64, device=device)
        return w

    def fused_views(self, device):
        """(histogram int64 [L], q-table float64 [L], cutoff int32 [1]) -- views of fused_workspace, made once per handle (three
        slicing / view calls per motif and call were ~12 us of a 350 us call)."""
        v = getattr(self, "_fused_views", None)
        w = self.fused_workspace(device)
        if v is None or v[3] is not w:
            torch, L = _torch(), self.L
            v = self._fused_views = (w[:L], w[L:2 * L].view(torch.float64), w[2 * L:2 * L + 1].view(torch.int32)[:1], w)
        return v[:3]

    # ---- device-pointer entry points (torch tensors as buffers)
    def score(self, kmers, scores, hist=None, select_cutoff=None, row_base=0, hit_rows=None,
              hit_count=None, stream=None, reset_hits=False, tail_stream=None):
        """Enqueue gfm_score_kmers.  kmers uint8 [n,W] (cuda), scores int32 [n] or None (no score is stored: histogram
        and hits only), hist int64 [L] (accumulated), hit_rows int64 [cap], hit_count int64 [1]."""
        n = int(kmers.shape[0])
        assert kmers.is_contiguous() and kmers.dtype == _torch().uint8
        assert n == 0 or kmers.shape[1] == self.width
        cut = nv.GFM_NO_SELECT if select_cutoff is None else int(select_cutoff)
        nv.check(nv.lib().gfm_score_kmers(
            self._h, kmers.data_ptr() if n else None, n, scores.data_ptr() if (n and scores is not None) else None,
            hist.data_ptr() if hist is not None else None, cut, int(row_base),
            hit_rows.data_ptr() if hit_rows is not None else None,
            int(hit_rows.numel()) if hit_rows is not None else 0,
            hit_count.data_ptr() if hit_count is not None else None,
            nv.GFM_FLAG_RESET_HITS if reset_hits else 0, _stream_ptr(stream),
            _stream_ptr(tail_stream) if tail_stream is not None else None))

    def qvalue_table(self, hist, threshold, on_qvalue, qtable=None, cutoff=None, nrows=None,
                     stream=None, clear_hist=False):
        nv.check(nv.lib().gfm_qvalue_table(
            self._h, hist.data_ptr(), float(threshold), int(bool(on_qvalue)),
            qtable.data_ptr() if qtable is not None else None,
            cutoff.data_ptr() if cutoff is not None else None,
            nrows.data_ptr() if nrows is not None else None,
            nv.GFM_FLAG_CLEAR_HIST if clear_hist else 0, _stream_ptr(stream)))

    def select_hits(self, scores, cutoff, hit_rows, hit_count, row_base=0, stream=None,
                    reset_hits=False):
        n = int(scores.numel())
        nv.check(nv.lib().gfm_select_hits(self._h, scores.data_ptr() if n else None, n, cutoff.data_ptr(),
                                          int(row_base), hit_rows.data_ptr(), int(hit_rows.numel()),
                                          hit_count.data_ptr(),
                                          nv.GFM_FLAG_RESET_HITS if reset_hits else 0,
                                          _stream_ptr(stream)))

    def select_hits_from(self, scores, cutoff, cand_rows, cand_count, hit_rows, hit_count, row_base=0, stream=None):
        """gfm_select_hits_from: the rows with score >= *cutoff, read from a candidate list (the fused selection
        at a lower cutoff) when it is complete, else from the scores."""
        n = int(scores.numel())
        nv.check(nv.lib().gfm_select_hits_from(self._h, scores.data_ptr() if n else None, n, cutoff.data_ptr(),
                                               int(row_base), cand_rows.data_ptr(), int(cand_rows.numel()),
                                               cand_count.data_ptr(), hit_rows.data_ptr(), int(hit_rows.numel()),
                                               hit_count.data_ptr(), _stream_ptr(stream)))

    # ---- measurement aid (bench.py)
    def profile_enable(self, slots: int, every: int = 1):
        nv.check(nv.lib().gfm_profile_enable(self._h, int(slots), int(every)))

    def profile_read(self, capacity: int = 4096):
        ms = np.empty(capacity, dtype=np.float32)
        n = ctypes.c_int(0)
        nv.check(nv.lib().gfm_profile_read(self._h, nv.ptr(ms), capacity, ctypes.byref(n)))
        return ms[: n.value].copy()

    def profile_mark_tail(self, stream=None):
        """gfm_profile_mark_tail: stop event of the tail of the last call if that call was a timed one."""
        nv.check(nv.lib().gfm_profile_mark_tail(self._h, _stream_ptr(stream)))

    def profile_read_tail(self, capacity: int = 4096):
        ms = np.empty(capacity, dtype=np.float32)
        n = ctypes.c_int(0)
        nv.check(nv.lib().gfm_profile_read_tail(self._h, nv.ptr(ms), capacity, ctypes.byref(n)))
        return ms[: n.value].copy()

    # ---- host one-call form
    def scan_host(self, kmers: np.ndarray, threshold: float, on_qvalue=False, want_qvalues=True,
                  capacity: Optional[int] = None):
        """gfm_scan_host: returns dict(rows, scaled, logodds, pvalue[, qvalue]) for the hits,
        ascending by row."""
        kmers = np.ascontiguousarray(kmers, dtype=np.uint8)
        n = int(kmers.shape[0])
        if n and kmers.shape[1] != self.width:
            raise ValueError("k-mer width differs from the motif width")
        cap = n if capacity is None else int(capacity)
        while True:
            rows = np.empty(cap, dtype=np.int64)
            sc = np.empty(cap, dtype=np.int32)
            lo = np.empty(cap, dtype=np.float64)
            pv = np.empty(cap, dtype=np.float64)
            qv = np.empty(cap, dtype=np.float64) if want_qvalues else None
            nh = ctypes.c_int64(0)
            rc = nv.lib().gfm_scan_host(self._h, nv.ptr(kmers) if n else None, n, float(threshold),
                                        int(bool(on_qvalue)), int(bool(want_qvalues)), cap,
                                        nv.ptr(rows), nv.ptr(sc), nv.ptr(lo), nv.ptr(pv),
                                        nv.ptr(qv), ctypes.byref(nh))
            if rc == nv.GFM_ERR_OVERFLOW and nh.value > cap:
                cap = int(nh.value)
                continue
            nv.check(rc)
            break
        k = nh.value
        out = dict(rows=rows[:k], scaled=sc[:k], logodds=lo[:k], pvalue=pv[:k])
        if want_qvalues:
            out["qvalue"] = qv[:k]
        return out


def score_multi(motifs, kmers, scores, hists=None, cutoffs=None, row_base=0, hit_rows=None,
                hit_counts=None, stream=None, reset_hits=False):
    """gfm_score_kmers_multi: several DeviceMotifs of one width over one k-mer matrix.
    scores / hists / hit_rows / hit_counts: lists of torch tensors (entries may be None where the
    scalar call allows it); cutoffs: list of ints or None."""
    M = len(motifs)
    n = int(kmers.shape[0])
    vp = ctypes.c_void_p

    def ptr_array(tensors):
        if tensors is None:
            return None
        return (vp * M)(*[(t.data_ptr() if t is not None else None) for t in tensors])

    handles = (vp * M)(*[m.handle for m in motifs])
    cuts = None
    if cutoffs is not None:
        cuts = (ctypes.c_int32 * M)(*[nv.GFM_NO_SELECT if c is None else int(c) for c in cutoffs])
    caps = None
    if hit_rows is not None:
        caps = (ctypes.c_int64 * M)(*[(int(t.numel()) if t is not None else 0) for t in hit_rows])
    nv.check(nv.lib().gfm_score_kmers_multi(
        handles, M, kmers.data_ptr() if n else None, n, ptr_array(scores), ptr_array(hists), cuts,
        int(row_base), ptr_array(hit_rows), caps, ptr_array(hit_counts),
        nv.GFM_FLAG_RESET_HITS if reset_hits else 0, _stream_ptr(stream)))


def multi_plan(motifs, with_hist=None):
    """gfm_score_kmers_multi_plan: how score_multi would group these same-width motifs.
    -> (group sizes int32[M], waves per workgroup int32[M])."""
    M = len(motifs)
    handles = (ctypes.c_void_p * M)(*[m.handle for m in motifs])
    wh = None
    if with_hist is not None:
        wh = (ctypes.c_int32 * M)(*[int(bool(x)) for x in with_hist])
    sizes = np.zeros(M, dtype=np.int32)
    waves = np.zeros(M, dtype=np.int32)
    nv.check(nv.lib().gfm_score_kmers_multi_plan(handles, M, wh, nv.ptr(sizes), nv.ptr(waves)))
    return sizes, waves


def qvalue_table_multi(motifs, hists, threshold, on_qvalue, qtables=None, cutoffs=None, nrows=None, stream=None,
                       clear_hist=False):
    """gfm_qvalue_table_multi: the q-tables of several DeviceMotifs (any widths) in three launches per eight
    motifs.  hists / qtables / cutoffs / nrows: lists of torch tensors (the optional ones may be None or hold
    None entries)."""
    M = len(motifs)
    vp = ctypes.c_void_p

    def ptr_array(tensors):
        if tensors is None:
            return None
        return (vp * M)(*[(t.data_ptr() if t is not None else None) for t in tensors])

    handles = (vp * M)(*[m.handle for m in motifs])
    nv.check(nv.lib().gfm_qvalue_table_multi(
        handles, M, ptr_array(hists), float(threshold), int(bool(on_qvalue)), ptr_array(qtables),
        ptr_array(cutoffs), ptr_array(nrows), nv.GFM_FLAG_CLEAR_HIST if clear_hist else 0, _stream_ptr(stream)))


def calibrate_stream(loads_per_store: int, in_bytes: int, write_through: bool, launches: int = 20):
    """gfm_calibrate_stream -> (us per launch, bytes read + written per launch): the bare-stream floor of the score
    kernel's byte mix on the current device."""
    us, nbytes = ctypes.c_double(), ctypes.c_double()
    nv.check(nv.lib().gfm_calibrate_stream(int(loads_per_store), int(in_bytes), int(bool(write_through)),
                                           int(launches), ctypes.byref(us), ctypes.byref(nbytes)))
    return us.value, nbytes.value


def _stream_ptr(stream):
    if stream is None:
        torch = _torch()
        return torch.cuda.current_stream().cuda_stream
    if isinstance(stream, int):
        return stream
    return stream.cuda_stream


# ---------------------------------------------------------------------------------------
# free functions that replace the reference's motif_processing numerics (S1 seam)
def compute_log_odds_dense(probs, bg):
    probs = np.ascontiguousarray(probs, dtype=np.float64)
    bg = np.ascontiguousarray(bg, dtype=np.float64)
    out = np.empty_like(probs)
    nv.check(nv.lib().gfm_compute_log_odds(nv.ptr(probs), probs.shape[1], nv.ptr(bg), nv.ptr(out)))
    return out


def scale_pwm_dense(logodds):
    lo = np.ascontiguousarray(logodds, dtype=np.float64)
    W = lo.shape[1]
    sm = np.empty((4, W), dtype=np.int64)
    mn, mx, sc = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    off = ctypes.c_double()
    nv.check(nv.lib().gfm_scale_pwm(nv.ptr(lo), W, nv.ptr(sm), ctypes.byref(mn), ctypes.byref(mx),
                                    ctypes.byref(sc), ctypes.byref(off)))
    return sm, mn.value, mx.value, sc.value, np.double(off.value)


def comp_pval_mat_dense(score_matrix, bg):
    """Device DP -> host pmf f64[1000*W+1]."""
    sm = np.ascontiguousarray(score_matrix, dtype=np.int64)
    bg = np.ascontiguousarray(bg, dtype=np.float64)
    W = sm.shape[1]
    out = np.empty(RANGE * W + 1, dtype=np.float64)
    nv.check(nv.lib().gfm_comp_pval_mat(nv.ptr(sm), W, nv.ptr(bg), nv.ptr(out)))
    return out
