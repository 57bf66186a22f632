"""Strand-max and per-region best hit -- the reductions BASELINE.json's north_star names ("wavefront-level
reductions for forward/reverse-strand max", "RCCL ... only for the final top-hit gather") and the reference
does not have: GRAFIMO keeps every strand as a row of its own (score_sequences.py:279-321) and reports rows.
Its one consumer of "the best hit of a region" is --top-graphs, which walks the p-sorted report and keeps the
first N distinct sequence_names (res_writer.py:153-157): the regions ranked by their best reported hit.

Nothing here changes the reported rows (mismatch (i)).  What is added:

  * region_best / locus_max -- the device reductions (gfm_region_best, gfm_locus_max: csrc/region_reduce.hip) over a
    resident batch: per region its best (score, row); per locus -- (region, {start, stop}), the span a k-mer
    occupies on either strand -- the best score of any of its rows;
  * best_rows_per_region / top_regions_table -- the same on the host for hit tables that were annotated there
    (the TSV scan keeps its rows' columns in host memory);
  * top_regions -- the reference's selection rule on a report table;
  * compute_top_regions_sharded -- under torch.distributed every rank reduces its hits to ONE row per region
    before the gather: rank 0 receives n_regions entries per rank instead of every hit.
"""
import ctypes
from typing import List, Optional

import numpy as np
import pandas as pd

from . import _native as nv
from .device import _stream_ptr, _torch

BEST_ROW_BITS = nv.GFM_BEST_ROW_BITS
_ROW_MASK = (1 << BEST_ROW_BITS) - 1


# ------------------------------------------------------------------------------ device reductions
def region_ids(region_off, n: int, device=None, stream=None):
    """int32 [n] region id of every row of a batch laid out region after region; `region_off` int64
    [n_regions + 1] (numpy or torch): rows [off[r], off[r + 1]) belong to region r."""
    torch = _torch()
    off = region_off if torch.is_tensor(region_off) else torch.from_numpy(np.ascontiguousarray(region_off, dtype=np.int64))
    device = device if device is not None else (off.device if off.is_cuda else torch.device("cuda", torch.cuda.current_device()))
    off = off.to(device=device, dtype=torch.int64).contiguous()
    out = torch.full((int(n),), -1, dtype=torch.int32, device=device)
    nv.check(nv.lib().gfm_region_ids(off.data_ptr(), int(off.numel()) - 1, int(n), out.data_ptr(), _stream_ptr(stream)))
    return out


def region_best(scores, region, n_regions: int, freq=None, min_score: int = 0, cutoff=None, row_base: int = 0,
                out=None, stream=None):
    """gfm_region_best over device tensors: scores int32 [n], region int32 [n], freq int64 [n] or None (rows with
    freq == 0 take no part: the report without --recomb), cutoff int32 [1] on the device or None.
    -> int64 [n_regions] keys (score << 44 | 2^44 - 1 - row; 0 = no row); pass `out` to accumulate batches."""
    torch = _torch()
    n = int(scores.numel())
    if out is None:
        out = torch.zeros(int(n_regions), dtype=torch.int64, device=scores.device)
    assert scores.dtype == torch.int32 and region.dtype == torch.int32 and int(region.numel()) == n
    assert freq is None or (freq.dtype == torch.int64 and int(freq.numel()) == n)
    nv.check(nv.lib().gfm_region_best(scores.data_ptr() if n else None, n, region.data_ptr() if n else None,
                                      int(n_regions), freq.data_ptr() if freq is not None and n else None,
                                      int(min_score), cutoff.data_ptr() if cutoff is not None else None, int(row_base),
                                      out.data_ptr() if int(n_regions) else None, _stream_ptr(stream)))
    return out


def decode_best(best):
    """keys of region_best (numpy / torch int64) -> (score int32, row int64, valid bool) per region."""
    b = best.cpu().numpy() if hasattr(best, "cpu") else np.asarray(best)
    b = b.astype(np.uint64)
    valid = b != 0
    score = (b >> np.uint64(BEST_ROW_BITS)).astype(np.int32)
    row = (np.uint64(_ROW_MASK) - (b & np.uint64(_ROW_MASK))).astype(np.int64)
    row[~valid] = -1
    score[~valid] = -1
    return score, row, valid


def locus_max(scores, region, n_regions: int, start, stop, freq=None, min_score: int = 0, cutoff=None, stream=None):
    """gfm_locus_max over device tensors -> int32 [n]: for every row the best score among the rows of its locus
    (same region, same {start, stop}: both strands, every haplotype k-mer of that span); -1 for rows that take no
    part (score below the cutoff, freq == 0 when freq is given)."""
    torch = _torch()
    n = int(scores.numel())
    out = torch.empty(n, dtype=torch.int32, device=scores.device)
    if n == 0:
        return out
    nbytes = int(nv.lib().gfm_locus_max_workspace(n))
    work = torch.empty((nbytes + 7) // 8, dtype=torch.int64, device=scores.device)
    assert start.dtype == torch.int64 and stop.dtype == torch.int64 and region.dtype == torch.int32
    nv.check(nv.lib().gfm_locus_max(scores.data_ptr(), n, region.data_ptr(), int(n_regions), start.data_ptr(),
                                    stop.data_ptr(), freq.data_ptr() if freq is not None else None, int(min_score),
                                    cutoff.data_ptr() if cutoff is not None else None, work.data_ptr(), nbytes,
                                    out.data_ptr(), _stream_ptr(stream)))
    lost = int(work[0].item())           # (synchronises: the table was sized for every row, so this is a self-check)
    if lost:
        raise OverflowError(f"gfm_locus_max: {lost} rows found no slot in the locus table")
    return out


# ------------------------------------------------------------------------------ the same on host hit tables
def best_rows_per_region(region_ix, scaled, rows, keep=None) -> np.ndarray:
    """Indices (into the given arrays) of the best hit of every region that has one: highest scaled score, lowest
    row among equals -- the first row of that region in the report sorted by p-value (resultsTmp.py:312).  `keep`:
    bool mask of the rows that are reported at all (haplotype_frequency > 0 without --recomb)."""
    region_ix = np.asarray(region_ix, dtype=np.int64)
    scaled = np.asarray(scaled, dtype=np.int64)
    rows = np.asarray(rows, dtype=np.int64)
    idx = np.arange(len(region_ix)) if keep is None else np.nonzero(np.asarray(keep, dtype=bool))[0]
    if not len(idx):
        return idx
    order = idx[np.lexsort((rows[idx], -scaled[idx], region_ix[idx]))]      # region, then best score, then lowest row
    first = np.concatenate(([True], region_ix[order][1:] != region_ix[order][:-1]))
    return order[first]


def locus_max_of_hits(region_ix, starts, stops, scaled) -> np.ndarray:
    """Per hit row the best scaled score at its locus -- (region, {start, stop}) -- among the given rows.  For the rows
    over a score cutoff this equals gfm_locus_max over ALL rows with that cutoff: a row under the cutoff can not be a
    locus' maximum when a row over it exists."""
    lo, hi = np.minimum(starts, stops), np.maximum(starts, stops)
    key = np.stack([np.asarray(region_ix, dtype=np.int64), lo.astype(np.int64), hi.astype(np.int64)], axis=1)
    if not len(key):
        return np.zeros(0, dtype=np.int32)
    _, inv = np.unique(key, axis=0, return_inverse=True)
    inv = inv.reshape(-1)
    best = np.full(int(inv.max()) + 1, -1, dtype=np.int64)
    np.maximum.at(best, inv, np.asarray(scaled, dtype=np.int64))
    return best[inv].astype(np.int32)


def top_regions(results: pd.DataFrame, top_graphs: int) -> List[str]:
    """The reference's --top-graphs selection (res_writer.py:153-157): the first `top_graphs` distinct
    sequence_names of the report (sorted by p-value), in report order."""
    out: List[str] = []
    seen = set()
    for r in results["sequence_name"].tolist():
        if len(out) >= top_graphs:
            break
        if r not in seen:
            seen.add(r)
            out.append(r)
    return out


def top_regions_table(results: pd.DataFrame, top_graphs: Optional[int] = None) -> pd.DataFrame:
    """One row per region -- its best reported hit -- in report order (ascending p-value); the first `top_graphs` of
    them.  `top_regions(results, n)` == list(top_regions_table(results, n)["sequence_name"])."""
    t = results.drop_duplicates("sequence_name", keep="first")
    if top_graphs is not None:
        t = t.head(int(top_graphs))
    return t.reset_index(drop=True)


# ------------------------------------------------------------------------------ sharded: the top-hit-only gather
def compute_top_regions_sharded(motif, sequence_loc: str, debug: bool, args_obj, top_graphs: Optional[int] = None,
                                group=None, backend=None, stats: Optional[dict] = None) -> Optional[pd.DataFrame]:
    """distributed.compute_results_sharded when only the top regions are asked for: every rank scans its shard of the
    TSV files, the score histograms are all-reduced as always (q-values are global), but each rank then keeps ONE hit
    per region -- the best of the rows it would report -- and rank 0 gathers those: n_regions entries per rank instead
    of every hit.  -> on rank 0 top_regions_table(compute_results_sharded(...), top_graphs), elsewhere None.
    `stats` (optional dict) receives the number of rows this rank sent."""
    import glob
    import os
    from . import distributed as D
    from .score_sequences import print_scoring_msg
    dist = D._dist()
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    no_qvalue, qval_t, recomb = bool(args_obj.noqvalue), bool(args_obj.qvalueT), bool(args_obj.recomb)
    if qval_t and no_qvalue:
        raise ValueError("q-value threshold without q-values")
    if rank == 0:
        print_scoring_msg(motif, bool(args_obj.noreverse), debug)
    width = motif.width
    files = sorted(glob.glob(os.path.join(sequence_loc, f"width_{width}", "*.tsv")))

    def keep_best(cols):
        keep = None if recomb else (np.asarray(cols["freq"]) > 0)
        sel = np.sort(best_rows_per_region(cols["name_id"], cols["scaled"], cols["rows"], keep))
        if stats is not None:
            stats["hits"] = int(len(cols["rows"]))
            stats["sent"] = int(len(sel))
        return {k: np.ascontiguousarray(v[sel]) for k, v in cols.items()}

    got, all_names, n_global = D._scan_width_sharded([motif], files, width, args_obj, group, backend, debug, reduce=keep_best)
    if rank != 0:
        return None
    df = D._frame_from_columns(motif, got[0], all_names, no_qvalue, recomb)
    print(f"Scanned sequences:\t{n_global}")
    print(f"Scanned nucleotides:\t{n_global * width}")
    return top_regions_table(df, top_graphs)
