#!/usr/bin/env python3
"""bench.py -- scored haplotype k-mers/s of the MI355X scoring path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--config 2|3|4|5]
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One "step" = one pass of the hot path over one batch of synthetic k-mers resident in HBM:
score kernel (+ fused p-value selection) -> histogram reduction -> [all-reduce of the score histogram
across ranks] -> BH q-value table -> [selection on q for --qvalueT] -> [gather of the hit rows to rank 0].

Workloads (BASELINE.json configs; weak scaling: every rank scores its own shard, value = all ranks'
units / max-over-ranks time):
  --config 2  (default at N=1) CTCF MA0139.1 W=19, 10 000 synthetic 200-bp regions x 2 000 haplotype
              k-mers = 2.0e7 windows per GPU, p < 1e-4, q-values on.  Two input buffers are used in
              turn (380 MB each) so that no step re-reads what the 256 MiB Infinity Cache may still hold.
  --config 3  (default at N>1) the per-GPU shard of the 1e9-window scan: 1.25e8 windows, generated on
              the device.
  --config 4  W=30 JASPAR-style motif, 50 000 regions x 2 000 = 1.0e8 windows, both strands, --qvalueT 1e-4.
  --config 5  50 JASPAR-style PWMs, widths cycling 8..25, per-motif background, 1.0e8 windows per width
              (split over the ranks); motifs of one width share each read of the k-mers
              (gfm_score_kmers_multi); unit = one (k-mer, motif) pair.

Timing: W warm-up steps, then `--bursts` (5) bursts of EXACTLY K steps, each burst bracketed by a barrier +
torch.cuda.synchronize() on both sides and reduced with MAX over the ranks; the line reports the MEDIAN
burst.  Rank 0 prints ONE JSON line carrying `roofline` (the score kernel alone: HIP events on its launch
stream, sampled inside the timed bursts), `cpu_baseline` / `cpu_baseline_table` (the CPU oracle's loop over
TSV text -- parse + score -- with the reference's two O(1000 W) sums per row, resp. one table lookup; forked
workers on every host core, bounded sample; N=1 only) and `e2e` (config 2, N=1: a TSV directory through
compute_results' streamed scan: parse threads -> pinned chunks -> H2D -> kernels -> hits).
"""
import argparse
import json
import multiprocessing as mp
import os
import shutil
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E vendor peak (MI355X_MICROARCH.md), GB/s


def load_ctcf():
    """CTCF MA0139.1 through the package's own motif pipeline (MEME parser -> log-odds -> scaling)."""
    from grafimo_amd.motif_ops import build_motif_meme_host
    path = os.path.join(ROOT, "tests", "golden", "ref_data", "MA0139.1.meme")
    return build_motif_meme_host(path, "unfrm_dst", 0.1, False)[0]


# ---------------------------------------------------------------------------- CPU baselines
def _cpu_worker(args):
    text, W, sm, tab, min_val, scale, offset, table, budget_s = args
    from oracle import oracle as orc
    t = time.perf_counter()
    n = passes = 0
    while time.perf_counter() - t < budget_s:          # a bounded amount of CPU work per worker
        n += orc.score_tsv_text(text, W, sm, tab, min_val, scale, offset, table=table)[0]
        passes += 1
    return n, passes


def cpu_baselines(mot, budget_s=8.0):
    """The reference's scoring loop restated (oracle/, the checker -- kind 'port'), timed on the host
    cores of this box: per TSV line the text handling of score_seqs (score_sequences.py:273-321) and
    compute_score_seq (:331-396).  'faithful' = the reference's two O(1000 W) f64 sums per k-mer,
    'table' = one p_table lookup instead (shows how much of the GPU/CPU ratio is algorithmic).  One
    forked worker per core like the reference's mp.Process fan-out (:133-147); every worker goes over
    its sample until `budget_s` seconds have passed."""
    from grafimo_amd import synth
    from oracle import oracle as orc
    orc.build()
    W = mot["width"]
    pmf = orc.comp_pval_mat(mot["sm"], mot["bg"])
    ptab = orc.p_table(pmf)
    sample = synth.make_batch(10, 2000, W, mot["probs"], synth.seed_for(0))
    cores = os.cpu_count() or 1
    out = {}
    for name, tab, table, rows in (("cpu_baseline", pmf, False, 1000), ("cpu_baseline_table", ptab, True, 20000)):
        text = synth.tsv_text(sample, np.arange(rows))
        job = (text, W, mot["sm"], tab, mot["min_val"], mot["scale"], mot["offset"], table, budget_s)
        t = time.perf_counter()
        with mp.get_context("fork").Pool(cores) as pool:
            res = pool.map(_cpu_worker, [job] * cores)
        wall = time.perf_counter() - t
        total = sum(r[0] for r in res)
        out[name] = {
            "value": total / wall, "unit": "k-mers/s", "cores": cores, "kind": "port",
            "sample": f"the first {rows} TSV rows of a batch of the same synthetic recipe, passed over again and again "
                      f"for {budget_s:.0f} s per worker ({min(r[1] for r in res)}..{max(r[1] for r in res)} passes), "
                      f"{cores} forked workers, text parse + "
                      + ("p_table lookup" if table else "per-row O(1000 W) tail sums")
                      + f" (oracle/grafimo_oracle.c orc_score_tsv_text), {wall:.1f} s wall",
        }
    return out


# ---------------------------------------------------------------------------- end to end (TSV -> hits)
def e2e_block(motif, dm, rows_regions=1000, rows_per_region=2000, regions_per_file=1):
    """A TSV directory (synth.write_tsv_dir, 2.0e6 rows) through the pipeline behind compute_results
    (gfm_scan_tsv): parse threads -> pinned chunks -> hipMemcpyAsync -> score kernel per chunk -> q-table ->
    hits back; next to the bare ingest (gfm_tsv_open) on the same files and thread count."""
    import ctypes
    import glob
    from grafimo_amd import _native as nv
    from grafimo_amd import synth
    from grafimo_amd.score_sequences import StreamScan
    W = motif.width
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    tmp = tempfile.mkdtemp(prefix="grafimo_e2e_", dir=base)
    try:
        batch = synth.make_batch(rows_regions, rows_per_region, W, np.asarray(motif.count_matrix), synth.seed_for(7))
        synth.write_tsv_dir(batch, tmp, regions_per_file=regions_per_file)
        files = sorted(glob.glob(os.path.join(tmp, f"width_{W}", "*.tsv")))
        nbytes = sum(os.path.getsize(f) for f in files)
        threads = os.cpu_count() or 1
        arr = (ctypes.c_char_p * len(files))(*[f.encode() for f in files])
        ingest = []
        for _ in range(3):
            h, n = ctypes.c_void_p(), ctypes.c_int64()
            t = time.perf_counter()
            nv.check(nv.lib().gfm_tsv_open(arr, len(files), W, 0, threads, ctypes.byref(h), ctypes.byref(n)))
            ingest.append(time.perf_counter() - t)
            nv.lib().gfm_tsv_close(h)
        ingest_s = float(np.median(ingest))
        runs = []
        for _ in range(4):       # the first call sizes the buffer pool
            t = time.perf_counter()
            sc = StreamScan(dm, files, False, threads, 1e-4, False, True)
            runs.append((time.perf_counter() - t, sc.stats.total_s, sc.stats.parse_s, sc.stats.h2d_s,
                         sc.stats.h2d_bytes, sc.stats.tail_s, sc.n, sc.n_hits, sc.stats.n_chunks))
        runs = sorted(runs[1:])
        wall, total_s, parse_s, h2d_s, h2d_bytes, tail_s, n, n_hits, n_chunks = runs[len(runs) // 2]
        assert n == len(batch)
        return {
            "rows": int(n), "tsv_bytes": int(nbytes), "files": len(files), "host_threads": threads, "hits": int(n_hits),
            "kmers_per_s": n / total_s, "ingest_rows_per_s": n / ingest_s, "h2d_GBps": h2d_bytes / h2d_s / 1e9,
            "total_ms": total_s * 1e3, "ingest_alone_ms": ingest_s * 1e3, "parse_ms_inside": parse_s * 1e3,
            "h2d_ms": h2d_s * 1e3, "after_parse_ms": tail_s * 1e3, "chunks": int(n_chunks),
            "total_over_max_ingest_h2d": total_s / max(ingest_s, h2d_s),
            "path": "gfm_scan_tsv (grafimo_amd.score_sequences.StreamScan, what compute_results calls)",
        }
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", type=int, choices=[2, 3, 4, 5], default=None,
                    help="BASELINE.json config (default: 2 on one GPU, 3 = its per-GPU shard on several)")
    ap.add_argument("--bursts", type=int, default=5, help="timed bursts of --steps steps; the median is reported")
    ap.add_argument("--rows", type=int, default=None, help="override the rows per GPU (diagnostics)")
    ap.add_argument("--threshold", type=float, default=1e-4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true")
    ap.add_argument("--slots", type=int, default=3,
                    help="buffer slots of the scan pipeline (2: the device waits for slot reuse; >= 3: the host does)")
    ap.add_argument("--force-dist", action="store_true",
                    help="diagnostic: initialise torch.distributed and issue the collectives even with "
                         "one rank (exercises the N > 1 code path on a 1-GPU box)")
    ap.add_argument("--event-every", type=int, default=None,
                    help="time the score kernel of every n-th step with a hipEvent pair on its dispatch "
                         "(default: every 8th, more often when the run is short, so that at least ~24 launches are timed)")
    ap.add_argument("--no-candidates", action="store_true",
                    help="config 4: select on the q-value threshold by a pass over every score instead of "
                         "filtering the p < t candidates the score kernel collects (measurement aid)")
    ap.add_argument("--overlap", choices=["auto", "on", "off"], default="auto",
                    help="run the per-step tail (post kernel, collective, q-table, gather) on a side "
                         "stream so that it overlaps the next step's score kernel; auto = on")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N with N > 1 must be launched through torch.distributed.run")
        args.gpus = world
    cfg = args.config if args.config is not None else (2 if world == 1 else 3)

    from grafimo_amd import synth
    ctcf = load_ctcf() if cfg in (2, 3) else None
    if cfg in (2, 3):
        mots = [dict(sm=ctcf.dense_score_matrix(), bg=ctcf.dense_bg(), min_val=ctcf.min_val, scale=ctcf.scale,
                     offset=float(ctcf.offset), probs=np.asarray(ctcf.count_matrix, dtype=np.float64), width=ctcf.width)]
    else:
        mots = synth.config_motifs(cfg)
    on_q = cfg == 4
    rows_default = {2: 20_000_000, 3: 125_000_000, 4: 100_000_000, 5: 100_000_000 // world}[cfg]
    n = int(args.rows) if args.rows else rows_default

    host_batch = None
    if cfg == 2 and not args.rows:
        host_batch = synth.make_batch(10_000, 2_000, mots[0]["width"], mots[0]["probs"], synth.seed_for(2, rank),
                                      region_base=rank * 10_000)
        n = len(host_batch)

    cpu = {}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baselines(mots[0])      # before any HIP initialisation: the workers are forked

    import torch
    import torch.distributed as dist
    from grafimo_amd.device import DeviceMotif
    from grafimo_amd.scan import KmerScanner, SameWidthScanner

    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)
        # RCCL's kernels need CUs of their own next to the persistent score grid: leave 16 free
        # (the score kernel is HBM-bound; 240 CUs move the same bytes, measured -1 %)
        os.environ.setdefault("GRAFIMO_RESERVE_CUS", "16")
    side = args.overlap != "off"

    dms = [DeviceMotif(m["sm"], m["bg"], m["min_val"], m["scale"], m["offset"]) for m in mots]   # DP on device
    # ---- inputs resident in HBM
    if cfg == 5:
        widths = sorted({m["width"] for m in mots})
        groups = {w: [j for j, m in enumerate(mots) if m["width"] == w] for w in widths}
        bufs = {w: [synth.make_device_kmers(n, w, mots[groups[w][0]]["probs"], synth.seed_for(50 + w, rank), dev)]
                for w in widths}
        scanners = {w: SameWidthScanner([dms[j] for j in groups[w]], n, max(4096, n // 64), dev,
                                        side_stream=args.overlap == "on")
                    for w in widths}
        units_per_step = n * len(mots)
        alg_bytes = sum(n * (w + 4 * len(groups[w])) for w in widths)

        def step(i):
            for w in widths:
                scanners[w].enqueue(bufs[w][0], args.threshold, on_qvalue=False, want_qvalues=True, row_base=rank * n)

        rotate = 1
    else:
        m0 = mots[0]
        W = m0["width"]
        if host_batch is not None:
            first = torch.from_numpy(host_batch.kmers).to(dev)
        else:
            first = synth.make_device_kmers(n, W, m0["probs"], synth.seed_for(cfg, rank), dev)
        bufs = [first]
        if first.numel() < (512 << 20):   # small enough to sit in the 256 MiB Infinity Cache in part: alternate
            bufs.append(torch.roll(first, shifts=n // 3, dims=0).contiguous())
        rotate = len(bufs)
        hit_cap = max(4096, n // 64)
        scanner = KmerScanner(dms[0], n, hit_capacity=hit_cap, device=dev, group=None, side_stream=side,
                              n_slots=args.slots, always_collective=args.force_dist,
                              candidates=not args.no_candidates)
        units_per_step = n
        alg_bytes = n * (W + 4)

        def step(i):
            return scanner.enqueue(bufs[i % rotate], args.threshold, on_qvalue=on_q, want_qvalues=True,
                                   row_base=rank * n, gather_hits=use_dist)

    def fence():
        # the device-wide synchronize covers the side streams too: no stream-to-stream waits (finish()) in front of it
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize(dev)

    it = 0
    for _ in range(max(args.warmup, rotate)):
        step(it)
        it += 1
    fence()
    if cfg != 5 and use_dist:
        scanner.size_gather()          # the per-step gather moves what is hit, not the whole hit buffer
    every = max(1, args.event_every) if args.event_every else max(1, min(8, args.bursts * args.steps // 24))
    for d in dms:
        d.profile_enable(min(1024, max(16, args.bursts * args.steps // every + 1)), every=every)
    burst_s = []
    for _ in range(max(1, args.bursts)):
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            last = step(it)
            it += 1
        fence()
        el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        if use_dist:
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
        burst_s.append(float(el.item()))
    kernel_ms = [d.profile_read() for d in dms]
    for d in dms:
        d.profile_enable(0)

    # outputs of the last step stay valid: self-checks outside the timed region
    if cfg == 5:
        torch.cuda.synchronize(dev)
        for w in widths:
            nr = scanners[w].nrows.cpu().numpy()
            assert (nr == n * world).all(), (w, nr, n, world)
        n_hits = int(sum(int(scanners[w].hits[:, 0].sum().item()) for w in widths))
    else:
        res = scanner.collect(last)
        n_hits = int(len(res["rows"]))
        assert res["n_scored"] == n * world, (res["n_scored"], n, world)
    assert n_hits > 0

    e2e = None
    if rank == 0 and world == 1 and cfg == 2 and not args.no_e2e and not args.rows:
        e2e = e2e_block(ctcf, dms[0])

    if rank == 0:
        elapsed = float(np.median(burst_s))
        ms_per_step = 1e3 * elapsed / args.steps
        value = units_per_step * world * args.steps / elapsed
        if cfg == 5:      # a step launches one batched kernel per group of <= 3 same-width motifs: sum of their means
            k_ms = float(sum(float(np.mean(k)) for k in kernel_ms if len(k)))
            timed = int(sum(len(k) for k in kernel_ms))
            kname = "score_quad_kernel<W, MM> (batched, one launch per group of <= 3 same-width motifs; summed per step)"
        else:
            k_ms = float(np.mean(kernel_ms[0])) if len(kernel_ms[0]) else float("nan")
            timed = int(len(kernel_ms[0]))
            kname = f"score_quad_kernel<{mots[0]['width']}, 1>"
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath) and cfg != 5:
            with open(tpath) as fh:
                tj = json.load(fh)
            if tj.get("rows_per_launch") == n and tj.get("width") == mots[0]["width"] and tj.get("kernel") == kname:
                traffic = tj.get("hbm_bytes_per_launch")
        workload = {
            2: f"BASELINE configs[1] per GPU: CTCF MA0139.1 W=19, 10000 synthetic 200bp regions x 2000 haplotype "
               f"k-mers = {n} windows",
            3: f"BASELINE configs[2], one GPU's shard of the 1e9-window scan: CTCF MA0139.1 W=19, {n} windows "
               f"generated on the device",
            4: f"BASELINE configs[3]: synthetic JASPAR-style PWM W=30, 50000 regions x 2000 = {n} windows, both "
               f"strands, --qvalueT (threshold on q)",
            5: f"BASELINE configs[4]: 50 synthetic JASPAR-style PWMs W=8..25 (per-motif background), {n} windows per "
               f"width per GPU, same-width motifs share each k-mer read; unit = (k-mer, motif) pair",
        }[cfg]
        out = {
            "metric": "scored haplotype k-mers/sec",
            "value": value,
            "unit": "k-mers/s" if cfg != 5 else "(k-mer, motif) pairs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": max(args.warmup, rotate),
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int32",
            "data": "synthetic",
            "config": {
                "workload": workload,
                "baseline_config": cfg,
                "threshold": args.threshold,
                "threshold_on": "q-value" if on_q else "p-value",
                "qvalues": True,
                "hits_last_step": n_hits,
                "input_buffers_rotated": rotate,
                "bursts": len(burst_s),
                "burst_ms": [round(1e3 * b, 4) for b in burst_s],
                "timing": "median burst of `steps` steps, each burst barrier + synchronize bracketed, MAX over ranks",
                "sharding": (f"regions split over {world} rank(s); all-reduce(score histogram) + gather(hit entries, "
                             f"sized from the observed hit counts) per step") if world > 1 else "single GPU",
            },
            "roofline": {
                "bound": "hbm",
                "kernel": kname,
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "algorithmic_bytes_per_launch": alg_bytes,
                "kernel_ms_avg": k_ms,
                "kernel_launches_timed": timed,
            },
            "cpu_baseline": cpu.get("cpu_baseline"),
            "cpu_baseline_table": cpu.get("cpu_baseline_table"),
            "e2e": e2e,
        }
        # RCCL's version banner sits in the C runtime's stdout buffer until the process ends: push it out first,
        # so that the JSON line is the last line of the run
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
