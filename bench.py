#!/usr/bin/env python3
"""bench.py -- scored haplotype k-mers/s of the MI355X scoring path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One "step" = one pass of the hot path over one batch of synthetic k-mers resident in HBM:
score kernel (+ fused p-value selection) -> histogram reduction -> [all-reduce of the
score histogram across ranks] -> BH q-value table -> [gather of hit rows to rank 0].
Workload at every N: BASELINE.json configs[1] per GPU -- CTCF MA0139.1 (W=19), 10 000
synthetic 200-bp regions x 2 000 haplotype k-mers = 2.0e7 windows (weak scaling: each rank
scores its own 2.0e7-row shard; value = all ranks' k-mers / max-over-ranks time).

Prints ONE JSON line on rank 0 (contract in the task statement) carrying `roofline`
(score kernel alone, HIP events on its launch stream) and `cpu_baseline` (the CPU oracle's
reference-faithful loop on the host cores, bounded sample; N=1 only).
"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E vendor peak (MI355X_MICROARCH.md), GB/s


def load_ctcf():
    """CTCF MA0139.1 through the package's own motif pipeline (MEME parser -> log-odds ->
    scaling); the p-value DP runs on the device when the DeviceMotif is created."""
    from grafimo_amd.motif_ops import build_motif_meme_host
    path = os.path.join(ROOT, "tests", "golden", "ref_data", "MA0139.1.meme")
    return build_motif_meme_host(path, "unfrm_dst", 0.1, False)[0]


# ---------------------------------------------------------------------------- CPU baseline
def _cpu_worker(args):
    kmers, sm, pmf, min_val, scale, offset = args
    from oracle import oracle as orc
    t = time.perf_counter()
    orc.score_kmers(kmers, sm, pmf, min_val, scale, offset, sum_mode=0)
    return time.perf_counter() - t


def cpu_baseline(kmers, sm, pmf, min_val, scale, offset, target_s=12.0):
    """Reference-faithful CPU loop (per k-mer: W-term integer sum + the two O(1000*W) f64 sums
    of score_sequences.py:390-391), one worker process per host core like the reference's
    mp.Process fan-out (score_sequences.py:133-147); oracle = the checker, kind 'port'."""
    from oracle import oracle as orc
    orc.build()
    cores = os.cpu_count() or 1
    probe = kmers[:2000]
    t = time.perf_counter()
    orc.score_kmers(probe, sm, pmf, min_val, scale, offset, sum_mode=0)
    per_row = (time.perf_counter() - t) / len(probe)
    per_core = max(2000, int(target_s / per_row))
    total = min(len(kmers), per_core * cores)
    per_core = total // cores
    total = per_core * cores
    parts = [(kmers[i * per_core:(i + 1) * per_core], sm, pmf, min_val, scale, offset)
             for i in range(cores)]
    t = time.perf_counter()
    with mp.get_context("fork").Pool(cores) as pool:
        pool.map(_cpu_worker, parts)
    wall = time.perf_counter() - t
    return {
        "value": total / wall, "unit": "k-mers/s", "cores": cores, "kind": "port",
        "sample": f"first {total} k-mers of the same batch, reference-faithful per-row tail sums "
                  f"(oracle/grafimo_oracle.c orc_score_kmers), {cores} worker processes, {wall:.1f} s",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--regions", type=int, default=10_000)
    ap.add_argument("--rows-per-region", type=int, default=2_000)
    ap.add_argument("--threshold", type=float, default=1e-4)
    ap.add_argument("--qvalue-threshold", action="store_true", help="--qvalueT: threshold on q")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--slots", type=int, default=2, help="buffer slots of the scan pipeline")
    ap.add_argument("--force-dist", action="store_true",
                    help="diagnostic: initialise torch.distributed and issue the collectives even with "
                         "one rank (exercises the N > 1 code path on a 1-GPU box)")
    ap.add_argument("--gather-group", action="store_true",
                    help="N > 1: run the hit gather on a second RCCL communicator and stream")
    ap.add_argument("--event-every", type=int, default=8,
                    help="bracket the score kernel of every n-th step with a hipEvent pair")
    ap.add_argument("--no-kernel-events", action="store_true",
                    help="diagnostic: do not bracket the score kernel with events in the timed region")
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

    from grafimo_amd import synth
    motif = load_ctcf()
    W = motif.width
    probs = np.asarray(motif.count_matrix, dtype=np.float64)
    sm = motif.dense_score_matrix()
    batch = synth.make_batch(args.regions, args.rows_per_region, W, probs,
                             synth.seed_for(2, rank), region_base=rank * args.regions)
    n = len(batch)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # before any HIP initialisation: the workers are forked
        from oracle import oracle as orc
        pmf_cpu = orc.comp_pval_mat(sm, motif.dense_bg())
        cpu = cpu_baseline(batch.kmers, sm, pmf_cpu, motif.min_val, motif.scale, float(motif.offset))

    import torch
    import torch.distributed as dist
    from grafimo_amd.device import DeviceMotif
    from grafimo_amd.scan import KmerScanner

    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    if use_dist:
        # RCCL's kernels need CUs of their own next to the persistent score grid: leave 16 free
        # (the score kernel is HBM-bound; 240 CUs move the same bytes, measured -1 %)
        os.environ.setdefault("GRAFIMO_RESERVE_CUS", "16")
    dm = DeviceMotif(sm, motif.dense_bg(), motif.min_val, motif.scale, motif.offset)  # device DP
    d_kmers = torch.from_numpy(batch.kmers).to(dev)
    hit_cap = max(4096, n // 64)   # fixed-size hit buffer (what the gather to rank 0 moves): 1.6 % of the rows
    # optional: hits on their own communicator so that gather(k) overlaps all-reduce(k+1).  Off by
    # default: two communicators whose kernels become ready in different orders on different ranks
    # are a classic deadlock hazard, and this path cannot be exercised on the 1-GPU test boxes.
    gather_group = dist.new_group(backend="nccl") if (use_dist and args.gather_group) else None
    scanner = KmerScanner(dm, n, hit_capacity=hit_cap, device=dev, gather_group=gather_group,
                          group=None, side_stream=args.overlap != "off", n_slots=args.slots,
                          always_collective=args.force_dist)

    def step():
        return scanner.enqueue(d_kmers, args.threshold, on_qvalue=args.qvalue_threshold,
                               want_qvalues=True, row_base=rank * n, gather_hits=use_dist)

    def fence():
        scanner.finish()
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    fence()
    if not args.no_kernel_events:
        dm.profile_enable(min(args.steps, 1024), every=args.event_every)
    fence()
    t0 = time.perf_counter()
    slot = None
    for _ in range(args.steps):
        slot = step()
    fence()
    elapsed = time.perf_counter() - t0
    kernel_ms = dm.profile_read()
    dm.profile_enable(0)

    # outputs of the last step stay valid: sanity-check them (outside the timed region)
    res = scanner.collect(slot)
    n_hits = int(len(res["rows"]))
    assert res["n_scored"] == n * world, (res["n_scored"], n, world)

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = n * world * args.steps / elapsed
        k_ms = float(np.mean(kernel_ms)) if len(kernel_ms) else float("nan")
        alg_bytes = n * (W + 4)
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            with open(tpath) as fh:
                tj = json.load(fh)
            if tj.get("rows_per_launch") == n and tj.get("width") == W:
                traffic = tj.get("hbm_bytes_per_launch")
        out = {
            "metric": "scored haplotype k-mers/sec",
            "value": value,
            "unit": "k-mers/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int32",
            "data": "synthetic",
            "config": {
                "workload": "BASELINE configs[1] per GPU: CTCF MA0139.1 W=19, "
                            f"{args.regions} synthetic 200bp regions x {args.rows_per_region} "
                            f"haplotype k-mers = {n} windows",
                "threshold": args.threshold,
                "threshold_on": "q-value" if args.qvalue_threshold else "p-value",
                "qvalues": True,
                "hits_last_step": n_hits,
                "sharding": f"regions split over {world} rank(s); all-reduce(score histogram) + "
                            "gather(hits) per step" if world > 1 else "single GPU",
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "score_hist_kernel",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "algorithmic_bytes_per_launch": alg_bytes,
                "kernel_ms_avg": k_ms,
                "kernel_launches_timed": int(len(kernel_ms)),
            },
            "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
