"""Two processes on ONE GPU with the gloo backend (RCCL refuses a duplicate GPU): exercises the N = 2
code path of KmerScanner -- histogram all-reduce, hit gather (or its all_gather fallback), global row ids,
merge on rank 0 -- on the real kernels.  Not a performance number."""
import os, sys, socket
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch, torch.distributed as dist
    from grafimo_amd import synth
    from grafimo_amd.device import DeviceMotif
    from grafimo_amd.motif_ops import build_motif_meme_host
    from grafimo_amd.scan import KmerScanner
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m = build_motif_meme_host(os.path.join(ROOT, "tests/golden/ref_data/MA0139.1.meme"), "unfrm_dst", 0.1, False)[0]
    dm = DeviceMotif(m.dense_score_matrix(), m.dense_bg(), m.min_val, m.scale, m.offset)
    full = synth.make_batch(400, 2000, 19, np.asarray(m.count_matrix), synth.seed_for(2))
    n_all = len(full)
    lo, hi = rank * n_all // world, (rank + 1) * n_all // world
    d = torch.from_numpy(full.kmers[lo:hi]).cuda()
    sc = KmerScanner(dm, hi - lo, hit_capacity=(hi - lo) // 16, device=d.device, side_stream=True)
    for on_q, thr in [(False, 1e-4), (True, 0.2)]:
        for _ in range(5):
            slot = sc.enqueue(d, thr, on_qvalue=on_q, row_base=lo, gather_hits=True)
        r = sc.collect(slot)
        if rank == 0:
            # single-process reference over all rows
            d_all = torch.from_numpy(full.kmers).cuda()
            one = KmerScanner(dm, n_all, device=d.device, side_stream=False, group=dist.new_group([0]))
            ref = one.collect(one.enqueue(d_all, thr, on_qvalue=on_q))
            ok = (np.array_equal(ref["rows"], r["rows"]) and np.array_equal(ref["scaled"], r["scaled"])
                  and np.array_equal(ref["qtable"], r["qtable"]) and ref["n_scored"] == r["n_scored"] == n_all)
            print(f"on_q={on_q}: {len(r['rows'])} hits of {r['n_scored']} rows, two ranks == one process: {ok}", flush=True)
            if not ok:
                open(out, "w").write("mismatch")
        else:
            dist.new_group([0])
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    out = "/tmp/two_rank_probe.flag"
    if os.path.exists(out): os.remove(out)
    mp.spawn(worker, args=(2, port, out), nprocs=2, join=True)
    sys.exit(1 if os.path.exists(out) else 0)
