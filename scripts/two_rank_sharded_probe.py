"""Two processes on ONE GPU over gloo through the PRODUCT's sharded entry points (compute_results_sharded,
compute_results_many_sharded: every rank runs the streamed scan over its shard of the TSV files, the histograms are
all-reduced between gfm_scan_tsv_begin and _finish, the hit rows are gathered): rank 0's tables must equal what one
process gives over all files.  Exits 1 otherwise.  Not a performance number."""
import contextlib, io, os, socket, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def table_key(df):
    key = ["p-value", "start", "stop", "strand", "matched_sequence"]
    return df.sort_values(key).reset_index(drop=True)


def worker(rank, world, port, seqdir, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch, torch.distributed as dist
    from grafimo_amd.distributed import compute_results_many_sharded, compute_results_sharded
    from grafimo_amd.motif_ops import build_motif_meme_host
    from grafimo_amd.score_sequences import compute_results
    from grafimo_amd.workflow import Findmotif
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    golden = os.path.join(ROOT, "tests", "golden")
    ctcf = build_motif_meme_host(os.path.join(golden, "ref_data", "MA0139.1.meme"), "unfrm_dst", 0.1, False)[0]
    ctcf2 = build_motif_meme_host(os.path.join(golden, "ref_data", "MA0139.1.meme"), os.path.join(golden, "synth", "bg_1.txt"), 0.1, False)[0]
    bad = []
    for kw in (dict(threshold=1e-3), dict(threshold=0.3, qval_t=True, recomb=True), dict(threshold=1e-2, no_qvalue=True)):
        wf = Findmotif(cores=4, **kw)
        with contextlib.redirect_stdout(io.StringIO()):
            one = compute_results_sharded(ctcf, seqdir, True, wf)
            many = compute_results_many_sharded([ctcf, ctcf2], seqdir, True, wf)
        if rank == 0:
            with contextlib.redirect_stdout(io.StringIO()):
                ref = [compute_results(m, seqdir, True, wf) for m in (ctcf, ctcf2)]
            for got, exp, name in ((one, ref[0], "sharded"), (many[0], ref[0], "many[0]"), (many[1], ref[1], "many[1]")):
                a, b = table_key(got), table_key(exp)
                same = len(a) == len(b) and all((a[c].astype(str) == b[c].astype(str)).all() for c in b.columns
                                                 if b[c].dtype.kind != "f") and \
                    all(np.allclose(a[c].to_numpy(float), b[c].to_numpy(float), rtol=1e-12, atol=0) for c in b.columns
                        if b[c].dtype.kind == "f")
                print(f"{kw} {name}: {len(a)} rows, two ranks == one process: {same}", flush=True)
                if not same or len(a) == 0:
                    bad.append((kw, name))
        else:
            assert one is None and all(x is None for x in many)
    # ---- the graph path: every rank is given the HOST index and uploads only the shard of the graph its regions touch
    from grafimo_amd import synth
    from grafimo_amd.extract_regions import DeviceGraph, compute_results_from_graph, compute_results_from_graph_many
    idx, regions = synth.make_graph_index(600, 19)
    reg = np.asarray(regions, dtype=np.int64)
    syn = synth.motif_object(synth.synthetic_motif(19, np.random.default_rng(7), np.full(4, 0.25)), "SYN19")
    solo = dist.new_group([0])                      # (a group of rank 0 alone: the one-process reference inside this job)
    for kw in (dict(threshold=1e-3), dict(threshold=0.5, qval_t=True, recomb=True)):
        wf = Findmotif(cores=2, **kw)
        with contextlib.redirect_stdout(io.StringIO()):
            one = compute_results_from_graph(ctcf, idx, reg, True, wf)
            many = compute_results_from_graph_many([ctcf, ctcf2, syn], idx, reg, True, wf)
        from grafimo_amd.extract_regions import _SHARD_GRAPHS
        mine = [g for g in _SHARD_GRAPHS.values() if g._source is idx]
        assert len(mine) == 1 and 0 < len(mine[0].index.pos) < len(idx.pos), "this rank holds a shard of the graph, not a replica"
        if rank == 0:
            full = DeviceGraph(idx)
            with contextlib.redirect_stdout(io.StringIO()):
                ref = [compute_results_from_graph(m, full, reg, True, wf, group=solo) for m in (ctcf, ctcf2, syn)]
            full.close()
            for got, exp, name in ((one, ref[0], "graph"), (many[0], ref[0], "graph many[0]"), (many[1], ref[1], "graph many[1]"),
                                   (many[2], ref[2], "graph many[2]")):
                a, b = table_key(got), table_key(exp)
                same = len(a) == len(b) and all((a[c].astype(str) == b[c].astype(str)).all() for c in b.columns
                                                 if b[c].dtype.kind != "f") and \
                    all(np.allclose(a[c].to_numpy(float), b[c].to_numpy(float), rtol=1e-12, atol=0) for c in b.columns
                        if b[c].dtype.kind == "f")
                print(f"{kw} {name}: {len(a)} rows, ranks holding shards of the graph == one process with all of it: {same}", flush=True)
                if not same or (len(a) == 0 and not kw.get("qval_t")):
                    bad.append((kw, name))
        else:
            assert one is None and all(x is None for x in many)
    # ---- the same through scan_graph's MANIFEST and the sharded entry points (the unchanged call sequence of grafimo.findmotif
    # under a process group): rank 0 leaves the manifest, every rank reads it, cuts its shard of the graph and of the regions,
    # the motif set goes through compute_results_many_sharded -> compute_results_from_graph_many
    from grafimo_amd import extract_regions as xr
    from grafimo_amd.distributed import compute_results_many_sharded as many_sharded, compute_results_sharded as one_sharded
    xr.drop_graph_cache()                           # (the shards of the sections above)
    box = [None]
    if rank == 0:
        gdir = os.path.join(seqdir, "graphs")
        os.makedirs(gdir, exist_ok=True)
        idx.save(os.path.join(gdir, "chr22"))
        bed = os.path.join(seqdir, "regions.bed")
        with open(bed, "w") as fh:
            fh.write("".join(f"chr22\t{s_}\t{e_}\n" for s_, e_ in regions))
        os.environ["GRAFIMO_SCAN_OUTPUT"] = "manifest"
        with contextlib.redirect_stdout(io.StringIO()):
            box[0] = xr.scan_graph({19}, Findmotif(graph_genome_dir=gdir, bedfile=bed, chroms_prefix="chr"), True)
    dist.broadcast_object_list(box, src=0)
    loc = box[0]
    wf = Findmotif(cores=2, threshold=1e-3)
    with contextlib.redirect_stdout(io.StringIO()):
        many = many_sharded([ctcf, ctcf2, syn], loc, True, wf)
        one = one_sharded(syn, loc, True, wf)
    mine = [g for g in xr._SHARD_GRAPHS.values() if g._h is not None]
    assert len(mine) == 1 and 0 < len(mine[0].index.pos) < len(idx.pos), "this rank holds a shard of the manifest's graph"
    if rank == 0:
        full = DeviceGraph(idx)
        with contextlib.redirect_stdout(io.StringIO()):
            ref = [compute_results_from_graph(m, full, reg, True, wf, group=solo) for m in (ctcf, ctcf2, syn)]
        full.close()
        for got, exp, name in ((many[0], ref[0], "manifest many[0]"), (many[1], ref[1], "manifest many[1]"), (many[2], ref[2], "manifest many[2]"),
                               (one, ref[2], "manifest one")):
            a, b = table_key(got), table_key(exp)
            same = len(a) == len(b) and len(a) > 0 and all((a[c].astype(str) == b[c].astype(str)).all() for c in b.columns if b[c].dtype.kind != "f") and \
                all(np.allclose(a[c].to_numpy(float), b[c].to_numpy(float), rtol=1e-12, atol=0) for c in b.columns if b[c].dtype.kind == "f")
            print(f"{name}: {len(a)} rows, the manifest under two ranks == one process with the whole graph: {same}", flush=True)
            if not same:
                bad.append(("manifest", name))
    else:
        assert one is None and all(x is None for x in many)
    dist.barrier()
    if rank == 0:
        import shutil
        shutil.rmtree(loc, ignore_errors=True)
    if rank == 0 and bad:
        open(out, "w").write(str(bad))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    import torch.multiprocessing as mp
    from grafimo_amd import synth
    from grafimo_amd.motif_ops import build_motif_meme_host
    m = build_motif_meme_host(os.path.join(ROOT, "tests/golden/ref_data/MA0139.1.meme"), "unfrm_dst", 0.1, False)[0]
    seqdir = tempfile.mkdtemp(prefix="gfm_two_rank_")
    # `--files K`: K regions = K files (default 9); K = 1 leaves rank 1 with an EMPTY shard: its scan has no file at all
    # (gfm_scan_tsv_begin with n_paths == 0 once polled for a chunk that never came) and its histogram still joins the
    # all-reduce
    n_files = int(sys.argv[sys.argv.index("--files") + 1]) if "--files" in sys.argv else 9
    synth.write_tsv_dir(synth.make_batch(n_files, 1500, 19, np.asarray(m.count_matrix), synth.seed_for(3)), seqdir)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    out = os.path.join(seqdir, "mismatch.flag")
    try:
        n_ranks = int(sys.argv[sys.argv.index("--ranks") + 1]) if "--ranks" in sys.argv else 2
        mp.spawn(worker, args=(n_ranks, port, seqdir, out), nprocs=n_ranks, join=True)
        sys.exit(1 if os.path.exists(out) else 0)
    finally:
        import shutil
        shutil.rmtree(seqdir, ignore_errors=True)
