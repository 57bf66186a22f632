"""Streamed scan of 2e7 rows in 10 000 files (bench.py's e2e_config2 input): wall time AND the CPU seconds the process
spent per scan (all threads) -- under a CPU quota the second number is what a caller in a loop pays for.
usage: ingest_cpu_probe.py [threads]     (GRAFIMO_SCAN_NO_AVX512=1: the AVX2 scanner on an AVX-512 host)"""
import glob, os, resource, shutil, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

threads = int(sys.argv[1]) if len(sys.argv) > 1 else 32
ctcf = bench.load_ctcf()
probs = np.asarray(ctcf.count_matrix, dtype=np.float64)
tmp, n = bench.make_tsv_dir(10_000, 2000, 19, probs, min(os.cpu_count(), 128))
import torch
from grafimo_amd.device import DeviceMotif
from grafimo_amd.score_sequences import StreamScan
dm = DeviceMotif.from_motif(ctcf)


def cpu_s():
    r = resource.getrusage(resource.RUSAGE_SELF)
    return r.ru_utime + r.ru_stime


try:
    files = sorted(glob.glob(os.path.join(tmp, "width_19", "*.tsv")))
    StreamScan(dm, files, False, threads, 1e-4, False, True)
    for rest in (0.3, 0.0):
        wall, cpu, par = [], [], []
        for _ in range(7):
            time.sleep(rest)
            c0, t0 = cpu_s(), time.perf_counter()
            sc = StreamScan(dm, files, False, threads, 1e-4, False, True)
            wall.append((time.perf_counter() - t0) * 1e3); cpu.append(cpu_s() - c0); par.append(sc.stats.parse_s * 1e3)
        print(f"{n} rows, {threads} threads, {rest:.1f} s rest between scans: wall median {np.median(wall):6.1f} ms (min {min(wall):6.1f}), "
              f"parse {np.median(par):6.1f} ms, CPU {np.median(cpu):.3f} s per scan = {np.median(cpu) / n * 1e9:.1f} ns per row; "
              f"avx512 scanner {'off' if os.environ.get('GRAFIMO_SCAN_NO_AVX512') else 'on'}", flush=True)
    os.environ["GRAFIMO_SCAN_TRACE"] = "1"
    time.sleep(0.3)
    c0 = cpu_s()
    StreamScan(dm, files, False, threads, 1e-4, False, True)
    print(f"traced scan: process CPU {cpu_s() - c0:.3f} s", flush=True)
    try:
        thp = open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip()
        anon = [l.strip() for l in open("/proc/self/smaps_rollup") if l.startswith("AnonHugePages")]
        print(f"rows, transparent huge pages: {thp}; {anon}", flush=True)
    except OSError as e:
        print("rows, thp state unreadable:", e)
finally:
    shutil.rmtree(tmp, ignore_errors=True)
