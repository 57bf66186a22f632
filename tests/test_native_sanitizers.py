"""The host-side readers of libgrafimo_hip.so (TSV ingest with its AVX2 field splitter and counting pass, the VCF reader)
rebuilt with g++ -fsanitize=address,undefined and driven over the reference's fixtures and hundreds of mutated copies
(tests/native/reader_fuzz.cpp): no out-of-bounds access, no undefined behaviour, and for every file the parser accepts
the counting pass of the streamed scan equals the parsed row count.  (GPU AddressSanitizer is not available on the pool:
sanitizers run on the CPU build only.)"""
import gzip
import os
import shutil
import subprocess

import pytest

from conftest import REF_DATA, ROOT


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_readers_under_asan_and_ubsan(tmp_path):
    csrc = os.path.join(ROOT, "grafimo_amd", "csrc")
    exe = str(tmp_path / "reader_fuzz")
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
           "-fno-sanitize-recover=undefined", f"-I{os.path.join(ROOT, 'include')}", f"-I{csrc}",
           os.path.join(ROOT, "tests", "native", "reader_fuzz.cpp"), os.path.join(csrc, "tsv_ingest.cpp"),
           os.path.join(csrc, "vcf_ingest.cpp"), os.path.join(csrc, "gfm_workers.cpp"), "-lpthread", "-lz", "-o", exe]
    build = subprocess.run(cmd, capture_output=True, text=True)
    if build.returncode != 0 and "sanitize" in build.stderr:
        pytest.skip("this g++ has no sanitizer runtime")
    assert build.returncode == 0, build.stderr[-3000:]
    vcf = tmp_path / "test.vcf"
    with gzip.open(os.path.join(REF_DATA, "test.vcf.gz"), "rb") as src, open(vcf, "wb") as dst:
        dst.write(src.read())
    work = tmp_path / "work"
    work.mkdir()
    for env_extra in ({}, {"GRAFIMO_SCAN_NO_AVX512": "1"}):        # the AVX-512 forms where the CPU has them, then the AVX2 / scalar ones
        run = subprocess.run([exe, os.path.join(REF_DATA, "width_19", "scoring_test_input.tsv"), str(vcf), str(work)],
                             capture_output=True, text=True, timeout=600,
                             env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", **env_extra))
        assert run.returncode == 0, (run.stdout[-2000:], run.stderr[-4000:])
        assert "tsv:" in run.stdout and "vcf:" in run.stdout and "MISMATCH" not in run.stdout
        assert "edges:" in run.stdout and "sort: ok" in run.stdout
        parsed = int(run.stdout.split("tsv:")[1].split("parsed")[0])
        assert parsed > 20                                  # some mutated files are still well-formed


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
@pytest.mark.parametrize("sanitizer", ["address,undefined", "thread"])
def test_hit_table_and_its_threaded_run_under_sanitizers(tmp_path, sanitizer):
    """csrc/hit_table.cpp -- gfm_graph_hit_columns, the jobs of a motif set on the library's host threads
    (gfm_graph_hit_columns_start / _wait) while the caller goes on, gfm_region_labels -- on random records
    (tests/native/hit_table_threads.cpp): no out-of-bounds access or undefined behaviour, no data race (ThreadSanitizer), and
    every job's columns and status equal the synchronous call's."""
    csrc = os.path.join(ROOT, "grafimo_amd", "csrc")
    exe = str(tmp_path / "hit_table_threads")
    cmd = ["g++", "-O1", "-g", "-std=c++17", f"-fsanitize={sanitizer}", "-fno-omit-frame-pointer",
           f"-I{os.path.join(ROOT, 'include')}", f"-I{csrc}", os.path.join(ROOT, "tests", "native", "hit_table_threads.cpp"),
           os.path.join(csrc, "hit_table.cpp"), os.path.join(csrc, "gfm_workers.cpp"), "-lpthread", "-o", exe]
    if "undefined" in sanitizer:
        cmd.insert(6, "-fno-sanitize-recover=undefined")
    build = subprocess.run(cmd, capture_output=True, text=True)
    if build.returncode != 0 and "sanitize" in build.stderr:
        pytest.skip("this g++ has no such sanitizer runtime")
    assert build.returncode == 0, build.stderr[-3000:]
    for seed in (1, 2, 3):
        run = subprocess.run([exe, "8", str(seed)], capture_output=True, text=True, timeout=600,
                             env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", TSAN_OPTIONS="halt_on_error=1"))
        if run.returncode != 0 and "unexpected memory mapping" in run.stderr:
            pytest.skip("ThreadSanitizer cannot map its shadow memory in this container")
        assert run.returncode == 0, (run.stdout[-2000:], run.stderr[-4000:])
        assert "threaded == synchronous" in run.stdout and "MISMATCH" not in run.stdout
        assert "WARNING: ThreadSanitizer" not in run.stderr and "ERROR: AddressSanitizer" not in run.stderr
