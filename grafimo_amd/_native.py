"""ctypes binding of libgrafimo_hip.so (the C ABI declared in include/grafimo_hip.h).

Loading the library does NOT initialise HIP (the reference forks worker processes,
extract_regions.py:128 -- a parent that touched the GPU could not).  There is no CPU
fallback: if the shared object is missing, or a call needs a GPU and none is present,
the call fails loudly.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GRAFIMO_HIP_LIB") or os.path.join(_HERE, "csrc", "libgrafimo_hip.so")

GFM_OK = 0
GFM_ERR_INVALID = -1
GFM_ERR_ASSERT = -2
GFM_ERR_HIP = -3
GFM_ERR_NOMEM = -4
GFM_ERR_NODEVICE = -5
GFM_ERR_IO = -6
GFM_ERR_OVERFLOW = -7
GFM_NO_SELECT = 2**31 - 1
GFM_FLAG_RESET_HITS = 1
GFM_FLAG_CLEAR_HIST = 2
GFM_FLAG_CALLER_ORDERS_REUSE = 4
GFM_MAX_WIDTH = 64
GFM_BEST_ROW_BITS = 44
GFM_GRAPH_FORWARD_ONLY = 1
GFM_TSV_NO_NODEPATH = 1
GFM_WORKSPACE_RING = 4
GFM_HITS_DROP_ZERO_FREQ = 1
GFM_HITS_FIRST_PER_REGION = 2
ABI_VERSION = 12
RANGE = 1000

c_int = ctypes.c_int
c_i32 = ctypes.c_int32
c_i64 = ctypes.c_int64
c_u64 = ctypes.c_uint64
c_double = ctypes.c_double
c_void_p = ctypes.c_void_p
P = ctypes.POINTER

# every exported symbol of include/grafimo_hip.h: name -> (restype, argtypes)
PROTOTYPES = {
    "gfm_abi_version": (c_int, []),
    "gfm_last_error": (ctypes.c_char_p, []),
    "gfm_device_count": (c_int, [P(c_int)]),
    "gfm_set_device": (c_int, [c_int]),
    "gfm_compute_log_odds": (c_int, [c_void_p, c_int, c_void_p, c_void_p]),
    "gfm_scale_pwm": (c_int, [c_void_p, c_int, c_void_p, P(c_int), P(c_int), P(c_int), P(c_double)]),
    "gfm_comp_pval_mat": (c_int, [c_void_p, c_int, c_void_p, c_void_p]),
    "gfm_motif_create": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_double, c_void_p, P(c_void_p)]),
    "gfm_motif_destroy": (None, [c_void_p]),
    "gfm_motif_width": (c_int, [c_void_p]),
    "gfm_motif_table_len": (c_int, [c_void_p]),
    "gfm_motif_score_range": (c_int, [c_void_p, P(c_i32), P(c_i32)]),
    "gfm_motif_tables": (c_int, [c_void_p, c_void_p, c_void_p]),
    "gfm_motif_pvalue_cutoff": (c_int, [c_void_p, c_double, P(c_i32)]),
    "gfm_motif_annotate": (c_int, [c_void_p, c_void_p, c_i64, c_void_p, c_void_p]),
    "gfm_score_kmers": (c_int, [c_void_p, c_void_p, c_i64, c_void_p, c_void_p, c_i32, c_i64,
                                c_void_p, c_i64, c_void_p, ctypes.c_uint32, c_void_p, c_void_p]),
    "gfm_score_kmers_multi": (c_int, [c_void_p, c_int, c_void_p, c_i64, c_void_p, c_void_p, c_void_p, c_i64,
                                      c_void_p, c_void_p, c_void_p, ctypes.c_uint32, c_void_p]),
    "gfm_score_kmers_multi_plan": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "gfm_profile_enable": (c_int, [c_void_p, c_int, c_int]),
    "gfm_profile_read": (c_int, [c_void_p, c_void_p, c_int, P(c_int)]),
    "gfm_profile_mark_tail": (c_int, [c_void_p, c_void_p]),
    "gfm_profile_read_tail": (c_int, [c_void_p, c_void_p, c_int, P(c_int)]),
    "gfm_calibrate_stream": (c_int, [c_int, c_i64, c_int, c_int, P(c_double), P(c_double)]),
    "gfm_qvalue_table": (c_int, [c_void_p, c_void_p, c_double, c_int, c_void_p, c_void_p, c_void_p,
                                 ctypes.c_uint32, c_void_p]),
    "gfm_qvalue_table_multi": (c_int, [c_void_p, c_int, c_void_p, c_double, c_int, c_void_p, c_void_p, c_void_p,
                                       ctypes.c_uint32, c_void_p]),
    "gfm_select_hits": (c_int, [c_void_p, c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_i64, c_void_p,
                                ctypes.c_uint32, c_void_p]),
    "gfm_select_hits_from": (c_int, [c_void_p, c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_i64, c_void_p,
                                     c_void_p, c_i64, c_void_p, c_void_p]),
    "gfm_region_ids": (c_int, [c_void_p, c_i32, c_i64, c_void_p, c_void_p]),
    "gfm_region_best": (c_int, [c_void_p, c_i64, c_void_p, c_i32, c_void_p, c_i32, c_void_p, c_i64, c_void_p, c_void_p]),
    "gfm_locus_max_workspace": (c_i64, [c_i64]),
    "gfm_locus_max": (c_int, [c_void_p, c_i64, c_void_p, c_i32, c_void_p, c_void_p, c_void_p, c_i32, c_void_p, c_void_p,
                              c_i64, c_void_p, c_void_p]),
    "gfm_scan_host": (c_int, [c_void_p, c_void_p, c_i64, c_double, c_int, c_int, c_i64, c_void_p,
                              c_void_p, c_void_p, c_void_p, c_void_p, P(c_i64)]),
    "gfm_tsv_open": (c_int, [P(ctypes.c_char_p), c_int, c_int, c_int, c_int, P(c_void_p), P(c_i64)]),
    "gfm_tsv_count_rows": (c_int, [ctypes.c_char_p, c_int, P(c_i64)]),
    "gfm_tsv_read": (c_int, [c_void_p] * 9),
    "gfm_tsv_name_count": (c_int, [c_void_p]),
    "gfm_tsv_names_bytes": (c_i64, [c_void_p]),
    "gfm_tsv_names": (c_int, [c_void_p, c_void_p, c_void_p]),
    "gfm_tsv_close": (None, [c_void_p]),
    "gfm_scan_tsv": (c_int, [c_void_p, P(ctypes.c_char_p), c_int, c_int, c_int, c_double, c_int, c_int, c_i64,
                             P(c_void_p), P(c_i64), P(c_i64)]),
    "gfm_scan_tsv_begin": (c_int, [c_void_p, c_int, P(ctypes.c_char_p), c_int, c_int, c_int, c_double, c_int, c_int,
                                   c_i64, c_void_p, P(c_void_p), P(c_i64)]),
    "gfm_scan_tsv_finish": (c_int, [c_void_p, c_void_p]),
    "gfm_scan_hits_of": (c_int, [c_void_p, c_int] + [c_void_p] * 12),
    "gfm_scan_hits": (c_int, [c_void_p] * 13),
    "gfm_scan_stats": (c_int, [c_void_p, c_void_p]),
    "gfm_scan_table": (c_void_p, [c_void_p]),
    "gfm_scan_close": (None, [c_void_p]),
    "gfm_scan_release_buffers": (None, []),
    "gfm_graph_create": (c_int, [c_void_p, c_i64, c_i32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                 c_void_p, c_i64, c_void_p, c_i32, P(c_void_p)]),
    "gfm_graph_destroy": (None, [c_void_p]),
    "gfm_graph_plan": (c_int, [c_void_p, c_i32, c_void_p, c_void_p, c_i32, P(c_i64), P(c_i64)]),
    "gfm_graph_plan_windows": (c_int, [c_void_p, c_i32, c_void_p, c_void_p, c_void_p, c_i32, P(c_i64), P(c_i64)]),
    "gfm_graph_emit": (c_int, [c_void_p] * 10),
    "gfm_graph_write_tsvs": (c_int, [c_void_p] * 9 + [c_i64, c_i32, c_i32, c_void_p, P(ctypes.c_char_p), P(ctypes.c_char_p),
                                     ctypes.c_char_p, ctypes.c_uint32, c_int, c_void_p, c_void_p, c_void_p]),
    "gfm_graph_score": (c_int, [c_void_p, c_void_p, c_i32, c_void_p, c_void_p, ctypes.c_uint32, c_i32, c_void_p, c_void_p, c_i64,
                                c_void_p, c_void_p, c_void_p, P(c_i64), c_void_p]),
    "gfm_graph_score_multi": (c_int, [c_void_p, c_void_p, c_i32, c_i32, c_void_p, c_void_p, ctypes.c_uint32, c_void_p, c_void_p, c_void_p,
                                      c_void_p, c_void_p, c_void_p, c_void_p, P(c_i64), c_void_p]),
    "gfm_graph_profile_enable": (c_int, [c_void_p, c_int]),
    "gfm_graph_profile_read": (c_int, [c_void_p, c_void_p, c_int, P(c_int)]),
    "gfm_graph_annotate": (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "gfm_graph_hit_columns": (c_int, [c_void_p, c_i32, c_i32, ctypes.c_double, c_i32, c_i32, c_void_p, c_void_p, c_void_p,
                                      c_void_p, ctypes.c_uint32, P(c_i64)] + [c_void_p] * 10),
    "gfm_graph_hit_columns_start": (c_int, [c_void_p, c_i32, P(c_void_p)]),
    "gfm_graph_hit_columns_wait": (c_int, [c_void_p]),
    "gfm_region_labels": (c_i64, [ctypes.c_char_p, c_void_p, c_void_p, c_i64, c_void_p, c_i64]),
    "gfm_vcf_open": (c_int, [ctypes.c_char_p, ctypes.c_char_p, c_int, c_int, P(c_void_p), P(c_i64), P(c_i32),
                             P(c_i64)]),
    "gfm_vcf_read": (c_int, [c_void_p] * 6),
    "gfm_vcf_ins_bytes": (c_i64, [c_void_p]),
    "gfm_vcf_read_insertions": (c_int, [c_void_p] * 4),
    "gfm_vcf_close": (None, [c_void_p]),
}


class HitColumnsJob(ctypes.Structure):
    """gfm_hit_columns_job_t"""
    _fields_ = [("h_ptable", c_void_p), ("h_recs", c_void_p), ("n_recs", c_void_p), ("h_entry_of", c_void_p),
                ("region_base", c_void_p), ("o_start", c_void_p), ("o_stop", c_void_p), ("o_freq", c_void_p),
                ("o_region", c_void_p), ("o_score", c_void_p), ("o_pvalue", c_void_p), ("o_qvalue", c_void_p),
                ("o_strand", c_void_p), ("o_ref", c_void_p), ("o_kmers", c_void_p), ("offset", ctypes.c_double),
                ("n_out", c_i64), ("table_len", c_i32), ("scale", c_i32), ("width", c_i32), ("n_parts", c_i32),
                ("flags", ctypes.c_uint32), ("status", c_i32)]


assert ctypes.sizeof(HitColumnsJob) == 160


class ScanStats(ctypes.Structure):
    """gfm_scan_stats_t"""
    _fields_ = [("n_rows", c_i64), ("n_hits", c_i64), ("n_chunks", c_i64), ("h2d_bytes", c_i64),
                ("total_s", c_double), ("parse_s", c_double), ("h2d_s", c_double), ("tail_s", c_double),
                ("parse_threads", c_i32), ("reserved", c_i32)]


class TsvWriteStats(ctypes.Structure):
    """gfm_tsv_write_stats_t"""
    _fields_ = [("n_rows", c_i64), ("n_files", c_i64), ("bytes", c_i64), ("total_s", c_double), ("copy_s", c_double),
                ("format_s", c_double), ("threads", c_i32), ("reserved", c_i32)]


class NativeError(RuntimeError):
    """A libgrafimo_hip call failed; .code holds the GFM_ERR_* value."""

    def __init__(self, code, msg):
        super().__init__(f"libgrafimo_hip error {code}: {msg}")
        self.code = code
        self.msg = msg


_lib = None


def _preload_hip_runtime():
    """One process must hold ONE HIP runtime.  PyTorch-ROCm wheels bundle their own
    libamdhip64.so (SONAME libamdhip64.so.7, the same as /opt/rocm's) but link it by file name,
    so if /opt/rocm's copy were loaded first torch would load a second runtime next to it and
    every later HIP call would misbehave.  When torch is installed, load ITS runtime first
    (without importing torch): our NEEDED libamdhip64.so.7 then binds to that copy by SONAME.
    GRAFIMO_HIP_RUNTIME=/path/to/libamdhip64.so overrides; "system" skips the preload."""
    import importlib.util
    override = os.environ.get("GRAFIMO_HIP_RUNTIME", "")
    if override == "system":
        return None
    path = override
    if not path:
        try:
            spec = importlib.util.find_spec("torch")
        except (ImportError, ValueError):
            spec = None
        if spec is not None and spec.origin:
            cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
            if os.path.exists(cand):
                path = cand
    if path:
        return ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
    return None


def lib():
    """The loaded library (loads on first use; raises if it was not built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with grafimo_amd/csrc/build.sh "
                "(or python -c 'import __graft_entry__ as g; g.build()'). "
                "grafimo_amd has no CPU fallback."
            )
        _preload_hip_runtime()
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(L, name)  # AttributeError if the ABI lost a symbol
            fn.restype = res
            fn.argtypes = args
        if L.gfm_abi_version() != ABI_VERSION:
            raise ImportError("libgrafimo_hip.so ABI version mismatch")
        _lib = L
    return _lib


def check(rc):
    if rc != GFM_OK:
        raise NativeError(rc, lib().gfm_last_error().decode("utf-8", "replace"))


def ptr(a):
    """Address of a numpy array / int address / None."""
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return a.ctypes.data
    return int(a)


def device_count():
    n = c_int(0)
    check(lib().gfm_device_count(ctypes.byref(n)))
    return n.value


def c_paths(paths):
    """A list of str paths as `const char *const *` for the library: ONE encoded blob and a numpy array of addresses into
    it (10 000 paths: 1.5 ms; a ctypes array of c_char_p built from 10 000 bytes objects: 6 ms -- a fifth of the scan that
    follows).  -> (pointer, keepalive): keep `keepalive` referenced until the call returns."""
    import numpy as np
    if len(paths) == 0:
        return ctypes.cast(None, P(ctypes.c_char_p)), None
    blob = ("\0".join(paths) + "\0").encode()
    buf = np.frombuffer(blob, dtype=np.uint8)
    ends = np.flatnonzero(buf == 0)
    if len(ends) != len(paths):
        raise ValueError("a path holds a NUL character")
    starts = np.empty(len(ends), dtype=np.uint64)
    starts[0] = 0
    starts[1:] = ends[:-1] + 1
    ptrs = starts + np.uint64(buf.ctypes.data)
    return ctypes.cast(ptrs.ctypes.data, P(ctypes.c_char_p)), (blob, buf, ptrs)
