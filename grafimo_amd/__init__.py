"""grafimo_amd -- MI355X-native k-mer scoring path of GRAFIMO.

Host orchestration is Python; every number on the hot path is produced by
libgrafimo_hip.so (hand-written HIP for gfx950, C ABI in include/grafimo_hip.h)
called through ctypes.  Importing this package does not touch the GPU.

Public surface mirrors the reference (paths relative to /root/reference/src/grafimo/):
  motif.Motif                          motif.py:18
  motif_processing.*                   the six callables of motif_processing.pyx
  motif_ops.*                          parsers + process_motif_for_logodds + scale_pwm
  score_sequences.compute_results      score_sequences.py:44
  resultsTmp.ResultTmp                 resultsTmp.py:15
"""
__version__ = "0.1.0"

from . import _native  # noqa: F401  (does not load the .so until first use)
