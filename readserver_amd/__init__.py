"""readserver_amd -- MI355X-native population-BWT query engine.

Replaces ReadServer's ``src/bwt`` backward-search path (``findInterval -> updateInterval ->
RLEBWT::getOcc``) with hand-written HIP kernels for gfx950 behind a C-ABI (``include/rsbwt.h``,
``lib/librsbwt.so``).  This package is the thin host-side mirror of the reference's ``BWT`` /
``query.h`` interface on top of that library; there is no CPU fallback.
"""
from ._native import build, lib, lib_path, RsbwtError  # noqa: F401
from .bwt import (  # noqa: F401
    BWTInterval,
    GpuBWT,
    ShardSet,
    count_kmers,
    extract_reads,
    extractPostfix,
    extractPrefix,
    find_intervals,
    find_intervals_1mm,
    hits_1mm,
    hits_1mm_batch,
    findInterval,
    query,
    query_batch,
    query_exactmatch,
    query_exactmatch_batch,
    synth_popbwt,
    write_bpi2,
    check_bpi2,
)
