#!/usr/bin/env python3
"""Cross-check of the wave-cooperative extraction kernels (extract_lines.hip) on a full-size shard,
where the CPU oracle cannot follow: readserver_amd/selfcheck.py replays the same walks step by step
with the class-BWT mirrors and compares every row's prefix and postfix character by character.
Prints one JSON line; exit code 1 on any difference.
usage: tools/check_extract_at_scale.py [run_bytes=2e10] [rows=2000000] [long_runs=0] [window_span=0]"""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import readserver_amd as rsb  # noqa: E402
from readserver_amd import selfcheck  # noqa: E402

R = int(float(sys.argv[1])) if len(sys.argv) > 1 else 20000000000
NR = int(float(sys.argv[2])) if len(sys.argv) > 2 else 2000000
LONG = int(sys.argv[3]) if len(sys.argv) > 3 else 0  # 1: the long-run stream, 2: the population stream
SPAN = int(sys.argv[4]) if len(sys.argv) > 4 else 0
L = rsb.lib()
dev = torch.device("cuda", 0)
d_runs = torch.empty(R, dtype=torch.uint8, device=dev)
seed = ((1 << 63) if LONG == 1 else (1 << 62) if LONG == 2 else 0) | 1000003
if L.rsbwt_synth_runs_dev(C.c_void_p(d_runs.data_ptr()), R, seed, 0, None) != 0:
    raise RuntimeError(L.rsbwt_last_error().decode())
torch.cuda.synchronize()
g = rsb.GpuBWT(device_runs=(d_runs.data_ptr(), R), ktab_depth=None, window_span=SPAN)
del d_runs
torch.cuda.empty_cache()
n = g.getBWLen()
rows = np.random.default_rng(99).integers(0, n, NR, dtype=np.uint64)
res = selfcheck.extraction_vs_mirrors(g, rows, stride=1024)
res = dict({"run_bytes": R, "symbols": int(n), "window_span": g.window_span(), "far_lines": g.far_lines()}, **res)
print(json.dumps(res))
g.close()
sys.exit(1 if res["rows_differing"] else 0)
