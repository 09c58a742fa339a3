#!/usr/bin/env python3
"""Time of building the index of one shard of the bench's run stream (synthesis excluded), for A/B runs of builder
variants (RSBWT_LIB=tools/bin/librsbwt_X.so).   usage: tools/build_time_probe.py [run_bytes=2e10] [stream=pop|mixed|long]"""
import ctypes as C
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import readserver_amd as rsb  # noqa: E402

R = int(float(sys.argv[1])) if len(sys.argv) > 1 else 20000000000
style = {"pop": 1 << 62, "mixed": 0, "long": 1 << 63}[sys.argv[2] if len(sys.argv) > 2 else "pop"]
L = rsb.lib()
d = torch.empty(R, dtype=torch.uint8, device="cuda:0")
assert L.rsbwt_synth_runs_dev(C.c_void_p(d.data_ptr()), R, style | 1000003, 0, None) == 0
torch.cuda.synchronize()
ts = []
for _ in range(2):
    t0 = time.time()
    g = rsb.GpuBWT(device_runs=(d.data_ptr(), R), ktab_depth=None)
    ts.append(time.time() - t0)
    info = {"window_span": g.window_span(), "lines": g.num_lines(), "far_lines": g.far_lines()}
    g.close()
print(json.dumps({"lib": rsb.lib_path(), "run_bytes": R, "build_s": ts, **info}))
