#!/usr/bin/env python3
"""What a set that sizes its own k-mer tables gets at full size: S shards of R run bytes resident, then
rsbwt_set_attach_ktabs_format(set, 0, AUTO) -- the call rsbwt_set_open makes for rsbwt_service.  One JSON line.
usage: tools/set_auto_tables_probe.py [S=8] [run_bytes=2e10]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C  # noqa: E402

import torch  # noqa: E402

import readserver_amd as rsb  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 8
R = int(float(sys.argv[2])) if len(sys.argv) > 2 else 20000000000
L = rsb.lib()
shards = []
for s in range(S):
    d = torch.empty(R, dtype=torch.uint8, device="cuda:0")
    assert L.rsbwt_synth_runs_dev(C.c_void_p(d.data_ptr()), R, (1 << 62) | 77, 0, None) == 0
    torch.cuda.synchronize()
    shards.append(rsb.GpuBWT(device_runs=(d.data_ptr(), R), ktab_depth=None))
    del d
    torch.cuda.empty_cache()
ss = rsb.ShardSet(shards)
free0 = torch.cuda.mem_get_info(0)[0]
assert L.rsbwt_set_attach_ktabs_format(ss._s, 0, 2) == 0
fmt, nbytes, left = shards[0].ktab_info()
print(json.dumps({"shards": S, "run_bytes_per_shard": R, "free_hbm_before_tables": free0, "free_hbm_after": torch.cuda.mem_get_info(0)[0],
                  "ktab_depth": shards[0].ktab_depth(), "ktab_format": "grouped" if fmt == 1 else "plain", "ktab_bytes_per_shard": nbytes,
                  "untabulated_fraction": left / 4 ** shards[0].ktab_depth()}))
