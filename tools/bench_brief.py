import json, sys
d = json.loads(open(sys.argv[1]).read())
r = d["roofline"]
print(f"{d['value']:.4g} q/s  kernel {r['kernel_ms']:.3f} ms  step {d['ms_per_step']:.3f} ms  frac {r['frac']:.3f}")
