"""Print the per-kernel table of a bench.py JSON line (detail.kernel_times_us or the like)."""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"])
det = d.get("detail", {})
for k, v in det.items():
    if isinstance(v, (list, dict)) and "kernel" in k:
        if isinstance(v, dict):
            for kk, vv in v.items(): print("  ", kk, vv)
        else:
            for row in v: print("  ", row)
