"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into HBM bytes per launch per kernel.

    python tools/pmc_summary.py gpurun_out/pmc_f/pmc_counter_collection.csv \
                                gpurun_out/pmc_w/pmc_counter_collection.csv profiles/r01_pmc_traffic.json
Corrections as /opt/skills/guides/MI355X_MICROARCH.md "HBM" prescribes for gfx950: both counters are
in KB; FETCH_SIZE reports exactly half of the bytes of wide (16 B/lane) coalesced reads -> doubled;
WRITE_SIZE is exact for 16 B/lane streaming stores.  The two counters come from separate passes.
"""
import collections
import csv
import json
import re
import sys


def avg_per_kernel(path):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        agg[r["Kernel_Name"]][0] += float(r["Counter_Value"])
        agg[r["Kernel_Name"]][1] += 1
    return {k: v[0] / v[1] for k, v in agg.items()}, {k: v[1] for k, v in agg.items()}


def short(name):
    m = re.match(r"(?:void )?([A-Za-z_0-9]+(?:<[^>]*>)?)", name)
    return m.group(1).replace(" ", "") if m else name


fetch, nf = avg_per_kernel(sys.argv[1])
write, nw = avg_per_kernel(sys.argv[2])
out = {}
for k in sorted(set(fetch) | set(write)):
    f_b = fetch.get(k, 0.0) * 1024 * 2.0
    w_b = write.get(k, 0.0) * 1024
    out[short(k)] = {"fetch_bytes_per_launch": f_b, "write_bytes_per_launch": w_b,
                     "hbm_bytes_per_launch": f_b + w_b, "launches_sampled": nf.get(k, 0)}
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import csrc_sha16  # noqa: E402
json.dump({"git_head": os.environ.get("PP_GIT_HEAD", "unknown"), "csrc_sha16": csrc_sha16(),
           "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of "
                   "`bench.py --steps 4 --warmup 1 --no-cpu-baseline --inflight 1`; KB -> bytes, FETCH_SIZE x2 "
                   "(gfx950 correction for 16 B/lane reads)", "kernels": out}, open(sys.argv[3], "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"])[:10]:
    print(f"{k:32s} {v['hbm_bytes_per_launch'] / 1e6:9.1f} MB/launch  (fetch {v['fetch_bytes_per_launch'] / 1e6:.1f} + write {v['write_bytes_per_launch'] / 1e6:.1f})")
