"""Per-kernel averages of the SQ / LDS / TA counter passes of tools/pmc_r03.sh.

    python tools/pmc_sq_summary.py gpurun_out/r03/pmc profiles/r03_pmc_sq.json
Every pass is `rocprofv3 --pmc <4 counters> --kernel-trace` over the same command; a kernel's value is the mean
over its launches in that pass.  Derived per kernel (MI355X_MICROARCH.md: SQ_WAVE_CYCLES / SQ_WAIT_* /
SQ_ACTIVE_INST_* count quad-cycles summed over waves, SQ_BUSY_CYCLES quad-cycles per SE-instance,
SQ_VALU_MFMA_BUSY_CYCLES cycles summed over SIMDs, GRBM_GUI_ACTIVE cycles summed over the 8 XCDs):
  kernel_cycles     = GRBM_GUI_ACTIVE / 8
  mfma_busy_frac    = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * kernel_cycles)
  waves_per_simd    = SQ_WAVE_CYCLES * 4 / (1024 * kernel_cycles)
  issue / stall / parked shares of a wave's life = ACTIVE_INST_ANY, WAIT_INST_ANY, WAIT_ANY over WAVE_CYCLES
"""
import collections
import csv
import glob
import json
import os
import re
import subprocess
import sys


def short(name):
    m = re.match(r"(?:void )?([A-Za-z_0-9]+(?:<[^>]*>)?)", name)
    return m.group(1).replace(" ", "") if m else name


root, out_path = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in sorted(glob.glob(os.path.join(root, "p*", "**", "*counter_collection.csv"), recursive=True)):
    for r in csv.DictReader(open(f)):
        a = agg[short(r["Kernel_Name"])][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"])
        a[1] += 1
out = {}
for k, d in agg.items():
    v = {c: s / n for c, (s, n) in d.items()}
    v["launches_sampled"] = max(n for (_, n) in d.values())
    if "GRBM_GUI_ACTIVE" in v and v["GRBM_GUI_ACTIVE"] > 0:
        kc = v["GRBM_GUI_ACTIVE"] / 8.0
        v["kernel_cycles"] = kc
        if "SQ_VALU_MFMA_BUSY_CYCLES" in v:
            v["mfma_busy_frac"] = v["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * kc)
        if "SQ_WAVE_CYCLES" in v:
            v["waves_per_simd"] = v["SQ_WAVE_CYCLES"] * 4.0 / (1024.0 * kc)
    if v.get("SQ_WAVE_CYCLES", 0) > 0:
        for c, key in (("SQ_ACTIVE_INST_ANY", "issue_share"), ("SQ_WAIT_INST_ANY", "stall_share"),
                       ("SQ_WAIT_ANY", "parked_share")):
            if c in v:
                v[key] = v[c] / v["SQ_WAVE_CYCLES"]
    out[k] = {c: (round(x, 4) if isinstance(x, float) and x < 10 else round(x)) for c, x in v.items()}
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import csrc_sha16  # noqa: E402
head = os.environ.get("PP_GIT_HEAD", "")
if not head:
    try:
        head = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], stderr=subprocess.DEVNULL).decode().strip()
    except Exception:
        head = "unknown"
json.dump({"note": __doc__.split("\n\n")[0] + "  Command: " + os.environ.get("PMC_CMD", "bench.py --plain --steps 4 --warmup 1 --inflight 1"),
           "git_head": head, "csrc_sha16": csrc_sha16(), "kernels": out}, open(out_path, "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1].get("kernel_cycles", 0))[:14]:
    print(f"{k:28s} cyc {v.get('kernel_cycles', 0):>8} mfma {v.get('mfma_busy_frac', 0):.3f} w/simd {v.get('waves_per_simd', 0):.2f} "
          f"issue {v.get('issue_share', 0):.2f} stall {v.get('stall_share', 0):.2f} parked {v.get('parked_share', 0):.2f} "
          f"valu {v.get('SQ_INSTS_VALU', 0)} lds {v.get('SQ_INSTS_LDS', 0)} conf {v.get('SQ_LDS_BANK_CONFLICT', 0)} ldsact {v.get('SQ_LDS_IDX_ACTIVE', 0)}")
