"""Kernel-tuning aid: timeline of the k_sep_u workgroups from a PP_KERNEL_STAMPS build.

    hipcc ... -DPP_KERNEL_STAMPS ...            (diagnostic build of libpp_hip.so)
    PP_STAMPS_OUT=gpurun_out/st.bin python tools/layer_bench.py --layers 7 --ablate 80
    python tools/stamps_timeline.py gpurun_out/st.bin
Stamps are wall_clock64() ticks (100 MHz): 0 entry, 1 taps barrier, 2 first chunk staged, 3 K loop done,
4 stores retired; 5 HW_ID, 6 XCC_ID.
"""
import sys

import numpy as np

raw = np.fromfile(sys.argv[1], dtype=np.int64)
it = raw[4096 * 8:].reshape(64, 64) if raw.size > 4096 * 8 else None
st = raw[:4096 * 8].reshape(-1, 8)
st = st[st[:, 0] != 0]
t = st[:, :5].astype(np.float64)
t0 = t[:, 0].min()
t = (t - t0) / 100.0   # us
hw, xcc = st[:, 5], st[:, 6] & 0xF
cu = ((hw >> 8) & 0xF) | (((hw >> 13) & 0x7) << 4) | (((hw >> 12) & 1) << 7) | (xcc << 8)
print(f"{len(st)} workgroups, {len(np.unique(cu))} distinct CUs, span {t[:, 4].max():.2f} us")
names = ["entry", "taps", "stage0", "loop", "stores"]
for i in range(5):
    print(f"  t[{names[i]:6s}] min {t[:, i].min():7.2f}  p50 {np.median(t[:, i]):7.2f}  max {t[:, i].max():7.2f}")
for i in range(1, 5):
    d = t[:, i] - t[:, i - 1]
    print(f"  d[{names[i - 1]}->{names[i]}] min {d.min():6.2f} p50 {np.median(d):6.2f} p90 {np.percentile(d, 90):6.2f} max {d.max():6.2f}")
# per-CU sequence of the first few CUs
for c in np.unique(cu)[:4]:
    rows = np.where(cu == c)[0]
    rows = rows[np.argsort(t[rows, 0])]
    print(f"CU {c:#x}: " + "  ".join(f"wg{r}[{t[r, 0]:.1f} {t[r, 2]:.1f} {t[r, 3]:.1f} {t[r, 4]:.1f}]" for r in rows))
cnt = np.bincount(np.unique(cu, return_inverse=True)[1])
print("workgroups per CU histogram:", np.bincount(cnt))

if it is not None:   # per-iteration stamps of the first workgroups: deltas in us
    for b in (0, 8, 17, 42):
        row = it[b]
        row = row[row != 0]
        if row.size > 1:
            print(f"wg{b} iteration deltas (us):", " ".join(f"{d / 100.0:.2f}" for d in np.diff(row)))
