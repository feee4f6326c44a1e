"""Tuning aid (-DPP_DECONV_ABLATE build): shader-clock stamps of k_deconv_r's phases for the first 64 workgroups.
    PP_HIP_LIB=libpp_hip_abl.so python tools/deconv_stamps.py [LAYER=18]"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pp_amd as pp
layer = int(sys.argv[1]) if len(sys.argv) > 1 else 18
B = 64
eng = pp.Engine(pp.config.pedestrian_d435i_config(B), max_batch=B, max_points_per_frame=20000)
eng.load_weights(pp.weights.init_weights(eng.d, seed=7))
eng.detect([pp.synth.d435i_cloud(i) for i in range(B)])
out = os.path.join(tempfile.gettempdir(), "stamps.bin")
os.environ["PP_STAMPS_OUT"] = out
ms = eng.bench_layer(layer, B, reps=5, ablate=64)
st = np.fromfile(out, dtype=np.int64)[4096 * 8:].reshape(64, 64)
print(eng.layer_tags()[layer], f"{ms * 1e3:.1f} us")
t0 = st[:, 0:1]
rel = np.where(st > 0, st - t0, -1)
for blk in (0, 1, 9, 33, 63):
    print("wg", blk, "prologue", rel[blk, 1], "units:", [list(rel[blk, 2 + 8 * u:9 + 8 * u]) for u in range(6) if st[blk, 2 + 8 * u] > 0])
ok = st[:, 2] > 0
units = []
for u in range(6):
    sel = ok & (st[:, 2 + 8 * u] > 0) & (st[:, 8 + 8 * u] > 0)
    if sel.any():
        d = st[sel]
        units.append((u, int(sel.sum()), np.median(d[:, 3 + 8 * u] - d[:, 2 + 8 * u]), np.median(d[:, 4 + 8 * u] - d[:, 3 + 8 * u]),
                      np.median(d[:, 5 + 8 * u] - d[:, 4 + 8 * u]), np.median(d[:, 6 + 8 * u] - d[:, 5 + 8 * u]),
                      np.median(d[:, 7 + 8 * u] - d[:, 6 + 8 * u]), np.median(d[:, 8 + 8 * u] - d[:, 7 + 8 * u])))
print("median cycles per unit: (unit, n, input tile, n-tile 0, 1, 2, 3, head store)")
for r in units:
    print("  ", r)
last = np.array([st[b, 8 + 8 * max(u for u in range(6) if st[b, 8 + 8 * u] > 0)] - st[b, 0] for b in range(64) if ok[b]])
print("workgroup lifetime cycles: median %d min %d max %d" % (np.median(last), last.min(), last.max()))
