"""Where the batch-1 wall latency goes on the host side: time inside upload_async / detect_async / sync."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pp_amd as pp
e = pp.Engine(pp.config.pedestrian_d435i_config(1), max_batch=1, max_points_per_frame=20000)
e.load_weights(pp.weights.init_weights(e.d, seed=7))
calib = pp.synth.default_calib()
e.set_calib(calib[0][None], calib[1][None], 1)
st = [e.staging([pp.synth.d435i_cloud(i)]) for i in range(16)]
rows = []
for i in range(208):
    t0 = time.perf_counter(); e.upload_async(st[i % 16]); t1 = time.perf_counter()
    e.detect_async(); t2 = time.perf_counter(); e.sync(); t3 = time.perf_counter()
    rows.append((t1 - t0, t2 - t1, t3 - t2, t3 - t0))
r = np.array(rows[8:]) * 1e6
print("median us: upload_async %.1f | detect_async %.1f | sync %.1f | total %.1f" % tuple(np.median(r, axis=0)))
ts = []
for i in range(100):
    e.timer_start(); e.detect_async(); ts.append(e.timer_stop())
print("GPU time of the graph alone (events): %.1f us" % (np.median(ts) * 1e3))
