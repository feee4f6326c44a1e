import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
import pp_amd as pp
cfg = pp.config.pedestrian_d435i_config(1)
e = pp.Engine(cfg, max_batch=1, max_points_per_frame=20000)
e.load_weights(pp.weights.init_weights(e.d, seed=7))
fr = [pp.synth.d435i_cloud(0)]
calib = pp.synth.default_calib()
e.upload(fr, calib[0][None], calib[1][None])
for _ in range(5): e.detect_async(); e.sync()
e.set_profiling(True)
tot = {}
for _ in range(20):
    e.detect_async(); e.sync()
    for tag, ms in e.kernel_times():
        tot[tag] = tot.get(tag, 0) + ms / 20
e.set_profiling(False)
for k, v in tot.items(): print(f"{k:45s} {v*1e3:7.1f} us")
print("sum", sum(tot.values()) * 1e3, "us")
ts = []
for _ in range(50):
    e.timer_start(); e.detect_async(); ts.append(e.timer_stop())
print("p50 gpu ms", np.median(ts))
ts = []
for _ in range(50):
    t0 = time.perf_counter(); e.detect_async(); e.sync(); ts.append((time.perf_counter() - t0) * 1e3)
print("p50 wall ms (launch->sync)", np.median(ts))
