"""Tuning aid: single-frame time of the voxeliser / PFN / post-process kernels against the point count."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pp_amd as pp  # noqa: E402

cfg = pp.config.pedestrian_d435i_config(1)
e = pp.Engine(cfg, max_batch=1, max_points_per_frame=20000)
e.load_weights(pp.weights.init_weights(e.d, seed=7))
calib = pp.synth.default_calib()
for n in (1024, 2048, 4096, 8192, 16384):
    e.upload([pp.synth.d435i_cloud(0, n)], calib[0][None], calib[1][None])
    for _ in range(5):
        e.detect_async(); e.sync()
    e.set_profiling(True)
    tot = {}
    for _ in range(20):
        e.detect_async(); e.sync()
        for tag, ms in e.kernel_times():
            tot[tag] = tot.get(tag, 0) + ms / 20
    e.set_profiling(False)
    keep = [k for k in tot if not k.startswith(("k_sep", "k_deconv"))]
    print(n, "  ".join(f"{k.split(':')[0]} {tot[k] * 1e3:.1f}" for k in keep))
