import sys, os, json
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import pp_amd as pp
batch = int(sys.argv[1])
cfg = pp.config.pedestrian_d435i_config(batch)
d = pp.config.Derived(cfg)
tr = pp.Trainer(cfg, pp.weights.init_weights(d, seed=7), max_batch=batch, max_points_per_frame=16384, learning_rate=2e-4, weight_decay=1e-4)
rng = np.random.default_rng(50)
frames = [pp.synth.d435i_cloud(5000 + i, 16384) for i in range(batch)]
labels = rng.choice([-1, 0, 0, 0, 0], size=(batch, d.num_anchors)).astype(np.int32)
reg = np.zeros((batch, d.num_anchors, 7), np.float32)
for b in range(batch):
    pos = rng.choice(d.num_anchors, 30, replace=False); labels[b, pos] = 1
    reg[b, pos] = rng.normal(0, 0.4, (30, 7)).astype(np.float32)
for _ in range(3): tr.forward_backward(frames, labels, reg)
tr.engine.set_profiling(True)
acc = {}
R = 5
for _ in range(R):
    tr.forward_backward(frames, labels, reg)
    for name, ms in tr.engine.kernel_times():
        a = acc.setdefault(name, [0.0, 0]); a[0] += ms; a[1] += 1
tot = 0
for k, (ms, n) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
    print(f"{ms / R * 1e3:9.1f} us  x{n // R:3d}  {k}")
    tot += ms / R
print("sum ms", tot)
