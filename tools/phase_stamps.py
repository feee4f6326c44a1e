"""Tuning aid (PP_KERNEL_STAMPS build): shader-clock stamps of the phases of ONE K-chunk iteration of k_sep_u
(MFMA | staging | load issue | barrier) for the first 64 workgroups.  PP_HIP_LIB=<stamps lib> python tools/phase_stamps.py LAYER"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pp_amd as pp
layer = int(sys.argv[1]) if len(sys.argv) > 1 else 7
B = 64
eng = pp.Engine(pp.config.pedestrian_d435i_config(B), max_batch=B, max_points_per_frame=20000)
eng.load_weights(pp.weights.init_weights(eng.d, seed=7))
eng.detect([pp.synth.d435i_cloud(i) for i in range(B)])
out = os.path.join(tempfile.gettempdir(), "stamps.bin")
os.environ["PP_STAMPS_OUT"] = out
ms = eng.bench_layer(layer, B, reps=5, ablate=64)
st = np.fromfile(out, dtype=np.int64)[4096 * 8:].reshape(64, 64)
ph = st[:, 40:45]
ok = (ph > 0).all(axis=1)
d = np.diff(ph[ok], axis=1)
print(eng.layer_tags()[layer], f"{ms*1e3:.1f} us; workgroups sampled {ok.sum()}")
print("phase cycles (median over workgroups): MFMA %d | stage %d | issue %d | epilogue+barrier %d | sum %d" %
      (*np.median(d, axis=0), np.median(d.sum(axis=1))))
it = st[:, 1:40]
per = np.diff(it, axis=1)
per = per[(it[:, 1:] > 0) & (it[:, :-1] > 0)]
print("double-iteration (wall clock ticks of 10 ns), median:", np.median(per) if per.size else None)
