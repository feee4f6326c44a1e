"""Host logic of the product package + the C-ABI surface (no compute: no GPU here)."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT, load_golden


def test_product_anchors_match_reference(pp):
    g = load_golden("ref_anchors.npz")
    c = pp.config
    for name, cfg in (("A", c.pedestrian_d435i_config()), ("T", c.tiny_config())):
        d = c.Derived(cfg)
        a = pp.anchors.build_anchors(d)
        assert a.dtype == np.float32 and np.array_equal(a, g[name + "_anchors"])
    d = c.Derived(c.kitti_shaped_config())
    a = pp.anchors.build_anchors(d)
    assert np.array_equal(a[::997], g["K_rows"]) and list(a.shape) == list(g["K_shape"])


def test_product_anchor_cells_reproduce_reference_mask(pp):
    """cells + integral image == fused_get_anchors_area of the reference."""
    gv, gm = load_golden("ref_voxel.npz"), load_golden("ref_mask.npz")
    d = pp.config.Derived(pp.config.pedestrian_d435i_config())
    a = pp.anchors.build_anchors(d)
    cells = pp.anchors.build_anchor_cells(a, d)
    for case in ("a2k", "a16k"):
        occ = np.zeros((d.ny, d.nx), np.int64)
        np.add.at(occ, (gv[case + "_coors"][:, 1], gv[case + "_coors"][:, 2]), 1)
        I = occ.cumsum(0).cumsum(1)
        area = I[cells[:, 3], cells[:, 2]] - I[cells[:, 3], cells[:, 0]] - I[cells[:, 1], cells[:, 2]] + I[cells[:, 1], cells[:, 0]]
        assert np.array_equal(area.astype(np.float32), gm[case + "_area"])


def test_config_derivation(pp):
    c = pp.config
    d = c.Derived(c.pedestrian_d435i_config())
    assert (d.nx, d.ny, d.nz) == (80, 64, 2)          # round(1.5) == 2: SURVEY fact 10
    assert (d.head_h, d.head_w, d.num_anchors) == (64, 80, 10240)
    assert d.pfn_in == 8 and d.concat_channels == 384
    k = c.Derived(c.kitti_shaped_config())
    assert (k.nx, k.ny, k.nz, k.head_h, k.head_w, k.num_anchors) == (432, 496, 1, 248, 216, 107136)
    two = c.Derived(c.kitti_shaped_config(num_class=2))        # 2 * (7 + 2 + 2) = 22 head columns
    assert two.num_class == 2 and two.num_anchor_per_loc == 2
    bad = c.pedestrian_d435i_config()
    bad["model"]["second"]["num_class"] = 8                    # 2 * (7 + 8 + 2) = 34 > 32 head columns
    with pytest.raises(NotImplementedError):
        c.Derived(bad)
    bad = c.pedestrian_d435i_config()
    bad["model"]["second"]["use_multi_class_nms"] = True       # TF stub in the reference
    with pytest.raises(NotImplementedError):
        c.Derived(bad)
    nd = c.pedestrian_d435i_config()
    nd["model"]["second"]["use_direction_classifier"] = False
    nd["model"]["second"]["voxel_feature_extractor"]["with_distance"] = True
    dd = c.Derived(nd)
    assert dd.pfn_in == 9 and not dd.use_direction_classifier
    assert "rpn/conv_dir_cls/kernel" not in pp.weights.init_weights(dd, seed=1)
    bad = c.pedestrian_d435i_config()
    bad["model"]["second"]["rpn"]["upsample_strides"] = [1, 2, 2]
    with pytest.raises(ValueError):
        c.Derived(bad)


def test_reference_yaml_schema_loads(pp, tmp_path):
    y = tmp_path / "cfg.yaml"
    import yaml
    y.write_bytes(b"\xef\xbb\xbf" + yaml.safe_dump(pp.config.pedestrian_d435i_config()).encode())
    d = pp.config.Derived(pp.config.load_yaml(str(y)))
    assert d.max_points == 50 and d.max_voxels == 12000


def test_weights_shapes_and_roundtrip(pp, tmp_path):
    d = pp.config.Derived(pp.config.pedestrian_d435i_config())
    w = pp.weights.init_weights(d, seed=7)
    pp.weights.check_weights(d, w)
    assert w["rpn/block1/0/depthwise_kernel"].shape == (3, 3, 128, 1)
    assert w["rpn/deconv3/kernel"].shape == (4, 4, 128, 256)
    assert w["rpn/conv_box/kernel"].shape == (1, 1, 384, 14)
    n_params = sum(v.size for v in w.values())
    assert 1.0e6 < n_params < 1.25e6                   # SURVEY: ~1.10 M parameters
    p = str(tmp_path / "w.npz")
    pp.weights.save_npz(p, w)
    w2 = pp.weights.load_npz(p)
    assert all(np.array_equal(w[k], w2[k]) for k in w)
    del w2["pfn/bn/beta"]
    with pytest.raises(ValueError):
        pp.weights.check_weights(d, w2)


def test_synth_cloud_is_seeded_and_shaped(pp):
    a = pp.synth.d435i_cloud(3)
    b = pp.synth.d435i_cloud(3)
    assert a.shape == (16384, 3) and a.dtype == np.float32 and np.array_equal(a, b)
    assert (a[:, 2] >= 1.0).any(), "second z-cell must be populated"
    k = pp.synth.kitti_cloud(0)
    assert k.shape == (20000, 4)


def test_cabi_exports_every_declared_symbol(pp, hip_lib):
    hdr = open(os.path.join(ROOT, "include", "pp_hip.h")).read()
    declared = set(re.findall(r"^\s*(?:int|const char\*)\s+(pp_[a-z_0-9]+)\s*\(", hdr, flags=re.M))
    assert declared == set(pp._lib.EXPORTS), declared ^ set(pp._lib.EXPORTS)
    for name in declared:
        assert hasattr(hip_lib, name), name
    assert hip_lib.pp_abi_version() == 4


def test_struct_layouts_match_header(pp):
    assert ctypes.sizeof(pp._lib.PPDetection) == 104
    assert ctypes.sizeof(pp._lib.PPConfig) == 72 + 4 * 4 + 15 * 4 + 2 * 4 + 2 * 4 + 3 * 4 + 2 * 4 + 2 * 4


def test_no_cpu_fallback(pp, hip_lib):
    """Without a GPU the product must fail loudly, not compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no HIP device|PP_ERR_HIP"):
        pp.Engine(pp.config.tiny_config())
    with pytest.raises(RuntimeError):
        pp.points_to_voxel(np.zeros((4, 3), np.float32), [0.08, 0.08, 4.0], [0, -2.56, -3, 6.4, 2.56, 3], 50, True, 100)


def test_product_does_not_import_oracle():
    """The product must never import, load or execute anything under oracle/."""
    pkg = os.path.join(ROOT, "3d-object-detection-for-autonomous-navigation_amd")
    pat = re.compile(r"^\s*(import|from)\s+oracle\b|oracle[./]|libpp_oracle|import_module\([\"']oracle", re.M)
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dp, f)).read()
                assert not pat.search(txt), os.path.join(dp, f)


class _FakeGroup(dict):
    """dict with an `.attrs` dict: the slice of the h5py interface weights.from_keras_h5 uses."""
    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.attrs = {}


def _fake_keras_checkpoint(pp, d, w):
    """The variable list Keras 2.3 / TF 2.2 writes for the reference's model tree (tests/keras_tree.py) as a dict tree."""
    from keras_tree import LAYER_NAMES, keras_variables
    root = _FakeGroup()
    root.attrs["layer_names"] = [n.encode() for n in LAYER_NAMES]
    for layer, variables in keras_variables(d).items():
        g = _FakeGroup()
        g.attrs["weight_names"] = [k.encode() for k, _ in variables]
        for k, ours in variables:
            g[k] = w[ours] if ours is not None else np.ones(7, np.float32)
        root[layer] = g
    return root


def test_keras_checkpoint_name_mapping(pp):
    """f4: the .h5 importer's name / shape mapping on a synthetic checkpoint tree (h5py is absent here; the real
    file is read with the same code through tools/h5_to_npz.py)."""
    d = pp.config.Derived(pp.config.pedestrian_d435i_config())
    w = pp.weights.init_weights(d, seed=13)
    got = pp.weights.from_keras_h5(_fake_keras_checkpoint(pp, d, w), d)
    assert set(got) == set(w) and all(np.array_equal(got[k], w[k]) for k in w)
    broken = _fake_keras_checkpoint(pp, d, w)
    names = broken["rpn"].attrs["weight_names"]
    broken["rpn"].attrs["weight_names"] = [n for n in names if b"deconv2" not in n]
    with pytest.raises(ValueError, match="lacks"):
        pp.weights.from_keras_h5(broken, d)
    swapped = _fake_keras_checkpoint(pp, d, w)
    k0 = "voxel_net/rpn/block2/separable_conv2d_4/pointwise_kernel:0"
    swapped["rpn"][k0] = swapped["rpn"][k0][..., :64]
    with pytest.raises(ValueError, match="shape"):
        pp.weights.from_keras_h5(swapped, d)


@pytest.mark.parametrize("order", ["lib_first", "torch_first"])
def test_one_hip_runtime_per_process(order):
    """Whatever the import order of pp_amd's library and torch, the process maps ONE libamdhip64 (two copies: the
    runtime that initialises second reports hipErrorNoDevice on the GPU box -- _lib._one_hip_runtime)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    first, second = ("import pp_amd; pp_amd._lib.lib()", "import torch") if order == "lib_first" else \
                    ("import torch", "import pp_amd; pp_amd._lib.lib()")
    code = (f"import sys; sys.path.insert(0, {root!r})\n{first}\n{second}\n"
            "libs = sorted({l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l})\n"
            "print('N', len(libs), libs)\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-800:]
    line = [l for l in r.stdout.splitlines() if l.startswith("N ")][0]
    assert line.split()[1] == "1", line


def test_bench_accounts_per_launch_when_a_layer_is_walked_in_sub_ranges(pp):
    """bench.py's roofline arithmetic: a layer the engine walks in four frame sub-ranges (pp_set_cache_budget) is four
    launches per step -- algorithmic bytes / flops per LAUNCH are the layer's divided by four, the layer's time per step
    is the sum of its launches, and a whole-batch layer of the same symbol still counts as one."""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    d = pp.config.Derived(pp.config.kitti_shaped_config(32, num_class=2))
    steps = 3
    samples = {"k_sep_u<64,1,3,1,0>:block1.1": [0.055] * (4 * steps),      # four sub-range launches per step
               "k_sep_u<64,1,3,1,0>:block1.2": [0.220] * steps,            # one whole-batch launch per step
               "k_postprocess": [0.06] * steps}
    kernel_ms, launches, per_layer, dropped = bench.summarise(samples, steps)
    assert dropped == 0 and launches["k_sep_u<64,1,3,1,0>"] == 5
    assert per_layer["block1.1"][2] == 4 and per_layer["block1.2"][2] == 1
    assert abs(kernel_ms["k_sep_u<64,1,3,1,0>"] - 0.44) < 1e-9
    roofs = bench.kernel_roofs(d, 32, 20000, 5800.0, kernel_ms, launches, per_layer, True)
    lb = bench.layer_bytes(d, 32, True, 5800.0)
    r = roofs["k_sep_u<64,1,3,1,0>"]
    assert abs(r["algorithmic_bytes_per_launch"] - (lb["block1.1"] + lb["block1.2"]) / 5) < 1.0
    assert abs(r["avg_launch_ms"] - 0.44 / 5) < 1e-9
