"""f4: the reference's checkpoints are Keras `save_weights` HDF5 files (train.py:407,436,731-734).
The package reads them without h5py (h5lite.py).  The fixtures under tests/golden/*.h5 were written by the REAL library
(h5py 3.3.0 on libhdf5 1.10.6, tools/gen_golden_h5.py); the expected values are recomputed here from the same seeds."""
import os

import numpy as np
import pytest

from keras_tree import LAYER_NAMES, keras_variables

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CKPT = os.path.join(GOLD, "keras_ckpt_tiny.h5")
CASES = os.path.join(GOLD, "h5lite_cases.h5")


def test_keras_checkpoint_file_written_by_the_real_library(pp):
    d = pp.config.Derived(pp.config.tiny_config())
    want = pp.weights.init_weights(d, seed=29)                      # tools/gen_golden_h5.py: SEED
    with pp.h5lite.File(CKPT) as f:
        assert [n.decode() for n in f.attrs["layer_names"]] == LAYER_NAMES
        assert f.attrs["backend"] == b"tensorflow" and f.attrs["keras_version"] == b"2.3.0-tf"
        assert sorted(f.keys()) == sorted(LAYER_NAMES)
        tree = keras_variables(d)
        for layer in LAYER_NAMES:
            names = [n.decode() for n in np.asarray(f[layer].attrs["weight_names"]).ravel().tolist()] \
                if np.asarray(f[layer].attrs["weight_names"]).size else []
            assert names == [k for k, _ in tree[layer]]
        # a variable's dataset sits under nested groups made by the slashes of its name
        k0 = "voxel_net/rpn/block2/separable_conv2d_2/pointwise_kernel:0"
        ds = f["rpn"][k0]
        assert ds.shape == (1, 1, 32, 32) and ds.dtype == np.float32
        assert np.array_equal(np.asarray(ds), want["rpn/block2/0/pointwise_kernel"])
        assert np.array_equal(np.asarray(f["/rpn/voxel_net/rpn/conv_box/bias:0"]), want["rpn/conv_box/bias"])
        got = pp.weights.from_keras_h5(f, d)
    assert set(got) == set(want) and all(np.array_equal(got[k], want[k]) and got[k].dtype == np.float32 for k in want)
    # the path-level entry points: by signature, not by extension
    assert pp.h5lite.is_hdf5(CKPT) and not pp.h5lite.is_hdf5(os.path.join(GOLD, "ref_anchors.npz"))
    got = pp.weights.load_any(CKPT, d)
    assert all(np.array_equal(got[k], want[k]) for k in want)
    # a checkpoint of another configuration is refused, as net.load_weights refuses a layout mismatch
    with pytest.raises(ValueError):
        pp.weights.load_keras_h5(CKPT, pp.config.Derived(pp.config.pedestrian_d435i_config()))


def test_name_lists_split_over_several_attributes(pp):
    """Keras splits a name list that does not fit one object-header attribute into `weight_names0`, `weight_names1`, ...
    (`save_attributes_to_hdf5_group`); the importer joins them as `load_attributes_from_hdf5_group` does."""
    d = pp.config.Derived(pp.config.tiny_config())
    w = pp.weights.init_weights(d, seed=3)

    class G(dict):
        def __init__(self):
            super().__init__()
            self.attrs = {}
    root = G()
    root.attrs["layer_names"] = np.array([n.encode() for n in LAYER_NAMES], dtype="S")
    for layer, variables in keras_variables(d).items():
        g = G()
        names = [k.encode() for k, _ in variables]
        if len(names) > 4:
            third = len(names) // 3
            for i, part in enumerate((names[:third], names[third:2 * third], names[2 * third:])):
                g.attrs[f"weight_names{i}"] = np.array(part, dtype="S")
        else:
            g.attrs["weight_names"] = np.array(names, dtype="S") if names else np.zeros((0,), np.float64)
        for k, ours in variables:
            g[k] = w[ours] if ours is not None else np.ones(7, np.float32)
        root[layer] = g
    got = pp.weights.from_keras_h5(root, d)
    assert all(np.array_equal(got[k], w[k]) for k in w)


def test_format_cases(pp):
    h5 = pp.h5lite
    rng = np.random.default_rng(5)                                  # tools/gen_golden_h5.py: write_cases draws in this order
    with h5.File(CASES) as f:                                       # (512-byte user block in front of the superblock)
        a = f.attrs
        assert a["title"] == b"h5lite cases"
        assert a["names_fixed"].tolist() == [b"voxel_net/rpn/conv_box/kernel:0", b"b:0"]
        assert a["vlen"] == "variable length ä" and a["vlen_list"].tolist() == ["a", "bc", ""]
        assert a["empty"].shape == (0,) and a["i64"] == -5 and a["i64"].dtype == np.int64
        assert np.array_equal(a["f32_vec"], np.arange(5, dtype=np.float32) / 3)
        assert np.array_equal(np.asarray(f["contig_f32"]), rng.standard_normal((7, 5)).astype(np.float32))
        be = np.asarray(f["be_f64"])
        assert be.dtype == np.float64 and np.array_equal(be, rng.standard_normal((3, 4)))
        assert np.array_equal(np.asarray(f["be_i16"]), np.arange(-6, 6, dtype=np.int16).reshape(3, 4))
        assert np.array_equal(np.asarray(f["u8"]), np.arange(200, dtype=np.uint8))
        assert np.asarray(f["scalar"]).shape == () and float(np.asarray(f["scalar"])) == 2.5
        assert np.array_equal(np.asarray(f["f16"]), (np.arange(9) / 7).astype(np.float16))
        assert np.array_equal(np.asarray(f["chunk_gzip"]), rng.integers(0, 50, (37, 23)).astype(np.int32))   # edge chunks
        assert np.array_equal(np.asarray(f["chunk_plain"]), rng.standard_normal((20, 6)))
        assert np.array_equal(np.asarray(f["chunk_fletcher"]), rng.standard_normal((16,)).astype(np.float32))
        assert np.array_equal(np.asarray(f["never_written"]), np.zeros((4, 3), np.float32))
        assert np.asarray(f["strings"]).tolist() == [b"ab", b"cde", b""]
        assert np.asarray(f["vlen_strings"]).tolist() == ["x", "yz"]
        assert np.array_equal(np.asarray(f["compact_i32"]), np.arange(6, dtype=np.int32) * 3)
        assert np.array_equal(f["chunk_gzip"][3:5, ::7], np.asarray(f["chunk_gzip"])[3:5, ::7])
        many = f["many"]                                            # 300 members: a B-tree with several leaf nodes
        assert len(many) == 300 and sorted(many.keys()) == [f"d{i:03d}" for i in range(300)]
        assert all(int(np.asarray(many[f"d{i:03d}"])) == i for i in (0, 1, 149, 150, 299))
        assert np.array_equal(np.asarray(f["a/b/c/leaf"]), np.arange(4)) and f["a"]["b"]["c"].name == "/a/b/c"
        ab = f["a/b"].attrs                                         # twelve attributes: a continuation block
        assert len(ab) == 12 and all(np.array_equal(ab[f"attr{i}"], np.full((3,), i, np.int32)) for i in range(12))
        assert "compound" in f and "nope" not in f
        with pytest.raises(KeyError):
            f["a/b/missing"]
        with pytest.raises(h5.Unsupported, match="compound"):       # listed, but never mis-read
            np.asarray(f["compound"])
        assert [p for p, _ in f["a"].visit_datasets()] == ["/a/b/c/leaf"]
    with h5.File(os.path.join(GOLD, "h5lite_cases_latest.h5")) as f:     # superblock 3, version-2 object headers, link messages
        assert f.attrs["note"] == "libver latest" and sorted(f.keys()) == ["g", "x"]
        assert np.array_equal(np.asarray(f["x"]), np.arange(10, dtype=np.float32))
        assert np.array_equal(np.asarray(f["g/y"]), np.arange(6, dtype=np.int16).reshape(2, 3))
    with h5.File(os.path.join(GOLD, "h5lite_cases_dense.h5")) as f:      # fractal-heap groups: refused by name
        with pytest.raises(h5.Unsupported, match="dense link storage"):
            f.keys()


def test_not_an_hdf5_file(pp, tmp_path):
    p = tmp_path / "x.h5"
    p.write_bytes(b"PK\x03\x04" + b"\0" * 600)
    with pytest.raises(pp.h5lite.FormatError):
        pp.h5lite.File(str(p))
    (tmp_path / "empty.h5").write_bytes(b"")
    with pytest.raises(pp.h5lite.FormatError):
        pp.h5lite.File(str(tmp_path / "empty.h5"))
    trunc = tmp_path / "trunc.h5"
    trunc.write_bytes(open(CKPT, "rb").read()[:4096])               # a cut-off download: an error, not garbage
    d = pp.config.Derived(pp.config.tiny_config())
    with pytest.raises((pp.h5lite.FormatError, ValueError, KeyError)):
        pp.weights.load_keras_h5(str(trunc), d)


def test_damaged_files_raise_and_nothing_else(pp, tmp_path):
    """Random truncations and byte flips of the fixtures: FormatError / Unsupported / KeyError, never another
    exception, a hang or a silently different array shape than the header promises."""
    h5 = pp.h5lite
    rng = np.random.default_rng(7)
    ok = (h5.Unsupported, h5.FormatError, KeyError)

    def walk(g, depth=0):
        dict(g.attrs)
        for k in g.keys():
            try:
                n = g[k]
            except ok:
                continue
            if isinstance(n, h5.Group):
                if depth < 6:
                    walk(n, depth + 1)
            else:
                try:
                    a = np.asarray(n)
                    assert n.shape is None or a.shape == tuple(n.shape)
                except ok:
                    pass
    for name, lo in (("h5lite_cases.h5", 512), ("keras_ckpt_tiny.h5", 0), ("h5lite_cases_latest.h5", 0)):
        src = open(os.path.join(GOLD, name), "rb").read()
        for it in range(60):
            b = bytearray(src)
            if it % 3 == 0:
                b = b[:rng.integers(100, len(b))]
            for _ in range(rng.integers(1, 20)):
                b[rng.integers(lo, min(len(b), lo + 20000))] = rng.integers(0, 256)
            p = tmp_path / "f.h5"
            p.write_bytes(bytes(b))
            try:
                with h5.File(str(p)) as f:
                    walk(f)
            except ok:
                pass
