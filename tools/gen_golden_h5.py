"""Writes tests/golden/keras_ckpt_tiny.h5 and tests/golden/h5lite_cases.h5 with the REAL HDF5 library.

h5py is not installed for the image's /usr/bin/python3, but the image carries an Anaconda tree whose own interpreter
has it (/opt/conda/bin/python3.9: h5py 3.3.0 on libhdf5 1.10.6).  This script runs in two stages:

    python tools/gen_golden_h5.py                 # stage 1, /usr/bin/python3: the weight tensors of the tiny
                                                  # configuration (weights.init_weights, seed 29) and the Keras variable
                                                  # tree (tests/keras_tree.py) -> a temporary .npz + .json; then it
                                                  # starts stage 2 in the other interpreter:
    /opt/conda/bin/python3.9 tools/gen_golden_h5.py --write <tmp.npz> <tmp.json> <out dir>

Stage 2 writes the checkpoint the way Keras's `save_weights` does (tensorflow/python/keras/saving/hdf5_format.py,
`save_weights_to_hdf5_group`, TF 2.2: root attributes `layer_names` / `backend` / `keras_version`, one group per
top-level layer with the attribute `weight_names`, one dataset per variable named by the variable -- the slashes in the
name make nested groups) and a second file of format cases for tests/test_h5lite.py.  The fixtures are data; nothing of
the reference is copied (TensorFlow itself is not in the image: the variable names are the ones its name scopes produce
for the reference's model tree, train.py:62-113).

    python tools/gen_golden_h5.py --full out.h5   # the same for the shipped (pedestrian) configuration, not committed
"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CONDA_PY = "/opt/conda/bin/python3.9"
SEED = 29


def stage1(argv):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import pp_amd as pp
    from keras_tree import LAYER_NAMES, keras_variables
    full = len(argv) >= 2 and argv[0] == "--full"
    cfg = pp.config.pedestrian_d435i_config() if full else pp.config.tiny_config()
    d = pp.config.Derived(cfg)
    w = pp.weights.init_weights(d, seed=SEED)
    plan = {"layer_names": LAYER_NAMES, "layers": {}, "full": argv[1] if full else None}
    arrays = {}
    for layer, variables in keras_variables(d).items():
        plan["layers"][layer] = []
        for k, ours in variables:
            key = f"a{len(arrays)}"
            arrays[key] = w[ours] if ours is not None else np.ones(7, np.float32)
            plan["layers"][layer].append([k, key])
    with tempfile.TemporaryDirectory() as td:
        npz, js = os.path.join(td, "w.npz"), os.path.join(td, "plan.json")
        np.savez(npz, **arrays)
        json.dump(plan, open(js, "w"))
        out = os.path.join(ROOT, "tests", "golden")
        subprocess.check_call([CONDA_PY, os.path.abspath(__file__), "--write", npz, js, out])


def write_keras(h5py, path, plan, arrays):
    with h5py.File(path, "w") as f:
        # TF 2.2 pins h5py < 3: a list of bytes becomes a numpy 'S' array and is stored as FIXED-LENGTH strings (h5py >= 3
        # would store the same list as variable-length strings: h5lite_cases.h5 covers that form)
        def S(names):
            return np.array(names, dtype="S") if names else np.zeros((0,), np.float64)   # np.asarray([]) is float64
        f.attrs.create("layer_names", S([n.encode("utf8") for n in plan["layer_names"]]))
        f.attrs.create("backend", np.bytes_(b"tensorflow"))
        f.attrs.create("keras_version", np.bytes_(b"2.3.0-tf"))
        for layer in plan["layer_names"]:
            g = f.create_group(layer)
            g.attrs.create("weight_names", S([k.encode("utf8") for k, _ in plan["layers"][layer]]))
            for k, key in plan["layers"][layer]:
                val = arrays[key]
                ds = g.create_dataset(k, val.shape, dtype=val.dtype)
                if not val.shape:
                    ds[()] = val
                else:
                    ds[:] = val


def write_cases(h5py, path):
    """One file with the format features h5lite claims, one per object (expected values are recomputed by the test)."""
    rng = np.random.default_rng(5)
    with h5py.File(path, "w", userblock_size=512) as f:
        f.attrs.create("title", np.bytes_(b"h5lite cases"))              # fixed-length string, scalar
        f.attrs.create("names_fixed", np.array([b"voxel_net/rpn/conv_box/kernel:0", b"b:0"], dtype="S"))
        f.attrs["vlen"] = "variable length ä"                        # variable-length UTF-8 string (global heap)
        f.attrs["vlen_list"] = np.array(["a", "bc", ""], dtype=h5py.string_dtype())
        f.attrs["empty"] = np.zeros((0,), np.float64)
        f.attrs["i64"] = np.int64(-5)
        f.attrs["f32_vec"] = np.arange(5, dtype=np.float32) / 3
        f.create_dataset("contig_f32", data=rng.standard_normal((7, 5)).astype(np.float32))
        f.create_dataset("be_f64", data=rng.standard_normal((3, 4)).astype(">f8"))
        f.create_dataset("be_i16", data=np.arange(-6, 6, dtype=">i2").reshape(3, 4))
        f.create_dataset("u8", data=np.arange(200, dtype=np.uint8))
        f.create_dataset("scalar", data=np.float32(2.5))
        f.create_dataset("f16", data=(np.arange(9) / 7).astype(np.float16))
        f.create_dataset("chunk_gzip", data=rng.integers(0, 50, (37, 23)).astype(np.int32), chunks=(8, 10),
                         compression="gzip", compression_opts=4, shuffle=True)
        f.create_dataset("chunk_plain", data=rng.standard_normal((20, 6)), chunks=(7, 4))
        f.create_dataset("chunk_fletcher", data=rng.standard_normal((16,)).astype(np.float32), chunks=(5,), fletcher32=True)
        f.create_dataset("never_written", shape=(4, 3), dtype=np.float32)
        f.create_dataset("strings", data=np.array([b"ab", b"cde", b""], dtype="S5"))
        f.create_dataset("vlen_strings", data=np.array(["x", "yz"], dtype=object), dtype=h5py.string_dtype())
        # compact layout through the low-level interface
        space = h5py.h5s.create_simple((6,))
        dcpl = h5py.h5p.create(h5py.h5p.DATASET_CREATE)
        dcpl.set_layout(h5py.h5d.COMPACT)
        dsid = h5py.h5d.create(f.id, b"compact_i32", h5py.h5t.STD_I32LE, space, dcpl)
        dsid.write(h5py.h5s.ALL, h5py.h5s.ALL, np.arange(6, dtype=np.int32) * 3)
        # a group with enough members for several symbol-table nodes and a two-level B-tree
        many = f.create_group("many")
        for i in range(300):
            many.create_dataset(f"d{i:03d}", data=np.int32(i))
        f.create_group("a/b/c").create_dataset("leaf", data=np.arange(4, dtype=np.int64))
        g = f["a/b"]
        for i in range(12):                                              # enough attributes for a continuation block
            g.attrs[f"attr{i}"] = np.full((3,), i, np.int32)
        f.create_dataset("compound", data=np.zeros(2, dtype=[("a", "i4"), ("b", "f4")]))   # h5lite must refuse it
    base = os.path.splitext(path)[0]
    with h5py.File(base + "_latest.h5", "w", libver="latest") as f:      # superblock 3, version-2 object headers
        f.attrs["note"] = b"libver latest"
        f.create_dataset("x", data=np.arange(10, dtype=np.float32))
        f.create_group("g").create_dataset("y", data=np.arange(6, dtype=np.int16).reshape(2, 3))
    with h5py.File(base + "_dense.h5", "w", libver="latest") as f:       # >8 links: dense storage, must be refused
        for i in range(20):
            f.create_dataset(f"d{i}", data=np.int32(i))


def stage2(argv):
    import h5py
    npz, js, out = argv
    plan = json.load(open(js))
    arrays = dict(np.load(npz))
    if plan.get("full"):
        write_keras(h5py, plan["full"], plan, arrays)
        print("wrote", plan["full"], os.path.getsize(plan["full"]), "bytes")
        return
    os.makedirs(out, exist_ok=True)
    p = os.path.join(out, "keras_ckpt_tiny.h5")
    write_keras(h5py, p, plan, arrays)
    print("wrote", p, os.path.getsize(p), "bytes; h5py", h5py.__version__, "hdf5", h5py.version.hdf5_version)
    p = os.path.join(out, "h5lite_cases.h5")
    write_cases(h5py, p)
    for q in (p, p[:-3] + "_latest.h5", p[:-3] + "_dense.h5"):
        print("wrote", q, os.path.getsize(q), "bytes")


if __name__ == "__main__":
    if len(sys.argv) >= 2 and sys.argv[1] == "--write":
        stage2(sys.argv[2:])
    else:
        stage1(sys.argv[1:])
