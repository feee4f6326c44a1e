"""Converts a Keras checkpoint of the reference (`model_weights_<epoch>.h5`, written by net.save_weights at
train.py:407,436 and read back by net.load_weights at train.py:731-734) into the .npz this package loads
(`VoxelNet.load_weights(path)` / `weights.load_npz`).

    python tools/h5_to_npz.py configs/train.yaml out/model_345/out_dir_checkpoints/model_weights_48.h5 weights_48.npz

Needs h5py (not installed in the build image: run it wherever the checkpoint lives).  The name mapping and its
checks are `weights.map_keras_weight_names` / `weights.from_keras_h5` (unit-tested with a synthetic group tree)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pp_amd  # noqa: E402


def main(argv):
    if len(argv) != 4:
        print(__doc__)
        return 2
    try:
        import h5py
    except ImportError:
        print("h5py is required to read the checkpoint (pip install h5py)")
        return 1
    d = pp_amd.config.Derived(pp_amd.config.load_yaml(argv[1]))
    with h5py.File(argv[2], "r") as f:
        w = pp_amd.weights.from_keras_h5(f, d)
    pp_amd.weights.save_npz(argv[3], w)
    print(f"{len(w)} tensors, {sum(v.size for v in w.values())} parameters -> {argv[3]}")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
