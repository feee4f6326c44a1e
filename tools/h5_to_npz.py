"""Converts a Keras checkpoint of the reference (`model_weights_<epoch>.h5`, written by net.save_weights at
train.py:407,436 and read back by net.load_weights at train.py:731-734) into the .npz this package loads
(`VoxelNet.load_weights(path)` / `weights.load_npz`).

    python tools/h5_to_npz.py configs/train.yaml out/model_345/out_dir_checkpoints/model_weights_48.h5 weights_48.npz

h5py is used when installed; otherwise the package's own reader (h5lite.py).  The name mapping and its checks are
`weights.map_keras_weight_names` / `weights.from_keras_h5` (tests/test_h5lite.py reads a checkpoint written by the real
HDF5 library in Keras's layout; `VoxelNet.load_weights(path)` takes the .h5 directly as well)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pp_amd  # noqa: E402


def main(argv):
    if len(argv) != 4:
        print(__doc__)
        return 2
    d = pp_amd.config.Derived(pp_amd.config.load_yaml(argv[1]))
    w = pp_amd.weights.load_keras_h5(argv[2], d)
    pp_amd.weights.save_npz(argv[3], w)
    print(f"{len(w)} tensors, {sum(v.size for v in w.values())} parameters -> {argv[3]}")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
