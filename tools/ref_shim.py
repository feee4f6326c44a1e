"""Import the reference's pure-numpy hot-path functions in THIS container only.

TEST INFRASTRUCTURE.  Nothing here ships to the GPU box's run time: the
reference tree (/root/reference) does not exist there.  This module is used
solely by tools/gen_golden.py to produce the committed fixtures under
tests/golden/ (inputs + expected outputs, data only).

The reference modules import tensorflow / numba / rospy / cv2 / ... at module
top (load_data.py:1-35, libraries/eval_helper_functions.py:1-28).  None of
those are installed and none are needed by the numpy functions we call, so a
meta-path finder fabricates empty stand-in modules for them; `numba.jit` and
friends become identity decorators so the decorated functions run as plain
Python, and `np.meshgrid` is wrapped to return a list (numpy>=2 returns a
tuple, the reference mutates the result, load_data.py:1630-1633).
Recipe recorded in SURVEY.md section 8c.
"""
import importlib
import importlib.abc
import importlib.machinery
import sys
import types

import numpy as np

REFERENCE_ROOT = "/root/reference"

_STUB_ROOTS = {
    "tensorflow", "tensorflow_addons", "tensorboard", "cv2", "ros_numpy", "rospy",
    "sensor_msgs", "jsk_recognition_msgs", "std_msgs", "geometry_msgs",
    "visualization_msgs", "fire", "wandb", "skimage", "matplotlib", "numba",
    "h5py", "pyqtgraph", "shapely", "tf", "tf2_ros", "pcl", "open3d",
}


# reference-internal modules that cannot load here and are not on the path we pin: the prebuilt CUDA
# extension nms.so (CPython 3.6 / libcudart 10.1) and the helper that would try to compile it
_STUB_EXACT = {
    "second.core.non_max_suppression.nms",
    "second.utils.buildtools.pybind11_build",
}

_NP_TYPES = {"float32": np.float32, "float64": np.float64, "int32": np.int32, "int64": np.int64,
             "uint64": np.uint64, "boolean": np.bool_}


def _local_array(shape, dtype=np.float32):
    # numba.cuda.local.array / shared.array when the "device function" runs as plain Python
    return np.zeros(shape, dtype=dtype)


def _identity_decorator(*args, **kwargs):
    # @jit / @jit(nopython=True) / @cuda.jit('sig', device=True)
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return args[0]
    return lambda fn: fn


class _Anything(types.ModuleType):
    """A module whose every attribute is another permissive stand-in."""

    def __getattr__(self, name):
        if name == "__version__":
            return "0.0-stub"
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        full = self.__name__ + "." + name
        if name in ("jit", "njit", "autojit", "vectorize", "guvectorize"):
            return _identity_decorator
        if self.__name__.split(".")[0] == "numba":
            if name in _NP_TYPES:
                return _NP_TYPES[name]
            if name == "prange":
                return range
            if name == "array" and self.__name__.split(".")[-1] in ("local", "shared"):
                return _local_array
        if name in ("Model", "Layer", "Loss"):
            return type(name, (object,), {})
        if name == "function":
            return _identity_decorator
        mod = sys.modules.get(full)
        if mod is None:
            mod = _Anything(full)
            mod.__path__ = []
            sys.modules[full] = mod
        return mod

    def __call__(self, *args, **kwargs):
        return _Anything(self.__name__ + "()")

    def __iter__(self):
        return iter(())


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path, target=None):
        if fullname.split(".")[0] in _STUB_ROOTS or fullname in _STUB_EXACT:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        mod = _Anything(spec.name)
        mod.__path__ = []
        return mod

    def exec_module(self, module):
        pass


_installed = False


def install():
    global _installed
    if _installed:
        return
    sys.dont_write_bytecode = True  # keep /root/reference untouched
    sys.meta_path.insert(0, _StubFinder())
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    _orig_meshgrid = np.meshgrid

    def _meshgrid_list(*a, **k):
        return list(_orig_meshgrid(*a, **k))

    np.meshgrid = _meshgrid_list
    _installed = True


def load_reference_eval():
    """Returns (second.utils.eval, second.core.non_max_suppression.nms_gpu) for the AP-evaluator fixtures."""
    install()
    nms_gpu = importlib.import_module("second.core.non_max_suppression.nms_gpu")
    ev = importlib.import_module("second.utils.eval")
    return ev, nms_gpu


def load_reference():
    """Returns (load_data, eval_helper_functions) reference modules."""
    install()
    load_data = importlib.import_module("load_data")
    ehf = importlib.import_module("libraries.eval_helper_functions")
    return load_data, ehf
