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

Round 4: `numba.cuda` is no longer an empty stand-in but a small EMULATOR of the CUDA execution model (below:
_CudaEmu), so that the reference's kernels themselves -- `nms_kernel` (libraries/eval_helper_functions.py:567-598) and
`rotate_iou_kernel_eval` (second/core/non_max_suppression/nms_gpu.py:493-527), with their block / thread indexing,
shared-memory staging and `syncthreads` -- run unmodified through the reference's own host wrappers (`nms_gpu`,
`rotate_iou_gpu_eval`): one Python thread per CUDA thread of a block, a barrier for `syncthreads`, one array per
`cuda.shared.array` call site and block.  (The image's second interpreter has a real numba, but it does not import
against that interpreter's numpy: an ordinary SystemError.)  Arithmetic caveat, as for the device functions before:
plain Python keeps float32 + integer literal in float32 where numba types it float64, so fixtures made this way pin
indexing and decisions (inputs are drawn with margins), not the last bit of an IoU.
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
        if full == "numba.cuda":
            mod = sys.modules.get(full)
            if not isinstance(mod, _CudaEmu):
                mod = _CudaEmu()
                sys.modules[full] = mod
            return mod
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


class _Dim3:
    def __init__(self, x=0, y=0, z=0):
        self.x, self.y, self.z = x, y, z


class _DevArray:
    """cuda.to_device result: the kernels index it like an array; copy_to_host writes back."""

    def __init__(self, a):
        self.a = np.array(a, copy=True)

    def copy_to_host(self, ary=None, stream=None):
        if ary is None:
            return self.a.copy()
        ary[...] = self.a.reshape(ary.shape)
        return ary


class _Stream:
    def auto_synchronize(self):
        import contextlib
        return contextlib.nullcontext(self)

    def synchronize(self):
        pass


class _CudaKernel:
    def __init__(self, emu, fn):
        self.emu, self.fn = emu, fn

    def __getitem__(self, cfg):
        grid, block = cfg[0], cfg[1]
        grid = tuple(grid) if isinstance(grid, (tuple, list)) else (int(grid),)
        block = tuple(block) if isinstance(block, (tuple, list)) else (int(block),)
        grid = tuple(int(g) for g in grid) + (1,) * (3 - len(grid))
        block = tuple(int(b) for b in block) + (1,) * (3 - len(block))

        def launch(*args):
            import threading
            args = [a.a if isinstance(a, _DevArray) else a for a in args]
            nthreads = block[0] * block[1] * block[2]
            for bz in range(grid[2]):
                for by in range(grid[1]):
                    for bx in range(grid[0]):
                        state = {"barrier": threading.Barrier(nthreads), "shared": {}, "lock": threading.Lock()}
                        errors = []

                        def run(tx, ty, tz):
                            tls = self.emu._tls
                            tls.threadIdx, tls.blockIdx = _Dim3(tx, ty, tz), _Dim3(bx, by, bz)
                            tls.blockDim, tls.gridDim = _Dim3(*block), _Dim3(*grid)
                            tls.state, tls.nshared = state, 0
                            try:
                                self.fn(*args)
                            except BaseException as ex:  # noqa: BLE001
                                errors.append(ex)
                                state["barrier"].abort()
                        ts = [threading.Thread(target=run, args=(tx, ty, tz)) for tz in range(block[2])
                              for ty in range(block[1]) for tx in range(block[0])]
                        for t in ts:
                            t.start()
                        for t in ts:
                            t.join()
                        if errors:
                            raise errors[0]
        return launch


class _CudaEmu(types.ModuleType):
    """The slice of numba.cuda the reference's kernels use, executed with Python threads."""

    def __init__(self, name="numba.cuda"):
        super().__init__(name)
        import threading
        self.__path__ = []
        self._tls = threading.local()
        emu = self

        class _Local:
            @staticmethod
            def array(shape, dtype=np.float32):
                return np.zeros(shape, dtype=dtype)

        class _Shared:
            @staticmethod
            def array(shape, dtype=np.float32):
                tls = emu._tls
                key = tls.nshared                      # the n-th shared.array call of this thread: one array per block
                tls.nshared += 1
                with tls.state["lock"]:
                    if key not in tls.state["shared"]:
                        tls.state["shared"][key] = np.zeros(shape, dtype=dtype)
                    return tls.state["shared"][key]
        self.local, self.shared = _Local, _Shared

    threadIdx = property(lambda self: self._tls.threadIdx)
    blockIdx = property(lambda self: self._tls.blockIdx)
    blockDim = property(lambda self: self._tls.blockDim)
    gridDim = property(lambda self: self._tls.gridDim)

    def syncthreads(self):
        self._tls.state["barrier"].wait()

    def jit(self, *args, **kwargs):
        device = bool(kwargs.get("device", False))
        if len(args) == 1 and callable(args[0]) and not kwargs:
            return _CudaKernel(self, args[0])
        return (lambda fn: fn) if device else (lambda fn: _CudaKernel(self, fn))

    def to_device(self, ary, stream=None):
        return _DevArray(ary)

    def stream(self):
        return _Stream()

    def select_device(self, i):
        return None

    def synchronize(self):
        return None


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path, target=None):
        if fullname.split(".")[0] in _STUB_ROOTS or fullname in _STUB_EXACT:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        if spec.name == "numba.cuda":
            return _CudaEmu()
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
