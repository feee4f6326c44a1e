"""Optimizer step and gradient exchange of the training loop.

  ExponentialDecay     tf.keras.optimizers.schedules.ExponentialDecay as built at train.py:224-229
                       (decay_steps is divided by the batch size there)
  AdamW                tfa.optimizers.AdamW(learning_rate=schedule, weight_decay, epsilon=1e-8), train.py:231-236,
                       applied by optimizer.apply_gradients (train.py:301): one HIP launch per step over ONE flat
                       float32 buffer (parameters, gradients and both moments live in HBM; torch owns the memory)
  allreduce_gradients  the data-parallel exchange BASELINE.json configs[4] names: the flat gradient buffer is
                       summed over the ranks and divided by the world size -- one collective per step
                       (RCCL through torch.distributed's "nccl" backend on the GPUs, gloo in the CPU tests)
The arithmetic of the update is in csrc/optim.hip; this module only holds the step counter and the schedule.
"""
import ctypes
import math
import os


class ExponentialDecay:
    def __init__(self, initial_learning_rate, decay_steps, decay_rate, staircase=False):
        self.initial_learning_rate = float(initial_learning_rate)
        self.decay_steps = float(decay_steps)
        self.decay_rate = float(decay_rate)
        self.staircase = bool(staircase)

    def __call__(self, step):
        p = float(step) / self.decay_steps
        if self.staircase:
            p = math.floor(p)
        return self.initial_learning_rate * self.decay_rate ** p

    @classmethod
    def from_config(cls, train_config, batch_size):
        """train.py:224-229: the YAML's decay_steps counts samples, the schedule counts optimizer steps."""
        c = train_config["optimizer"]["adam_optimizer"]["learning_rate"]["exponential_decay_learning_rate"]
        return cls(c["initial_learning_rate"], c["decay_steps"] / batch_size, c["decay_factor"], c["staircase"])


class AdamW:
    """params / grads: flat contiguous float32 torch tensors on the same GPU (the caller keeps `grads` filled)."""

    def __init__(self, params, learning_rate, weight_decay, beta_1=0.9, beta_2=0.999, epsilon=1e-8):
        import torch
        if params.dtype != torch.float32 or not params.is_contiguous() or params.dim() != 1:
            raise ValueError("params must be a flat contiguous float32 tensor")
        if not params.is_cuda:
            raise RuntimeError("AdamW runs on the GPU only (HIP kernel k_adamw); there is no CPU path")
        self.params = params
        self.m = torch.zeros_like(params)
        self.v = torch.zeros_like(params)
        self.learning_rate = learning_rate
        self.weight_decay = float(weight_decay)
        self.beta_1, self.beta_2, self.epsilon = float(beta_1), float(beta_2), float(epsilon)
        self.iterations = 0

    def lr_t(self):
        lr = self.learning_rate(self.iterations) if callable(self.learning_rate) else float(self.learning_rate)
        t = self.iterations + 1
        return lr * math.sqrt(1.0 - self.beta_2 ** t) / (1.0 - self.beta_1 ** t)

    def apply_gradients(self, grads):
        import torch
        from . import _lib
        if grads.shape != self.params.shape or grads.dtype != torch.float32 or not grads.is_contiguous() or \
                grads.device != self.params.device:
            raise ValueError("grads must match params (flat float32, same device)")
        stream = torch.cuda.current_stream(self.params.device).cuda_stream
        st = _lib.lib().pp_adamw_step_device(self.params.device.index or 0, ctypes.c_void_p(stream),
                                             ctypes.c_void_p(self.params.data_ptr()), ctypes.c_void_p(grads.data_ptr()),
                                             ctypes.c_void_p(self.m.data_ptr()), ctypes.c_void_p(self.v.data_ptr()),
                                             self.params.numel(), self.lr_t(), self.beta_1, self.beta_2, self.epsilon,
                                             self.weight_decay)
        if st != 0:
            raise RuntimeError("pp_adamw_step_device failed: " + _lib.lib().pp_last_error(None).decode())
        self.iterations += 1


def allreduce_gradients(flat_grads, dist=None):
    """Mean of the flat gradient buffer over the ranks, in place; a single collective.  `dist` is
    torch.distributed (or None / uninitialised / world size 1: nothing to do)."""
    if dist is None or not dist.is_initialized():
        return flat_grads
    if dist.get_world_size() == 1 and os.environ.get("PP_FORCE_ALLREDUCE") != "1":
        return flat_grads       # (PP_FORCE_ALLREDUCE=1: the one-rank rehearsal still sends the buffer through the backend)
    dist.all_reduce(flat_grads, op=dist.ReduceOp.SUM)
    flat_grads /= dist.get_world_size()
    return flat_grads
