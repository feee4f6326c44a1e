"""Optimizer step + gradient exchange.  The update formula restates tfa.optimizers.AdamW 0.11.2
over tf.keras Adam 2.2.0 (neither installable here: parity unpinned); the checker below is that formula written
in float64 numpy."""
import math
import os
import subprocess
import sys

import numpy as np
import pytest

import pp_amd as pp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_exponential_decay_schedule():
    s = pp.optim.ExponentialDecay.from_config(
        {"optimizer": {"adam_optimizer": {"learning_rate": {"exponential_decay_learning_rate": {
            "initial_learning_rate": 0.002, "decay_steps": 7000, "decay_factor": 0.8, "staircase": False}},
            "weight_decay": 0.0001}}}, batch_size=2)
    assert s.decay_steps == 3500.0
    assert s(0) == 0.002
    assert abs(s(3500) - 0.0016) < 1e-12 and abs(s(1750) - 0.002 * 0.8 ** 0.5) < 1e-12
    st = pp.optim.ExponentialDecay(0.002, 3500, 0.8, staircase=True)
    assert st(3499) == 0.002 and abs(st(3500) - 0.0016) < 1e-12


def _adamw_numpy(w, g, m, v, lr_t, b1, b2, eps, wd):
    # the hyper-parameters are float32 values in the variable's dtype (as in TF): 1 - beta is the float32 difference
    f = np.float32
    lr_t, eps, wd = float(f(lr_t)), float(f(eps)), float(f(wd))
    omb1, omb2 = float(f(1) - f(b1)), float(f(1) - f(b2))
    b1, b2 = float(f(b1)), float(f(b2))
    w = w - wd * w
    m = b1 * m + omb1 * g
    v = b2 * v + omb2 * g * g
    return w - lr_t * m / (np.sqrt(v) + eps), m, v


@pytest.mark.gpu
def test_adamw_kernel_matches_formula():
    import torch
    rng = np.random.default_rng(4)
    n = 1_100_003                                    # the model's size, not a multiple of 4
    w0 = rng.normal(0, 0.1, n).astype(np.float32)
    dev = torch.device("cuda", 0)
    w = torch.tensor(w0, device=dev)
    sched = pp.optim.ExponentialDecay(0.002, 3500, 0.8)
    opt = pp.optim.AdamW(w, sched, weight_decay=0.0001)
    wr, mr, vr = w0.astype(np.float64), np.zeros(n), np.zeros(n)
    for step in range(3):
        g0 = rng.normal(0, 0.01, n).astype(np.float32)
        lr_t = sched(step) * math.sqrt(1 - 0.999 ** (step + 1)) / (1 - 0.9 ** (step + 1))
        assert abs(opt.lr_t() - lr_t) < 1e-15
        opt.apply_gradients(torch.tensor(g0, device=dev))
        wr, mr, vr = _adamw_numpy(wr, g0.astype(np.float64), mr, vr, lr_t, 0.9, 0.999, 1e-8, 0.0001)
    torch.cuda.synchronize()
    np.testing.assert_allclose(w.cpu().numpy(), wr, rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(opt.m.cpu().numpy(), mr, rtol=1e-5, atol=1e-9)
    np.testing.assert_allclose(opt.v.cpu().numpy(), vr, rtol=1e-5, atol=1e-12)
    assert opt.iterations == 3
    with pytest.raises(ValueError):
        opt.apply_gradients(torch.zeros(5, device=dev))


def test_adamw_refuses_cpu_tensors():
    import torch
    with pytest.raises(RuntimeError):
        pp.optim.AdamW(torch.zeros(8), 0.001, 0.0)


_WORKER = r"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch, torch.distributed as dist
import pp_amd as pp
dist.init_process_group(backend="gloo")
r, n = dist.get_rank(), dist.get_world_size()
g = torch.full((1000,), float(r + 1))
pp.optim.allreduce_gradients(g, dist)
assert torch.allclose(g, torch.full((1000,), (n + 1) / 2.0)), g[:3]
print("rank", r, "ok")
dist.destroy_process_group()
"""


def test_gradient_allreduce_two_ranks_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29533", str(script)], cwd=ROOT,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    assert r.stdout.count("ok") == 2
    import torch
    g = torch.ones(4)
    assert pp.optim.allreduce_gradients(g, None) is g
