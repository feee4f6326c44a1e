"""Training loop body of train.py:228-304 on the HIP engine.

    trainer = Trainer(config, weights, max_batch=2)            # one per GPU / rank
    out = trainer.step(frames, labels, reg_targets, dist)      # forward + loss + backward, all-reduce, AdamW
    trainer.weights()                                          # Keras-layout dict (Engine.load_weights, save_npz)

What runs where: the frames are uploaded and voxelised by the engine, `pp_train_step` (csrc/train.hip) runs the
training-mode forward pass, the loss and the backward pass and leaves the gradients of all trainable tensors in one
flat float32 buffer; that buffer is averaged over the ranks with ONE all-reduce (`torch.distributed`, backend
"nccl" = RCCL over xGMI; BatchNorm statistics stay per replica, as in the single-GPU reference) and consumed by the
AdamW kernel (csrc/optim.hip).  torch owns the flat device buffers and the communicator; no torch operator touches
the numbers.  Labels / regression targets come from `target_assigner` (the reference's training dataloader).
"""
import numpy as np

from . import optim
from .engine import Engine


class TrainBatch:
    """Page-locked staging of one training batch (Trainer.stage)."""

    def close(self):
        for k in ("points", "_lab", "_reg"):
            o = getattr(self, k, None)
            if o is not None:
                o.close()
                setattr(self, k, None)
        self.labels = self.reg_targets = None


class Trainer:
    def __init__(self, config, weights, max_batch=None, max_points_per_frame=32768, device=0, learning_rate=None,
                 weight_decay=None):
        import torch
        self.torch = torch
        self.engine = Engine(config, max_batch=max_batch, max_points_per_frame=max_points_per_frame, device=device)
        self._prefetched = None      # the TrainBatch whose points are already on their way (forward_backward(prefetch=))
        self._ext_stream = None      # torch.cuda.ExternalStream over the engine's stream (_engine_stream)
        self.time_allreduce = False  # True: every step's gradient all-reduce is bracketed by an event pair (allreduce_ms)
        self._allreduce_events = []
        d = self.engine.d
        self.layout, n_params, n_state = self.engine.train_layout()
        self.device = torch.device("cuda", device)
        self.params = torch.zeros(n_params, dtype=torch.float32, device=self.device)
        self.grads = torch.zeros_like(self.params)
        self.state = torch.zeros(n_state, dtype=torch.float32, device=self.device)
        self.set_weights(weights)
        # train_config absent: the shipped YAML's values (configs/train.yaml: learning rate 2e-4, weight decay 1e-4).
        # train_config present but without the keys train.py:224-239 reads: an error, not a silent default.
        tc = config.get("train_config") if isinstance(config, dict) else None
        if learning_rate is None:
            if tc:
                try:
                    learning_rate = optim.ExponentialDecay.from_config(tc, max_batch or d.batch_size)
                except (KeyError, TypeError) as ex:
                    raise ValueError(f"train_config: learning-rate schedule keys missing or malformed ({ex!r})") from ex
            else:
                learning_rate = 2e-4
        if weight_decay is None:
            if tc:
                try:
                    weight_decay = float(tc["optimizer"]["adam_optimizer"]["weight_decay"])
                except (KeyError, TypeError) as ex:
                    raise ValueError(f"train_config: optimizer.adam_optimizer.weight_decay missing or malformed ({ex!r})") from ex
            else:
                weight_decay = 1e-4
        self.optimizer = optim.AdamW(self.params, learning_rate, weight_decay)

    # ---- Keras-layout dict <-> flat buffers ----
    def set_weights(self, w):
        p = np.zeros(self.params.numel(), np.float32)
        s = np.zeros(self.state.numel(), np.float32)
        for name, off, size, is_state in self.layout:
            a = np.ascontiguousarray(w[name], dtype=np.float32).reshape(-1)
            if a.size != size:
                raise ValueError(f"weight {name!r}: {a.size} values, the layout expects {size}")
            (s if is_state else p)[off:off + size] = a
        self.params.copy_(self.torch.from_numpy(p))
        self.state.copy_(self.torch.from_numpy(s))

    def _unflatten(self, flat_params, flat_state, like):
        out = {}
        for name, off, size, is_state in self.layout:
            src = flat_state if is_state else flat_params
            out[name] = src[off:off + size].reshape(like[name].shape).copy()
        return out

    def weights(self):
        from . import weights as _w
        shapes = _w.expected_shapes(self.engine.d)
        like = {k: np.empty(v, np.float32) for k, v in shapes.items()}
        return self._unflatten(self.params.cpu().numpy(), self.state.cpu().numpy(), like)

    def gradients(self):
        """The last step's gradients as a Keras-layout dict (trainable tensors only)."""
        from . import weights as _w
        shapes = _w.expected_shapes(self.engine.d)
        g = self.grads.cpu().numpy()
        return {name: g[off:off + size].reshape(shapes[name]).copy() for name, off, size, st in self.layout if not st}

    def decisions(self):
        """Parity tap (pp_train_fetch_decisions): what the last step decided at its non-differentiable points.
        {"pfn": int32 [batch, max_voxels, C] winning row of every pillar slot (-1 a padded row, -2 no gradient),
         "<layer>/bn": bool mask of the layer's ReLU in NCHW order (as a torch graph of the network holds the tensor)}
        for every separable layer and transposed convolution, forward order."""
        import ctypes
        from . import weights as _w
        eng = self.engine
        d = eng.d
        L = eng._lib

        def fetch(layer, itemsize):
            n = ctypes.c_int64(0)
            eng._check(L.pp_train_fetch_decisions(eng._h, layer, None, 0, ctypes.byref(n)), "pp_train_fetch_decisions")
            buf = np.empty(n.value * itemsize, np.uint8)
            eng._check(L.pp_train_fetch_decisions(eng._h, layer, buf.ctypes.data_as(ctypes.c_void_p), buf.size,
                                                  ctypes.byref(n)), "pp_train_fetch_decisions")
            return buf
        out = {}
        arg = fetch(-1, 4).view(np.int32)
        out["pfn"] = arg.reshape(-1, d.max_voxels, d.pfn_filters)
        B = out["pfn"].shape[0]
        k = 0
        for kind, name, s in _w.layer_table(d):
            if kind == "head":
                continue
            m = fetch(k, 1).astype(bool)
            k += 1
            if kind == "sep":
                out[name + "/bn"] = m.reshape(B, -1, s["cout"])             # [b][pixels][c]; the caller knows H x W
            else:
                out[name + "/bn"] = m.reshape(B, -1, s["k"], s["k"], s["cout"])   # [b][input pixels][ti][tj][c]
        return out

    # ---- one optimizer step ----
    def stage(self, frames, labels, reg_targets):
        """A training batch in page-locked host memory: the points as an engine Staging, labels / regression targets
        as pinned arrays (`.labels`, `.reg_targets`: refill them in place for the next batch of the same shape).  A
        staged batch goes to the GPU as three DMA transfers; ordinary numpy arrays take the runtime's pageable path
        (an extra host copy of ~0.5 MB per frame at the shipped configuration)."""
        d = self.engine.d
        B = len(frames)
        st = TrainBatch()
        st.points = self.engine.staging(frames)
        st._lab = self.engine.pinned((B, d.num_anchors), np.int32)
        st._reg = self.engine.pinned((B, d.num_anchors, 7), np.float32)
        st.labels, st.reg_targets = st._lab.array, st._reg.array
        st.labels[...] = np.asarray(labels, dtype=np.int32).reshape(B, d.num_anchors)
        st.reg_targets[...] = np.asarray(reg_targets, dtype=np.float32).reshape(B, d.num_anchors, 7)
        return st

    def _launch(self, frames, labels, reg_targets, prefetch):
        """Enqueue the step (and the upload of the next batch beside it); the caller waits with engine.train_step_wait()."""
        if isinstance(frames, TrainBatch):
            tb = frames
            if self._prefetched is not tb:
                self.engine.upload_async(tb.points)
            self._prefetched = None
            self.engine.train_step_async(self.params.data_ptr(), self.grads.data_ptr(), self.state.data_ptr(), tb.labels,
                                         tb.reg_targets)
            if isinstance(prefetch, TrainBatch):
                self.engine.upload_async(prefetch.points)
                self._prefetched = prefetch
            return
        self._prefetched = None
        self.engine.upload(frames)
        self.engine.train_step_async(self.params.data_ptr(), self.grads.data_ptr(), self.state.data_ptr(), labels, reg_targets)

    def forward_backward(self, frames, labels=None, reg_targets=None, prefetch=None):
        """frames: a list of clouds with labels / reg_targets, or one TrainBatch from stage().
        prefetch: the TrainBatch of the NEXT step -- its points go to the GPU (the handle's other input buffer, the copy
        stream) while this step's kernels run, the loader's hand-over of train.py:228-304; pass that same batch as
        `frames` of the next call."""
        self._launch(frames, labels, reg_targets, prefetch)
        return self.engine.train_step_wait()

    def _engine_stream(self):
        """torch's view of the engine's own stream: the all-reduce and the AdamW kernel are enqueued THERE, behind the
        backward pass, so nothing needs a host round trip between trainStep's halves (and nothing can overtake)."""
        if self._ext_stream is None:
            self._ext_stream = self.torch.cuda.ExternalStream(self.engine.stream_ptr(), device=self.device)
        return self._ext_stream

    def _enqueue_update(self, dist):
        with self.torch.cuda.stream(self._engine_stream()):
            if self.time_allreduce:      # an event pair around the exchange alone, on the stream it runs on
                ev = (self.torch.cuda.Event(enable_timing=True), self.torch.cuda.Event(enable_timing=True))
                ev[0].record()
            optim.allreduce_gradients(self.grads, dist)        # one collective per step over the flat buffer
            if self.time_allreduce:
                ev[1].record()
                self._allreduce_events.append(ev)
            self.optimizer.apply_gradients(self.grads)

    def allreduce_ms(self):
        """Device time (ms) of each gradient exchange since `time_allreduce` was set (call after the steps were waited
        for); the list is emptied."""
        evs, self._allreduce_events = self._allreduce_events, []
        return [a.elapsed_time(b) for a, b in evs]

    def apply_gradients(self, dist=None):
        """optimizer.apply_gradients (train.py:301) after the data-parallel mean of the flat gradient buffer.  The
        all-reduce and the AdamW kernel run on the engine's stream (behind the step that produced the gradients); this
        returns when both are through, so a caller may read the parameters."""
        self._enqueue_update(dist)
        self._engine_stream().synchronize()

    def step(self, frames, labels=None, reg_targets=None, dist=None, prefetch=None):
        """One optimizer step: forward + loss + backward, gradient exchange, AdamW -- enqueued back to back on the engine's
        stream, ONE host wait at the end."""
        self._launch(frames, labels, reg_targets, prefetch)
        try:
            self._enqueue_update(dist)
        except BaseException:
            # the step is in flight: wait for it (its own error is secondary) so the engine is not left "pending",
            # and forget the prefetch -- the next call uploads its batch itself
            self._abandon_step()
            raise
        return self.engine.train_step_wait()

    def _abandon_step(self):
        self._prefetched = None
        try:
            self.engine.train_step_wait()
        except Exception:      # noqa: BLE001 -- the caller's exception is the one to report
            pass

    def close(self):
        self.engine.close()
