"""Flat parameter storage.

All float32 parameters and buffers of a module are re-homed into one contiguous
HBM allocation (and the int64 BatchNorm counters into another), in state-dict
order; the module's tensors become views.  This is what makes the teacher EMA
(ltr/actor/transt_sparse_actor.py:11-33) a single streaming kernel instead of
several hundred tiny launches plus a load_state_dict, and what lets gradients
be all-reduced in a few large RCCL buckets instead of per-tensor messages.
"""
import torch


class ParamArena:
    def __init__(self, module, with_grads=False):
        self.module = module
        entries = list(module.state_dict(keep_vars=True).items())
        self.f32 = [(k, t) for k, t in entries if t.dtype == torch.float32]
        self.i64 = [(k, t) for k, t in entries if t.dtype == torch.int64]
        other = [k for k, t in entries if t.dtype not in (torch.float32, torch.int64)]
        if other:
            raise RuntimeError("ParamArena: unsupported dtypes for %s" % other)
        dev = self.f32[0][1].device
        # 64-float (256-byte) aligned slots so every view is 16-byte aligned for the kernels
        offs, off = [], 0
        for _, t in self.f32:
            offs.append(off)
            off += (t.numel() + 63) // 64 * 64
        self.flat = torch.zeros(off, dtype=torch.float32, device=dev)
        self.offsets = offs
        for (k, t), o in zip(self.f32, offs):
            view = self.flat[o:o + t.numel()].view(t.shape)
            view.copy_(t.data)
            t.data = view
        self.flat_i64 = torch.zeros(max(len(self.i64), 1), dtype=torch.int64, device=dev)
        for j, (k, t) in enumerate(self.i64):
            self.flat_i64[j] = t.data
            t.data = self.flat_i64[j]
        self.grad_flat = None
        self._ready_listeners = []
        self._owed = {}
        if with_grads:
            self.attach_grads()

    def attach_grads(self):
        """Give every trainable parameter a .grad that is a view into one flat buffer, and mark it as a
        gradient sink: kernels may add into .grad directly (see ops._sink) and report completion here."""
        self.grad_flat = torch.zeros_like(self.flat)
        for (k, t), o in zip(self.f32, self.offsets):
            if isinstance(t, torch.nn.Parameter) and t.requires_grad:
                t.grad = self.grad_flat[o:o + t.numel()].view(t.shape)
                t._spot_grad_sink = self
        return self.grad_flat

    def add_ready_listener(self, fn):
        self._ready_listeners.append(fn)

    def expect(self, p):
        """Forward of an op that will add p's gradient into p.grad itself: one more contribution is owed."""
        self._owed[id(p)] = self._owed.get(id(p), 0) + 1

    def grad_ready(self, p):
        """A kernel finished enqueueing one contribution to p.grad (the autograd hook does not fire for it).
        Listeners hear about p once, when every op that used p in the forward pass has delivered (a parameter
        can serve several ops: both backbone passes share their BatchNorm layers, input_proj runs twice)."""
        left = self._owed.get(id(p), 1) - 1
        self._owed[id(p)] = left
        if left > 0:
            return
        for fn in self._ready_listeners:
            fn(p)

    def zero_grad(self):
        """Clear the flat gradient buffer.  Call it BEFORE the forward pass of a step: it also forgets which
        kernels still owe a gradient (see expect / grad_ready)."""
        if self.grad_flat is not None:
            self.grad_flat.zero_()
        self._owed.clear()
        from . import ops
        ops._DROPPED.clear()        # masked gradients of a backward pass that never asked for them

    def signature(self):
        return [(k, tuple(t.shape)) for k, t in self.f32], [k for k, _ in self.i64]
