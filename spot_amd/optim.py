"""AdamW over the flat parameter buffer: one kernel launch per step.

Drop-in for the ``torch.optim.AdamW`` the reference builds in
ltr/train_settings/spot/spot_transt_003.py:148-157 (two parameter groups: everything but the
backbone at lr, the backbone at lr * backbone_ratio; weight decay 1e-4): same constructor
signature for the arguments used there, same ``param_groups`` (learning-rate schedulers work
unchanged), same update rule.  All parameters must live in one ``ParamArena`` with gradients
attached (the trainer arranges that); the first and second moments are two more flat buffers.
torch's foreach AdamW needs ~2 ms for the 23 M parameters (a handful of multi-tensor launches
per group, each streaming the buffers again); this is one 650 MB stream (~0.15 ms).

Differences from torch.optim.AdamW that a caller can observe:
  * every parameter handed in must require a gradient (a frozen one is rejected); a parameter that receives no
    gradient in a step is updated with gradient zero (it still decays and its moments shrink), where torch skips
    parameters whose .grad is None;
  * one step counter for all parameters: a checkpoint whose per-parameter steps differ is rejected;
  * the update itself is deterministic, but the gradients it consumes are summed by split-reduction GEMMs and attention
    kernels with float atomics, so two runs of the same step agree to rounding, not bit for bit.
state_dict() is torch's layout (per-parameter step / exp_avg / exp_avg_sq, views of the flat buffers), so checkpoints
travel in both directions between this class and torch.optim.AdamW.
"""
import ctypes as C
import math

import torch

from . import lib
from .ops import _p, _st


class _Chunk(C.Structure):
    _fields_ = [("offset", C.c_long), ("count", C.c_int), ("group", C.c_int)]


class _Groups(C.Structure):
    _fields_ = [("lr", C.c_float * 4), ("weight_decay", C.c_float * 4), ("beta1", C.c_float), ("beta2", C.c_float),
                ("eps", C.c_float), ("bias_correction1", C.c_float), ("sqrt_bias_correction2", C.c_float)]


class FlatAdamW(torch.optim.Optimizer):
    CHUNK = 4096

    def __init__(self, params, arena, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        if len(self.param_groups) > 4:
            raise ValueError("FlatAdamW handles up to 4 parameter groups")
        if arena.grad_flat is None:
            arena.attach_grads()
        self.arena = arena
        offset_of = self._offset_of = {id(t): (o, t.numel()) for (k, t), o in zip(arena.f32, arena.offsets)}
        first = self.param_groups[0]
        chunks = []
        for gi, group in enumerate(self.param_groups):
            if tuple(group["betas"]) != tuple(first["betas"]) or group["eps"] != first["eps"]:
                raise ValueError("FlatAdamW: betas and eps must be the same in every group")
            for p in group["params"]:
                if id(p) not in offset_of:
                    raise ValueError("FlatAdamW: every parameter must live in the given ParamArena")
                if not p.requires_grad:
                    raise ValueError("FlatAdamW: a parameter with requires_grad=False was handed in; leave frozen "
                                     "parameters out of the groups (they would decay and drift)")
                off, n = offset_of[id(p)]
                for s in range(0, n, self.CHUNK):
                    chunks.append((off + s, min(self.CHUNK, n - s), gi))
        arr = (_Chunk * len(chunks))()
        for e, (o, c, gi) in zip(arr, chunks):
            e.offset, e.count, e.group = o, c, gi
        self._n_chunks = len(chunks)
        self._chunks = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(arena.flat.device)
        self.exp_avg = torch.zeros_like(arena.flat)
        self.exp_avg_sq = torch.zeros_like(arena.flat)
        self._step = 0

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        self._step += 1
        b1, b2 = self.param_groups[0]["betas"]
        hp = _Groups()
        for gi, group in enumerate(self.param_groups):
            hp.lr[gi], hp.weight_decay[gi] = group["lr"], group["weight_decay"]
        hp.beta1, hp.beta2, hp.eps = b1, b2, self.param_groups[0]["eps"]
        hp.bias_correction1 = 1.0 - b1 ** self._step
        hp.sqrt_bias_correction2 = math.sqrt(1.0 - b2 ** self._step)
        lib.call("spot_adamw_f32", _p(self.arena.flat), _p(self.arena.grad_flat), _p(self.exp_avg),
                 _p(self.exp_avg_sq), _p(self._chunks), self._n_chunks, C.byref(hp), _st())
        return loss

    def zero_grad(self, set_to_none=False):
        self.arena.zero_grad()

    def _moment_views(self):
        """(exp_avg, exp_avg_sq) views per parameter, in the order torch numbers them in a state dict."""
        out = []
        for group in self.param_groups:
            for p in group["params"]:
                off, n = self._offset_of[id(p)]
                out.append((self.exp_avg[off:off + n].view_as(p), self.exp_avg_sq[off:off + n].view_as(p)))
        return out

    def state_dict(self):
        sd = super().state_dict()
        if self._step > 0:
            step = torch.tensor(float(self._step))
            sd["state"] = {i: {"step": step.clone(), "exp_avg": m, "exp_avg_sq": v}
                           for i, (m, v) in enumerate(self._moment_views())}
        return sd

    @torch.no_grad()
    def load_state_dict(self, state_dict):
        """Accepts what state_dict() returns, what torch.optim.AdamW saves for the same groups (the reference's
        checkpoints), and this class's older {"flat": ...} form.  The caller's dict is left as it was."""
        sd = dict(state_dict)
        flat = sd.pop("flat", None)
        state = sd.get("state") or {}
        sd["state"] = {}
        # validate everything first: a rejected checkpoint must leave the live moments and step count as they were
        step = 0
        if flat is None and state:
            n = sum(len(g["params"]) for g in self.param_groups)
            missing = [i for i in range(n) if i not in state]
            if missing:
                raise ValueError("FlatAdamW.load_state_dict: no state for parameters %s (a checkpoint of an optimizer "
                                 "that skipped them?)" % missing[:8])
            if any(state[i].get("max_exp_avg_sq") is not None for i in range(n)):
                raise ValueError("FlatAdamW.load_state_dict: amsgrad state is not supported")
            # one host read for all step counters (they may be device tensors)
            raw = [state[i]["step"] for i in range(n)]
            if all(torch.is_tensor(t) for t in raw):
                values = torch.stack([t.detach().reshape(()).double() for t in raw]).tolist()
            else:
                values = [float(t) for t in raw]
            steps = {int(v) for v in values}
            if len(steps) != 1:
                raise ValueError("FlatAdamW.load_state_dict: per-parameter step counts differ (%s); this optimizer "
                                 "keeps one" % sorted(steps)[:8])
            step = steps.pop()
        super().load_state_dict(sd)
        if flat is not None:
            self._step = int(flat["step"])
            self.exp_avg.copy_(flat["exp_avg"])
            self.exp_avg_sq.copy_(flat["exp_avg_sq"])
            return
        self.exp_avg.zero_(), self.exp_avg_sq.zero_()
        self._step = step
        if not state:
            return
        for i, (m, v) in enumerate(self._moment_views()):
            m.copy_(state[i]["exp_avg"])
            v.copy_(state[i]["exp_avg_sq"])
