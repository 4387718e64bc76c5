"""Thin host containers kept from the reference's API surface.

TensorDict: ltr/utils/tensordict.py:6-35 (an OrderedDict whose unknown attributes
fan out to the tensors it holds, e.g. ``td.to(device)``).
NestedTensor: ltr/utils/misc.py:347-367 (tensor + padding mask pair).
"""
from collections import OrderedDict

import torch


class TensorDict(OrderedDict):
    def concat(self, other):
        return TensorDict(self, **other)

    def copy(self):
        return TensorDict(super().copy())

    def __getattr__(self, name):
        if not hasattr(torch.Tensor, name):
            raise AttributeError("'TensorDict' object has not attribute '{}'".format(name))

        def apply_attr(*args, **kwargs):
            return TensorDict({n: getattr(e, name)(*args, **kwargs) if hasattr(e, name) else e
                               for n, e in self.items()})
        return apply_attr

    def attribute(self, attr, *args):
        return TensorDict({n: getattr(e, attr, *args) for n, e in self.items()})

    def apply(self, fn, *args, **kwargs):
        return TensorDict({n: fn(e, *args, **kwargs) for n, e in self.items()})


class NestedTensor(object):
    def __init__(self, tensors, mask):
        self.tensors = tensors
        self.mask = mask

    def to(self, device):
        return NestedTensor(self.tensors.to(device), self.mask.to(device) if self.mask is not None else None)

    def decompose(self):
        return self.tensors, self.mask

    def __repr__(self):
        return str(self.tensors)


def nested_tensor_from_tensor(t):
    """Equal-size batch -> NestedTensor with an all-False padding mask (misc.py:283-324)."""
    if t.dim() != 4:
        raise ValueError("not supported")
    b, _, h, w = t.shape
    return NestedTensor(t, torch.zeros((b, h, w), dtype=torch.bool, device=t.device))
