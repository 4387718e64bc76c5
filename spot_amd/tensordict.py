"""Thin host containers with the reference's API surface.

TensorDict: ltr/utils/tensordict.py:6-35 (an OrderedDict whose unknown attributes
fan out to the tensors it holds, e.g. ``td.to(device)``).
NestedTensor: ltr/utils/misc.py:347-367 (tensor + padding mask pair).
"""
from collections import OrderedDict

import torch


class TensorDict(OrderedDict):
    """Ordered name -> tensor mapping with the public methods of ltr/utils/tensordict.py:6-31.

    Any ``torch.Tensor`` method name read as an attribute gives a callable that invokes that method on every
    entry which has it and returns the results under the same names (``batch.to("cuda")``, ``batch.float()``);
    entries without the method (lists, scalars) pass through untouched.  Other attribute names raise
    ``AttributeError``, so ``hasattr`` and pickling probes behave."""

    def _mapped(self, fn):
        out = TensorDict()
        for name, entry in self.items():
            out[name] = fn(entry)
        return out

    def concat(self, other):
        """A new container with this one's entries followed by ``other``'s (its values win on equal names);
        the tensors themselves are shared, not copied."""
        merged = TensorDict(self)
        merged.update(other)
        return merged

    def copy(self):
        """Shallow copy that stays a TensorDict (``OrderedDict.copy`` of a subclass already does on CPython,
        but the reference's callers rely on the type, so it is explicit)."""
        return TensorDict(self.items())

    def __getattr__(self, name):
        # only reached for names that are neither instance attributes nor OrderedDict's
        if name.startswith("__") or not hasattr(torch.Tensor, name):
            raise AttributeError("%s has no attribute %r (only torch.Tensor method names fan out to the entries)"
                                 % (type(self).__name__, name))

        def fan_out(*args, **kwargs):
            def call(entry):
                method = getattr(entry, name, None)
                return entry if method is None else method(*args, **kwargs)
            return self._mapped(call)
        fan_out.__name__ = name
        return fan_out

    def attribute(self, attr, *default):
        """``getattr(entry, attr[, default])`` of every entry (``batch.attribute("shape")``)."""
        return self._mapped(lambda entry: getattr(entry, attr, *default))

    def apply(self, fn, *args, **kwargs):
        """``fn(entry, *args, **kwargs)`` of every entry."""
        return self._mapped(lambda entry: fn(entry, *args, **kwargs))


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
