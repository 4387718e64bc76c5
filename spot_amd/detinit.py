"""Deterministic, platform-independent parameter filler.

Real SPOT checkpoints are not available offline, so parity tests and the
benchmark fill a module's ``state_dict`` from a counter-based generator keyed
by (seed, tensor name).  The same call fills the reference module when the
golden vectors are generated and this package's module on the GPU box, which
works because both expose identical state-dict key names
(reference: ltr/models/tracking/transt_model.py:47-56,
ltr/models/neck/transt_fusion.py:118-133,169-199).

Magnitudes are chosen so that activations stay O(1) through the ResNet
bottleneck stack and the fusion layers (no saturated softmax, no exploding
residual stream), which keeps the goldens well conditioned.
"""
import zlib

import numpy as np
import torch


def _rng(name, seed):
    return np.random.default_rng([int(seed), zlib.crc32(name.encode("utf-8"))])


def det_array(name, shape, seed=0):
    """Return a float64 numpy array for state-dict entry ``name``."""
    shape = tuple(int(s) for s in shape)
    rng = _rng(name, seed)
    leaf = name.rsplit(".", 1)[-1]
    parent = name.rsplit(".", 1)[0] if "." in name else ""
    is_norm = ("bn" in parent.rsplit(".", 1)[-1] or "norm" in parent.rsplit(".", 1)[-1]
               or parent.endswith("downsample.1"))
    if leaf == "num_batches_tracked":
        return np.zeros(shape)
    if leaf == "running_mean":
        return rng.uniform(-0.1, 0.1, shape)
    if leaf == "running_var":
        return rng.uniform(0.5, 1.5, shape)
    if is_norm and leaf == "weight":
        # last BN of a bottleneck: keep the residual branch small so the
        # 13-block residual stack does not blow up with random weights
        if parent.endswith("bn3"):
            return rng.uniform(0.1, 0.3, shape)
        return rng.uniform(0.5, 1.5, shape)
    if leaf in ("bias", "in_proj_bias"):
        return rng.uniform(-0.1, 0.1, shape)
    if len(shape) >= 2:
        fan_in = int(np.prod(shape[1:]))
        conv = len(shape) == 4 and "backbone" in name
        bound = np.sqrt((6.0 if conv else 3.0) / fan_in)
        return rng.uniform(-bound, bound, shape)
    return rng.uniform(-0.1, 0.1, shape)


def det_state_dict(module, seed=0):
    """Build a state dict with the keys/shapes/dtypes of ``module``."""
    out = {}
    for k, v in module.state_dict().items():
        a = det_array(k, v.shape, seed)
        out[k] = torch.from_numpy(np.ascontiguousarray(a)).to(v.dtype).reshape(v.shape)
    return out


def det_fill_(module, seed=0):
    module.load_state_dict(det_state_dict(module, seed))
    return module


def det_fill_positive_norm_(module, seed=0):
    """det_fill_, then every BatchNorm scale mapped to 0.5 + 5 |w|: for modules whose normalisation layers sit in
    nn.Sequential containers (names like ``tower.1.weight``), which the name rules above do not recognise."""
    det_fill_(module, seed)
    with torch.no_grad():
        for m in module.modules():
            if isinstance(m, torch.nn.modules.batchnorm._BatchNorm) and m.weight is not None:
                m.weight.copy_(0.5 + 5.0 * m.weight.abs())
    return module


def det_input(name, shape, seed=0, lo=0.0, hi=1.0, dtype=np.float32):
    """Deterministic synthetic input tensor (numpy)."""
    rng = _rng("input:" + name, seed)
    if np.issubdtype(np.dtype(dtype), np.integer):
        return rng.integers(int(lo), int(hi), size=shape, dtype=dtype)
    return rng.uniform(lo, hi, size=shape).astype(dtype)
