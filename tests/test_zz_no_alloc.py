"""The allocation-free split-key attention forward (static device scratch) against the workspace variant it shares its
kernels with: bit-identical outputs on two calls in a row, the second with a smaller input (the scratch then still
holds the first call's partial sums), and a clear error beyond the scratch's size."""
import math

import numpy as np
import pytest
import torch

from spot_amd.detinit import det_input

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _both(ops, L, B, Lq, Lk, nh, nsplit, tag, masked):
    dh = 32
    C = nh * dh
    G = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    q, k, v = (G(det_input(tag + n, (B, L_, C), 7, -2, 2)).view(-1, C) for n, L_ in (("q", Lq), ("k", Lk), ("v", Lk)))
    kpm = np.zeros((B, Lk), np.uint8)
    if masked:
        kpm[:, :min(64, Lk - 1)] = 1
    kp = G(kpm)
    P = ops._p
    scale = 1.0 / math.sqrt(dh)
    ref = torch.empty((B * Lq, C), device=DEV)
    ws = torch.empty((L.load().spot_mha_fwd_splitk_ws_floats(B, Lq, nh, nsplit),), device=DEV)
    L.call("spot_mha_fwd_splitk_f32", P(q), C, P(k), C, P(v), C, P(ref), C, P(kp), P(ws), nsplit, B, Lq, Lk, nh, dh, scale,
           ops._st())
    out = torch.full((B * Lq, C), float("nan"), device=DEV)
    args = (P(q), C, P(k), C, P(v), C, P(out), C, P(kp), nsplit, B, Lq, Lk, nh, dh, scale, ops._st())
    if not _both.warm:
        L.call("spot_mha_fwd_splitk_noalloc_f32", *args)      # the first call on a device resolves the array's address
        _both.warm = True
    torch.cuda.synchronize()
    # free device memory as the driver sees it (hipMemGetInfo): a hipMalloc / hipFree inside the library would move
    # it, which the framework's own allocator statistics cannot see
    free_before, stats_before = torch.cuda.mem_get_info()[0], torch.cuda.memory_stats()["allocation.all.allocated"]
    L.call("spot_mha_fwd_splitk_noalloc_f32", *args)
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == free_before
    assert torch.cuda.memory_stats()["allocation.all.allocated"] == stats_before
    assert torch.isfinite(out).all()
    assert torch.equal(out, ref)


_both.warm = False


def test_noalloc_split_key_attention_equals_workspace_variant_twice():
    from spot_amd import lib as L, ops
    L.load()
    # the tracker's largest call fills the scratch exactly: 8 ranges x 8 heads x 1024 queries
    assert L.load().spot_mha_fwd_splitk_ws_floats(1, 1024, 8, 8) == 2228224
    _both(ops, L, 1, 1024, 1024, 8, 8, "a", masked=False)
    # a smaller, ragged call next: the array still holds the first call's partial sums
    _both(ops, L, 1, 50, 77, 4, 3, "b", masked=True)
    _both(ops, L, 2, 256, 1024, 8, 4, "c", masked=True)


def test_noalloc_rejects_sizes_beyond_the_scratch():
    from spot_amd import lib as L, ops
    l = L.load()
    x = torch.zeros((2 * 1024, 256), device=DEV)
    rc = l.spot_mha_fwd_splitk_noalloc_f32(ops._p(x), 256, ops._p(x), 256, ops._p(x), 256, ops._p(x), 256, None, 8, 2, 1024,
                                           1024, 8, 32, 0.1, ops._st())
    assert rc < 0 and b"exceeds the static scratch" in l.spot_last_error()
