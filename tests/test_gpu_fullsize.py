"""Properties that hold at any size, checked at the sizes the benchmark configurations run at (where an element-wise
CPU reference would take minutes): linearity, permutation equivariance, idempotence, agreement of two
independently written kernels, and checksums of checksums.  Needs a real MI355X."""
import numpy as np
import pytest
import torch

from spot_amd.detinit import det_fill_
from spot_amd.synthetic import Settings, unsup_batch

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def ops():
    from spot_amd import lib, ops as _ops
    lib.load()
    return _ops


def _r(seed, *shape):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1).to(DEV)


def test_groupdw_head_is_linear_and_matches_its_parts_at_cfg5_size(ops):
    """384^2 search, B=128, C=256: the fused head equals the softmax-weighted sum of three single correlations,
    is linear in the search features, and its gradients are those of that sum."""
    B, Cn, hx, hz = 128, 256, 45, 5
    xs = [_r(1, B, Cn, hx, hx), _r(2, B, Cn, hx - 2, hx), _r(3, B, Cn, hx, hx - 2)]
    zs = [_r(4, B, Cn, hz, hz), _r(5, B, Cn, hz - 2, hz), _r(6, B, Cn, hz, hz - 2)]
    w = torch.tensor([0.3, -0.2, 0.5], device=DEV)
    with torch.no_grad():
        y = ops.groupdw(zs, xs, w)
        sm = torch.softmax(w, 0)
        parts = sum(sm[i] * ops.xcorr_depthwise(xs[i], zs[i]) for i in range(3))
        assert y.shape == (B, Cn, hx - hz + 1, hx - hz + 1)
        np.testing.assert_allclose(y.cpu().numpy(), parts.cpu().numpy(), rtol=1e-5, atol=1e-5)
        xs2 = [_r(10 + i, *x.shape) for i, x in enumerate(xs)]
        y2 = ops.groupdw(zs, xs2, w)
        ysum = ops.groupdw(zs, [a + 2.0 * b for a, b in zip(xs, xs2)], w)
        np.testing.assert_allclose(ysum.cpu().numpy(), (y + 2.0 * y2).cpu().numpy(), rtol=1e-4, atol=1e-4)
    up = _r(20, *y.shape)
    leaves = [t.clone().requires_grad_() for t in xs + zs + [w]]
    (ops.groupdw(leaves[3:6], leaves[:3], leaves[6]) * up).sum().backward()
    ref = [t.clone().requires_grad_() for t in xs + zs + [w]]
    smr = torch.softmax(ref[6], 0)
    (sum(smr[i] * ops.xcorr_depthwise(ref[i], ref[3 + i]) for i in range(3)) * up).sum().backward()
    for a, b in zip(leaves, ref):
        scale = float(b.grad.abs().max())
        np.testing.assert_allclose(a.grad.cpu().numpy(), b.grad.cpu().numpy(), rtol=0, atol=2e-4 * scale)


def test_teacher_pipeline_is_permutation_equivariant_at_cfg2_size():
    """32 clips through the teacher in eval() mode, decode and filter: permuting the clips permutes every output
    bit for bit (no arithmetic depends on a clip's position in the batch)."""
    from spot_amd.actor import TranstSparseActor
    from spot_amd.middleware import SparseMiddleware
    from spot_amd.tensordict import TensorDict
    from spot_amd.transt import transt_resnet50
    S = Settings()
    net = det_fill_(transt_resnet50(Settings(device="cpu")), 31).to(DEV).eval()
    actor = TranstSparseActor(S, net, net, None)
    mw = SparseMiddleware(S)
    B = 32
    data, mid = unsup_batch(B, 13, device=DEV)
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(3)).to(DEV)

    def run(order):
        d = TensorDict({k: (v if order is None else v[order]).contiguous() for k, v in data.items()
                        if k.startswith(("key_", "unlabeled_"))})
        m = {k: (v if order is None else v[order]).contiguous() for k, v in mid.items()}
        pseudo = actor(mode="inference", teacher_data=d)
        return pseudo, mw.record_relay_results(m, pseudo)

    with torch.no_grad():
        p0, r0 = run(None)
        p1, r1 = run(perm)
    for k in ("best_bbox", "best_score", "best_conf"):
        assert torch.equal(p0[k][perm], p1[k]), k
    assert torch.equal(r0["aug_success"][perm], r1["aug_success"])
    assert torch.equal(r0["aug_anno"][perm], r1["aug_anno"])
    assert torch.isfinite(p0["best_bbox"]).all() and ((p0["best_conf"] > 0) & (p0["best_conf"] < 1)).all()


def test_compaction_properties_at_cfg3_size(ops):
    """1024 clips: kept clips come first in their order, the refill cycles through them, nothing kept leaves
    everything as it was, everything kept is the identity."""
    B = 1024
    g = torch.Generator().manual_seed(7)
    keep = (torch.rand(B, generator=g) < 0.7).to(torch.uint8).to(DEV)
    idx, n = ops.compact_indices(keep, B)
    n = int(n)
    kept = torch.nonzero(keep.cpu()).flatten()
    assert n == kept.numel()
    assert torch.equal(idx.cpu()[:n].long(), kept)
    assert torch.equal(idx.cpu()[n:].long(), kept[torch.arange(B - n) % n])
    full, nf = ops.compact_indices(torch.ones(B, dtype=torch.uint8, device=DEV), B)
    assert int(nf) == B and torch.equal(full.cpu().long(), torch.arange(B))
    rows = _r(8, B, 4)
    out = ops.gather_rows(rows, idx)
    assert torch.equal(out, rows[idx.long()])


def test_ema_endpoints_and_contraction_over_the_whole_network(ops):
    """23 M parameters in flat storage: keep_rate 1 leaves the teacher, keep_rate 0 copies the student, and one
    update at 0.998 moves every parameter 0.2 % of the way."""
    from spot_amd.arena import ParamArena
    from spot_amd.transt import transt_resnet50
    stu = det_fill_(transt_resnet50(Settings(device="cpu")), 41).to(DEV)
    tea = det_fill_(transt_resnet50(Settings(device="cpu")), 42).to(DEV)
    a_s, a_t = ParamArena(stu), ParamArena(tea)
    assert a_s.flat.numel() > 20_000_000
    t0 = a_t.flat.clone()
    ops.ema_lerp_(a_t.flat, a_s.flat, 1.0)
    assert torch.equal(a_t.flat, t0)
    ops.ema_lerp_(a_t.flat, a_s.flat, 0.998)
    moved = (a_t.flat - t0).double().abs().sum().item() / (a_s.flat - t0).double().abs().sum().item()
    assert abs(moved - 0.002) < 2e-5, moved
    assert float((a_t.flat - (t0 * 0.998 + a_s.flat * (1 - 0.998))).abs().max()) <= 1e-6
    ops.ema_lerp_(a_t.flat, a_s.flat, 0.0)
    assert torch.equal(a_t.flat, a_s.flat)
    # the module's parameters are views of the flat storage: they moved with it
    assert torch.equal(next(tea.parameters()), next(stu.parameters()))


def test_two_convolution_paths_agree_at_layer3_size(ops):
    """layer3's 3x3 convolution at the student batch (40 x 256 x 32 x 32): the Winograd path and the patch-matrix +
    GEMM path are separate kernels and must agree, forward and both gradients; both are linear."""
    B, Cn, H = 40, 256, 32
    x, w, dy = _r(1, B, Cn, H, H), _r(2, Cn, Cn, 3, 3) * 0.05, _r(3, B, Cn, H, H)
    outs = []
    for fn in (lambda a, b: ops.Conv3x3Fn.apply(a, b, 1), lambda a, b: ops.ConvIm2colFn.apply(a, b, 1, 1, 1, None)):
        xa, wa = x.clone().requires_grad_(), w.clone().requires_grad_()
        y = fn(xa, wa)
        y.backward(dy)
        outs.append((y.detach(), xa.grad, wa.grad))
    for a, b in zip(*outs):
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=0, atol=3e-5 * float(b.abs().max()))
    with torch.no_grad():
        x2 = _r(4, B, Cn, H, H)
        ysum = ops.Conv3x3Fn.apply(x + 3.0 * x2, w, 1)
        y2 = ops.Conv3x3Fn.apply(x2, w, 1)
    np.testing.assert_allclose(ysum.cpu().numpy(), (outs[0][0] + 3.0 * y2).cpu().numpy(), rtol=0,
                               atol=3e-5 * float(ysum.abs().max()))


def test_gemm_checksums_at_ffn_size(ops):
    """FFN linear1 of the student step (M = 40960, N = 2048, K = 256): row and column checksums of C equal the
    products with the operands' checksums (C 1 = A (B^T 1), 1^T C = (1^T A) B^T), in float64; both workgroup tiles."""
    M, N, K = 40960, 2048, 256
    A, Bm = _r(1, M, K), _r(2, N, K)
    Ad, Bd = A.double(), Bm.double()
    for tile in (64, 128):
        Cc = torch.empty(M, N, device=DEV)
        ops.gemm(A, Bm, Cc, M, N, K, tile=tile)
        Cd = Cc.double()
        np.testing.assert_allclose(Cd.sum(1).cpu().numpy(), (Ad @ Bd.sum(0)).cpu().numpy(), rtol=0, atol=2e-3)
        np.testing.assert_allclose(Cd.sum(0).cpu().numpy(), (Ad.sum(0) @ Bd.t()).cpu().numpy(), rtol=0, atol=2e-2)
        rows = torch.arange(0, M, 4099, device=DEV)
        np.testing.assert_allclose(Cd[rows].cpu().numpy(), (Ad[rows] @ Bd.t()).cpu().numpy(), rtol=0, atol=2e-5)


def test_teacher_shard_of_128_clips_at_cfg3_size():
    """One rank's shard of the 1024-clip pseudo-labelling configuration (128 clips, eval mode): permuting the clips
    permutes every output bit for bit, and the shard processed as four blocks of 32 clips gives the same bits as the
    whole shard at once (no kernel's arithmetic depends on the batch size)."""
    from spot_amd.actor import TranstSparseActor
    from spot_amd.dist import shard_range
    from spot_amd.middleware import SparseMiddleware
    from spot_amd.tensordict import TensorDict
    from spot_amd.transt import transt_resnet50
    S = Settings()
    net = det_fill_(transt_resnet50(Settings(device="cpu")), 33).to(DEV).eval()
    actor = TranstSparseActor(S, net, net, None)
    mw = SparseMiddleware(S)
    B = 128
    data, mid = unsup_batch(B, 17, device=DEV)
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(5)).to(DEV)

    def run(sel):
        d = TensorDict({k: v[sel].contiguous() for k, v in data.items() if k.startswith(("key_", "unlabeled_"))})
        m = {k: v[sel].contiguous() for k, v in mid.items()}
        pseudo = actor(mode="inference", teacher_data=d)
        return pseudo, mw.record_relay_results(m, pseudo)

    with torch.no_grad():
        # centre the confidences on the threshold so that both outcomes of the filter occur
        conf = run(slice(0, B))[0]["best_conf"].clamp(1e-6, 1 - 1e-6)
        net.iou_embed.layers[2].bias += float(np.log(0.85 / 0.15)) - torch.log(conf / (1 - conf)).median()
        p0, r0 = run(slice(0, B))
        p1, r1 = run(perm)
        parts = [run(slice(*shard_range(B, r, 4))) for r in range(4)]
    assert 0 < int(r0["aug_success"].sum()) < B
    for k in ("best_bbox", "best_score", "best_conf"):
        assert torch.equal(p0[k][perm], p1[k]), k
        assert torch.equal(torch.cat([p[k] for p, _ in parts]), p0[k]), k
    for k in ("aug_success", "aug_anno"):
        assert torch.equal(r0[k][perm], r1[k]), k
        assert torch.equal(torch.cat([r[k] for _, r in parts]), r0[k]), k


def test_student_step_of_64_pairs_with_bf16_attention_at_cfg4_size():
    """The training configuration's per-GPU batch (16 labelled + 48 pseudo-labelled pairs) with bf16 attention products
    against the same step with exact fp32 products: loss within 1e-2 relative, gradient cosine >= 0.999, and the bf16
    kernels really ran."""
    from spot_amd.actor import TranstSparseActor
    from spot_amd.criterion import transt_loss
    from spot_amd.tensordict import TensorDict
    from spot_amd.transt import transt_resnet50
    from tests.test_oracle_model import student_batches
    S = Settings(dropout=0.0)
    sup, sp = student_batches(41, 16, 48)
    grads, losses = [], []
    for dtype in ("f32", "bf16"):
        net = det_fill_(transt_resnet50(Settings(device="cpu", dropout=0.0)), 41).to(DEV).train()
        net.featurefusion_network.set_attention_dtype(dtype)
        actor = TranstSparseActor(S, net, net, transt_loss(S))
        loss, _ = actor(mode="spsup", burnin_data=TensorDict({k: v.to(DEV) for k, v in sup.items()}),
                        sp_data=TensorDict({k: v.to(DEV) for k, v in sp.items()}))
        loss.backward()
        losses.append(loss.item())
        grads.append(torch.cat([p.grad.flatten() for _, p in net.named_parameters() if p.grad is not None]).double())
        del net, actor, loss
        torch.cuda.empty_cache()
    assert np.isfinite(losses).all()
    np.testing.assert_allclose(losses[1], losses[0], rtol=1e-2)
    cos = float((grads[0] @ grads[1]) / (grads[0].norm() * grads[1].norm()))
    assert cos >= 0.999, cos
    assert not torch.equal(grads[0], grads[1])
