"""Parity of each HIP kernel (called through the C ABI) against the CPU oracle and
the golden vectors produced by the reference.  Needs a real MI355X."""
import math

import numpy as np
import pytest
import torch

from oracle import spot_oracle as O
from spot_amd.detinit import det_array, det_input

pytestmark = pytest.mark.gpu

DEV = "cuda"


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def G(a):
    return T(a).to(DEV)


@pytest.fixture(scope="module")
def ops():
    from spot_amd import ops as _ops, lib
    lib.load()
    return _ops


# --------------------------------------------------------------------------- pseudo-label path
def test_decode_filter_golden(ops, golden):
    g = golden("g6_decode")
    lg, bx, io = G(g["logits"]), G(g["boxes"]), G(g["ious"].reshape(16, 1024))
    bbox, score, conf, idx = ops.decode_teacher(lg, bx, io, 0.40, 32, 256, 256)
    dec = O.decode_teacher(T(g["logits"]), T(g["boxes"]), T(g["ious"]), 0.40, 32, 256, 256)
    assert np.array_equal(idx.cpu().numpy(), dec["best_idx"].numpy().astype(np.int32))
    np.testing.assert_allclose(bbox.cpu().numpy(), g["best_bbox"], rtol=1e-6, atol=1e-5)
    np.testing.assert_allclose(score.cpu().numpy(), g["best_score"], rtol=1e-6)
    # middleware on the golden's confidences (tau-eps, tau, tau+eps hand-placed at 3..5)
    conf_g = G(g["best_conf"])
    rec, aug, keep = ops.middleware_filter(G(g["best_bbox"]), conf_g, G(g["mid_unlabeled_box_extract"].reshape(16, 4)),
                                           G(g["mid_unlabeled_resize_factors"]),
                                           G(g["mid_aug_box_extract"].reshape(16, 4)),
                                           G(g["mid_aug_resize_factors"]), 256, 0.85)
    assert np.array_equal(keep.cpu().numpy().astype(bool), g["aug_success"])
    np.testing.assert_allclose(aug.cpu().numpy(), g["aug_anno"], rtol=1e-6, atol=1e-4)
    # fused variant must agree with the two-call variant bit for bit
    fused = ops.decode_filter(lg, bx, io, 0.40, 32, 256, 256, G(g["mid_unlabeled_box_extract"].reshape(16, 4)),
                              G(g["mid_unlabeled_resize_factors"]), G(g["mid_aug_box_extract"].reshape(16, 4)),
                              G(g["mid_aug_resize_factors"]), 256, 0.85)
    rec2, aug2, keep2 = ops.middleware_filter(bbox, conf, G(g["mid_unlabeled_box_extract"].reshape(16, 4)),
                                              G(g["mid_unlabeled_resize_factors"]),
                                              G(g["mid_aug_box_extract"].reshape(16, 4)),
                                              G(g["mid_aug_resize_factors"]), 256, 0.85)
    assert torch.equal(fused["aug_anno"], aug2) and torch.equal(fused["aug_success"], keep2)
    assert torch.equal(fused["best_bbox"], bbox) and torch.equal(fused["bbox_recovered"], rec2)


def test_decode_random_vs_oracle(ops):
    B, L = 128, 1024
    lg = det_input("lg", (B, L, 2), 5, -6, 6)
    bx = det_input("bx", (B, L, 4), 5, 0, 1)
    io = det_input("io", (B, L, 1), 5, 0, 1)
    bbox, score, conf, idx = ops.decode_teacher(G(lg), G(bx), G(io.reshape(B, L)), 0.40, 32, 256, 256)
    dec = O.decode_teacher(T(lg), T(bx), T(io), 0.40, 32, 256, 256)
    # exclude clips whose top-2 blended scores are closer than fp32 rounding of exp()
    top2 = torch.topk(dec["score_map"], 2, dim=1).values
    ok = ((top2[:, 0] - top2[:, 1]) > 1e-6).numpy()
    assert ok.sum() > B * 0.9
    assert np.array_equal(idx.cpu().numpy()[ok], dec["best_idx"].numpy().astype(np.int32)[ok])
    np.testing.assert_allclose(bbox.cpu().numpy()[ok], dec["best_bbox"].numpy()[ok], rtol=1e-6, atol=1e-5)
    np.testing.assert_allclose(conf.cpu().numpy()[ok], dec["best_conf"].numpy()[ok], rtol=0, atol=0)
    np.testing.assert_allclose(score.cpu().numpy(), dec["best_score"].numpy(), rtol=2e-6)


def test_decode_with_nan_scores_stays_in_range_and_filters_the_clip(ops):
    """A diverged teacher: score maps that are NaN everywhere or in places.  torch.max returns the first NaN's index and
    a NaN score; the clip's confidence then fails the threshold.  The kernels must do the same and never read outside
    the clip's map."""
    from spot_amd import lib as L_
    B, L = 5, 1024
    lg = det_input("lg", (B, L, 2), 8, -3, 3)
    bx = det_input("bx", (B, L, 4), 8, 0.1, 0.9)
    io = det_input("io", (B, L, 1), 8, 0.9, 1.0)
    lg[0] = np.nan                                  # every score NaN -> token 0
    lg[1, 700, 0] = np.nan
    lg[1, 300, 1] = np.nan                          # two NaN scores -> the first one, token 300
    io[2, :, 0] = np.nan                            # finite scores, NaN confidences -> not kept
    lg[3, 1023] = np.nan                            # the last token only
    ue = np.concatenate([det_input("uxy", (B, 2), 8, 0, 400), det_input("uwh", (B, 2), 8, 20, 200)], -1).astype(np.float32)
    rf = det_input("rf", (B,), 8, 0.5, 2)
    res = ops.decode_filter(G(lg), G(bx), G(io.reshape(B, L)), 0.40, 32, 256, 256, G(ue), G(rf), G(ue), G(rf), 256, 0.85)
    dec = O.decode_teacher(T(lg), T(bx), T(io), 0.40, 32, 256, 256)
    assert dec["best_idx"].tolist()[:2] == [0, 300] and dec["best_idx"].tolist()[3] == 1023
    assert np.array_equal(res["best_idx"].cpu().numpy(), dec["best_idx"].numpy().astype(np.int32))
    np.testing.assert_array_equal(np.isnan(res["best_score"].cpu().numpy()), [True, True, False, True, False])
    np.testing.assert_array_equal(res["best_conf"].cpu().numpy(), dec["best_conf"].numpy())
    np.testing.assert_array_equal(res["best_bbox"].cpu().numpy(), dec["best_bbox"].numpy())
    flt = O.middleware_filter(dec["best_bbox"], dec["best_conf"], T(ue), T(rf), T(ue), T(rf), 256, 0.85)
    assert res["aug_success"].cpu().numpy().astype(bool).tolist() == flt["aug_success"].tolist() == [True, True, False, True, True]
    # the online tracker's decode: numpy.argmax picks the first NaN too
    h = np.hanning(32)
    window = np.outer(h, h).flatten()
    for i in (0, 1, 3, 4):
        best6 = torch.empty((6,), dtype=torch.float64, device=DEV)
        lg_i, bx_i, win = G(lg[i:i + 1]), G(bx[i:i + 1]), G(window)          # named: they must outlive the launch
        L_.call("spot_tracker_decode", ops._p(lg_i), ops._p(bx_i), ops._p(win), 0.49, ops._p(best6), None, 1, L, ops._st())
        _, pscore, _, _, _ = O.tracker_post(lg[i], bx[i], window, 0.49, np.array([100.0, 100.0]), np.array([50.0, 50.0]), (300, 400))
        assert int(best6[1].item()) == int(np.argmax(pscore))
        np.testing.assert_array_equal(best6[2:].cpu().numpy(), bx[i, int(np.argmax(pscore))].astype(np.float64))


def test_decode_ragged_L(ops):
    # token counts that are not a multiple of the workgroup size, B = 1 and B = 0
    for B, n in ((1, 5), (3, 9), (2, 20)):
        L = n * n
        lg = det_input("lg", (B, L, 2), 6, -3, 3)
        bx = det_input("bx", (B, L, 4), 6, 0, 1)
        io = det_input("io", (B, L, 1), 6, 0, 1)
        bbox, score, conf, idx = ops.decode_teacher(G(lg), G(bx), G(io.reshape(B, L)), 0.3, n, 64, 48)
        dec = O.decode_teacher(T(lg), T(bx), T(io), 0.3, n, 64, 48)
        assert np.array_equal(idx.cpu().numpy(), dec["best_idx"].numpy().astype(np.int32))
        np.testing.assert_allclose(bbox.cpu().numpy(), dec["best_bbox"].numpy(), rtol=1e-6, atol=1e-5)
    e = torch.empty((0, 16, 2), device=DEV)
    out = ops.decode_teacher(e, torch.empty((0, 16, 4), device=DEV), torch.empty((0, 16), device=DEV), 0.4, 4, 8, 8)
    assert out[0].shape == (0, 4)


def test_compaction_golden(ops, golden):
    g = golden("g7_compaction")
    for n in [k[5:] for k in g.files if k.startswith("mask_")]:
        keep = G(g["mask_" + n].astype(np.uint8))
        idx, nk = ops.compact_indices(keep, keep.numel())
        assert np.array_equal(idx.cpu().numpy(), g["idx_" + n].astype(np.int32)), n
        assert int(nk.item()) == int(g["mask_" + n].sum())
    idx, nk = ops.compact_indices(torch.zeros(24, dtype=torch.uint8, device=DEV), 24)
    assert int(nk.item()) == 0 and bool((idx == -1).all())
    # large ragged batch, different output size
    keep = det_input("k", (3001,), 9, 0, 1) > 0.7
    idx, nk = ops.compact_indices(G(keep.astype(np.uint8)), 4096)
    ref = O.compact_indices(keep, 4096)
    assert np.array_equal(idx.cpu().numpy(), ref.astype(np.int32))


def test_gather_rows(ops):
    src = G(det_input("src", (24, 3, 16, 16), 3, 0, 256, np.uint8))
    idx = torch.tensor([5, 0, 23, 5, 7], dtype=torch.int32, device=DEV)
    assert torch.equal(ops.gather_rows(src, idx), src[idx.long()])
    srcf = G(det_input("srcf", (24, 3), 3, 0, 1))       # 12-byte rows: byte path
    assert torch.equal(ops.gather_rows(srcf, idx), srcf[idx.long()])
    none = torch.full((5,), -1, dtype=torch.int32, device=DEV)      # nothing was kept: defined, all-zero rows
    assert not ops.gather_rows(src, none).any() and not ops.gather_rows(srcf, none).any()


def test_boxes_golden(ops, golden):
    g = golden("g8_boxes")
    b1, b2 = G(g["b1"]), G(g["b2"])
    giou, iou, uni = ops.box_iou_pair(b1, b2, cxcywh=True)
    np.testing.assert_allclose(giou.cpu().numpy(), g["giou"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(iou.cpu().numpy(), g["iou"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(uni.cpu().numpy(), g["union"], rtol=1e-6)
    x1, x2 = O.box_cxcywh_to_xyxy(T(g["b1"])), O.box_cxcywh_to_xyxy(T(g["b2"]))
    giou2, iou2, _ = ops.box_iou_pair(x1.to(DEV).contiguous(), x2.to(DEV).contiguous(), cxcywh=False)
    np.testing.assert_allclose(giou2.cpu().numpy(), g["giou"], rtol=1e-6, atol=1e-7)
    # backward of sum(1 - giou) w.r.t. the first box set
    s = b1.clone().requires_grad_()
    gi, _ = ops.GIoUPairFn.apply(s, b2)
    (1 - gi).sum().backward()
    np.testing.assert_allclose(s.grad.cpu().numpy(), g["dgiou_db1"], rtol=1e-4, atol=1e-5)


def test_xcorr_golden(ops, golden):
    g = golden("g12_xcorr")
    for tag in ("ocean255", "cfg5"):
        B, C, hx, hz = [int(v) for v in g["dims_" + tag]]
        shp_x = [(hx - 2, hx - 2), (hx - 4, hx - 2), (hx - 2, hx - 4)]
        shp_z = [(hz - 2, hz - 2), (hz - 4, hz - 2), (hz - 2, hz - 4)]
        xs = [G(det_input("x%d%s" % (i, tag), (B, C) + s, 71, -1, 1)) for i, s in enumerate(shp_x)]
        zs = [G(det_input("z%d%s" % (i, tag), (B, C) + s, 71, -1, 1)) for i, s in enumerate(shp_z)]
        re11 = ops.xcorr_depthwise(xs[0], zs[0])
        np.testing.assert_allclose(re11.cpu().numpy(), g["re11_" + tag], rtol=1e-4, atol=1e-5)
        s = ops.groupdw(zs, xs, G(np.asarray(g["weight"], np.float32)))
        np.testing.assert_allclose(s.cpu().numpy(), g["groupdw_" + tag], rtol=1e-4, atol=1e-5)


def test_xcorr_odd_shapes_vs_oracle(ops):
    # generic template sizes (runtime path), single plane, planes whose byte offset is not 16-aligned
    for (B, C, hx, wx, hz, wz) in ((1, 1, 9, 7, 3, 2), (2, 3, 31, 31, 7, 7), (1, 5, 12, 17, 1, 1), (3, 2, 8, 8, 8, 8)):
        x = det_input("x", (B, C, hx, wx), 7, -1, 1)
        z = det_input("z", (B, C, hz, wz), 7, -1, 1)
        out = ops.xcorr_depthwise(G(x), G(z)).cpu().numpy()
        np.testing.assert_allclose(out, O.xcorr_depthwise(x, z), rtol=1e-4, atol=1e-5)


def test_preprocess_golden(ops, golden):
    g = golden("g14_pre_pos")
    img = det_input("img", (2, 3, 16, 16), 91, 0, 256, np.uint8)
    out = ops.preprocess_u8(G(img), None, [0.485, 0.456, 0.406], [0.229, 0.224, 0.225])
    np.testing.assert_allclose(out.cpu().numpy(), g["img_out"], rtol=1e-6, atol=1e-6)
    # brightness jitter path, values that clamp
    img = det_input("img2", (4, 3, 32, 32), 92, 0, 256, np.uint8)
    br = np.array([0.8, 1.0, 1.2, 1.19], np.float32)
    out = ops.preprocess_u8(G(img), G(br / np.float32(255.0)), [0.485, 0.456, 0.406], [0.229, 0.224, 0.225])
    ref, _ = O.preprocess(T(img), torch.zeros(4, 3), [0.485, 0.456, 0.406], [0.229, 0.224, 0.225], T(br))
    np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), rtol=1e-6, atol=1e-6)


def test_ema(ops, golden):
    n = 100003
    t = det_input("t", (n,), 1, -1, 1)
    s = det_input("s", (n,), 2, -1, 1)
    tg = G(t)
    ops.ema_lerp_(tg, G(s), 0.998)
    ref = T(s) * (1 - 0.998) + T(t) * 0.998
    assert torch.equal(tg.cpu(), ref)
    ti = torch.tensor([100, 7, 0], dtype=torch.int64, device=DEV)
    si = torch.tensor([7, 100, 5], dtype=torch.int64, device=DEV)
    ops.ema_lerp_i64_(ti, si, 0.998)
    ref = (si.cpu() * (1 - 0.998) + torch.tensor([100, 7, 0]) * 0.998).to(torch.int64)
    assert torch.equal(ti.cpu(), ref)


# --------------------------------------------------------------------------- dense kernels
def _ref_gemm(A, B, transA, transB):
    a = A.double().t() if transA else A.double()
    b = B.double() if transB else B.double().t()
    return a @ b


@pytest.mark.parametrize("transA", [False, True])
@pytest.mark.parametrize("transB", [False, True])
def test_gemm_layouts(ops, transA, transB):
    for (M, N, K) in ((256, 256, 256), (384, 130, 96), (100, 2, 256), (37, 53, 19), (512, 768, 64), (64, 320, 40)):
        A = det_input("A", (K, M) if transA else (M, K), 1, -1, 1)
        B = det_input("B", (K, N) if transB else (N, K), 2, -1, 1)
        ref = _ref_gemm(T(A), T(B), transA, transB)
        got = {}
        for tile in (0, 64, 128):
            Cg = torch.empty((M, N), device=DEV)
            ops.gemm(G(A), G(B), Cg, M, N, K, transA=transA, transB=transB, tile=tile)
            np.testing.assert_allclose(Cg.cpu().double().numpy(), ref.numpy(), rtol=1e-5, atol=1e-5 * math.sqrt(K))
            got[tile] = Cg.cpu()
        # both workgroup tiles sum k in the same order: bit-identical results
        assert torch.equal(got[64], got[128]) and torch.equal(got[0], got[128])


@pytest.mark.parametrize("tile", [64, 128])
def test_gemm_epilogues_splitk_segments(ops, tile):
    import functools
    ops = type("TiledOps", (), {"gemm": staticmethod(functools.partial(ops.gemm, tile=tile))})
    M, N, K = 512, 256, 384
    A, B = det_input("A", (M, K), 3, -1, 1), det_input("B", (N, K), 4, -1, 1)
    bias, rb = det_input("b", (N,), 5, -1, 1), det_input("rb", (128, N), 6, -1, 1)
    res = det_input("res", (M, N), 7, -1, 1)
    base = T(A).double() @ T(B).double().t()
    Cg = torch.empty((M, N), device=DEV)
    ops.gemm(G(A), G(B), Cg, M, N, K, bias=G(bias), rowbias=G(rb), rb_rows=128, ldrb=N, relu=True, residual=G(res),
             ldres=N)
    ref = torch.relu(base + T(bias).double() + T(rb).double().repeat(M // 128, 1)) + T(res).double()
    np.testing.assert_allclose(Cg.cpu().double().numpy(), ref.numpy(), rtol=1e-5, atol=1e-4)
    mask = det_input("mask", (M, N), 8, -1, 1)
    ops.gemm(G(A), G(B), Cg, M, N, K, mask=G(mask), ldmask=N, alpha=0.5)
    ref = torch.where(T(mask) > 0, 0.5 * base, torch.zeros((), dtype=torch.float64))
    np.testing.assert_allclose(Cg.cpu().double().numpy(), ref.numpy(), rtol=1e-5, atol=1e-4)
    # split-K accumulate
    Cg.zero_()
    ops.gemm(G(A), G(B), Cg, M, N, K, splitk=4, accumulate=True)
    np.testing.assert_allclose(Cg.cpu().double().numpy(), base.numpy(), rtol=1e-5, atol=1e-4)
    # weight-gradient layout with the row sums of opA(A) (the bias gradient) taken on the way, plain and split-K
    At, Bt = det_input("At", (K, M), 13, -1, 1), det_input("Bt", (K, N), 14, -1, 1)
    for sk in (1, 3):
        Cg.zero_()
        rsum = torch.zeros((M,), device=DEV)
        ops.gemm(G(At), G(Bt), Cg, M, N, K, transA=True, transB=True, splitk=sk, accumulate=True, rowsum=rsum)
        np.testing.assert_allclose(Cg.cpu().double().numpy(), (T(At).double().t() @ T(Bt).double()).numpy(), rtol=1e-5,
                                   atol=1e-4)
        np.testing.assert_allclose(rsum.cpu().double().numpy(), T(At).double().sum(0).numpy(), rtol=1e-5, atol=1e-4)
    # ragged weight-gradient problem (guarded loader, partial tiles)
    Ar, Br = det_input("Ar", (75, 100), 15, -1, 1), det_input("Br", (75, 37), 16, -1, 1)
    Cr, rr = torch.zeros((100, 37), device=DEV), torch.zeros((100,), device=DEV)
    ops.gemm(G(Ar), G(Br), Cr, 100, 37, 75, transA=True, transB=True, accumulate=True, rowsum=rr)
    np.testing.assert_allclose(Cr.cpu().double().numpy(), (T(Ar).double().t() @ T(Br).double()).numpy(), rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(rr.cpu().double().numpy(), T(Ar).double().sum(0).numpy(), rtol=1e-5, atol=1e-4)
    # batched
    Ab, Bb = det_input("Ab", (3, 64, 32), 9, -1, 1), det_input("Bb", (3, 48, 32), 10, -1, 1)
    Cb = torch.empty((3, 64, 48), device=DEV)
    ops.gemm(G(Ab), G(Bb), Cb, 64, 48, 32, batch=3, strideA=64 * 32, strideB=48 * 32, strideC=64 * 48)
    np.testing.assert_allclose(Cb.cpu().numpy(), (T(Ab) @ T(Bb).transpose(1, 2)).numpy(), rtol=1e-5, atol=1e-5)
    # segmented K: sum over 4 slabs of a [4, K=64, M=96] x [4, N=40, K=64] pair (NCHW-style reduction)
    As, Bs = det_input("As", (4, 64, 96), 11, -1, 1), det_input("Bs", (4, 40, 64), 12, -1, 1)
    Cs = torch.empty((96, 40), device=DEV)
    ops.gemm(G(As), G(Bs), Cs, 96, 40, 256, transA=True, lda=96, ldb=64, kseg=64, segA=64 * 96, segB=40 * 64)
    ref = sum(T(As[i]).double().t() @ T(Bs[i]).double().t() for i in range(4))
    np.testing.assert_allclose(Cs.cpu().double().numpy(), ref.numpy(), rtol=1e-5, atol=1e-4)


def test_layernorm(ops):
    for rows, C in ((1000, 256), (37, 64), (8, 1024)):
        x = det_input("x", (rows, C), 1, -2, 2)
        gm, bt = det_input("g", (C,), 2, 0.5, 1.5), det_input("b", (C,), 3, -1, 1)
        dy = det_input("dy", (rows, C), 4, -1, 1)
        xg, gg, bg = G(x).requires_grad_(), G(gm).requires_grad_(), G(bt).requires_grad_()
        y = ops.layer_norm(xg, gg, bg)
        (y * G(dy)).sum().backward()
        xc, gc, bc = T(x).double().requires_grad_(), T(gm).double().requires_grad_(), T(bt).double().requires_grad_()
        yc = torch.nn.functional.layer_norm(xc, (C,), gc, bc, 1e-5)
        (yc * T(dy).double()).sum().backward()
        np.testing.assert_allclose(y.detach().cpu().numpy(), yc.detach().numpy(), rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(xg.grad.cpu().numpy(), xc.grad.numpy(), rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(gg.grad.cpu().numpy(), gc.grad.numpy(), rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(bg.grad.cpu().numpy(), bc.grad.numpy(), rtol=1e-4, atol=1e-4)


def _ref_attn(q, k, v, kpm, nhead):
    B, Lq, C = q.shape
    dh = C // nhead
    qh = q.view(B, Lq, nhead, dh).transpose(1, 2) / math.sqrt(dh)
    kh = k.view(B, -1, nhead, dh).transpose(1, 2)
    vh = v.view(B, -1, nhead, dh).transpose(1, 2)
    s = qh @ kh.transpose(-1, -2)
    if kpm is not None:
        s = s.masked_fill(kpm[:, None, None, :], float("-inf"))
    return (torch.softmax(s, -1) @ vh).transpose(1, 2).reshape(B, Lq, C)


@pytest.mark.parametrize("Lq,Lk", [(256, 256), (1024, 256), (256, 1024), (16, 64), (50, 77)])
def test_mha_core_fwd_bwd(ops, Lq, Lk):
    from spot_amd import lib as L
    B, nh, dh = 2, 4, 32
    C = nh * dh
    q, k, v = (det_input(n, (B, L_, C), 1, -2, 2) for n, L_ in (("q", Lq), ("k", Lk), ("v", Lk)))
    do = det_input("do", (B, Lq, C), 2, -1, 1)
    kpm = np.zeros((B, Lk), bool)
    kpm[1, Lk // 2:Lk // 2 + 5] = True
    kpm[1, -3:] = True
    qg, kg, vg, kp = G(q).view(-1, C), G(k).view(-1, C), G(v).view(-1, C), G(kpm.astype(np.uint8))
    out, lse = ops.mha_core_fwd(qg, C, kg, C, vg, C, B, Lq, Lk, nh, dh, kp, True)
    qc, kc, vc = (T(a).double().requires_grad_() for a in (q, k, v))
    ref = _ref_attn(qc, kc, vc, T(kpm), nh)
    np.testing.assert_allclose(out.view(B, Lq, C).cpu().numpy(), ref.detach().numpy(), rtol=1e-4, atol=1e-5)
    (ref * T(do).double()).sum().backward()
    dq, dk, dv = torch.empty_like(qg), torch.empty_like(kg), torch.empty_like(vg)
    delta = torch.empty((B, nh, Lq), device=DEV)
    dog = G(do).view(-1, C)
    P = ops._p
    L.call("spot_mha_bwd_f32", P(qg), C, P(kg), C, P(vg), C, P(out), C, P(dog), C, P(lse), P(kp), P(delta),
           P(dq), C, P(dk), C, P(dv), C, B, Lq, Lk, nh, dh, 1.0 / math.sqrt(dh), 0.0, 0, None, ops._st())
    np.testing.assert_allclose(dq.view(B, Lq, C).cpu().numpy(), qc.grad.numpy(), rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(dk.view(B, Lk, C).cpu().numpy(), kc.grad.numpy(), rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(dv.view(B, Lk, C).cpu().numpy(), vc.grad.numpy(), rtol=2e-4, atol=2e-5)


def _cos(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a @ b) / (a.norm() * b.norm()))


@pytest.mark.parametrize("Lq,Lk", [(256, 256), (1024, 256), (256, 1024), (50, 77)])
def test_mha_core_bf16_products_track_fp32(ops, Lq, Lk):
    """bf16 matrix products with fp32 softmax/accumulation: within bf16 rounding of the exact result
    (outputs 1e-2 of their scale; gradients by cosine >= 0.999, the bar of the bf16 training configuration)."""
    from spot_amd import lib as L
    B, nh, dh = 2, 4, 32
    C = nh * dh
    q, k, v = (det_input(n, (B, L_, C), 1, -2, 2) for n, L_ in (("q", Lq), ("k", Lk), ("v", Lk)))
    do = det_input("do", (B, Lq, C), 2, -1, 1)
    kpm = np.zeros((B, Lk), bool)
    kpm[1, Lk // 2:Lk // 2 + 5] = True
    qg, kg, vg, kp = G(q).view(-1, C), G(k).view(-1, C), G(v).view(-1, C), G(kpm.astype(np.uint8))
    out, lse = ops.mha_core_fwd(qg, C, kg, C, vg, C, B, Lq, Lk, nh, dh, kp, True, bf16=True)
    qc, kc, vc = (T(a).double().requires_grad_() for a in (q, k, v))
    ref = _ref_attn(qc, kc, vc, T(kpm), nh)
    o = out.view(B, Lq, C).cpu()
    assert float((o - ref.detach()).abs().max()) <= 1e-2 * float(ref.detach().abs().max())
    assert _cos(o, ref.detach()) > 0.9999
    (ref * T(do).double()).sum().backward()
    dq, dk, dv = torch.empty_like(qg), torch.empty_like(kg), torch.empty_like(vg)
    delta = torch.empty((B, nh, Lq), device=DEV)
    dog = G(do).view(-1, C)
    P = ops._p
    L.call("spot_mha_bwd_bf16", P(qg), C, P(kg), C, P(vg), C, P(out), C, P(dog), C, P(lse), P(kp), P(delta),
           P(dq), C, P(dk), C, P(dv), C, B, Lq, Lk, nh, dh, 1.0 / math.sqrt(dh), 0.0, 0, None, ops._st())
    for got, want in ((dq, qc.grad), (dk, kc.grad), (dv, vc.grad)):
        assert _cos(got.cpu(), want) > 0.999
        assert float((got.cpu().view_as(want) - want).abs().max()) <= 3e-2 * float(want.abs().max())


# --------------------------------------------------------------------------- fusion network
def _load_det(module, seed):
    from spot_amd.detinit import det_fill_
    det_fill_(module, seed)
    return module.to(DEV)


def test_mha_block_golden(ops, golden):
    """Full attention block (projections + core) against nn.MultiheadAttention outputs of the reference run."""
    from spot_amd.fusion import _AttentionParams
    g = golden("g3_mha")
    d, nh, B = 256, 8, 2
    att = _load_det(_AttentionParams(d, nh), 13)
    for tag, (Lq, Lk) in {"tt": (256, 256), "ss": (1024, 1024), "ts": (256, 1024), "st": (1024, 256)}.items():
        a = lambda n, L_: G(det_input(n + tag, (L_, B, d), 13, -1, 1)).transpose(0, 1).contiguous()
        xq, pq = a("xq", Lq), a("pq", Lq)
        self_attn = tag in ("tt", "ss")
        xk, pk = (xq, pq) if self_attn else (a("xk", Lk), a("pk", Lk))
        kpm = G(g["kpm_" + tag].astype(np.uint8))
        with torch.no_grad():
            o = att(xq, xk, pq.reshape(-1, d), pk.reshape(-1, d), kpm, False, self_attn)
        o = o.transpose(0, 1).cpu().numpy()
        np.testing.assert_allclose(o[::8], g["out_" + tag], rtol=1e-3, atol=2e-5)
        np.testing.assert_allclose(np.linalg.norm(o, axis=-1), g["rownorm_" + tag], rtol=1e-4)


def test_fusion_layer_fwd_bwd_golden(ops, golden):
    from spot_amd.fusion import FeatureFusionLayer
    g = golden("g4_layer_bwd")
    d, nh, ff, B, Lt, Ls = [int(v) for v in g["dims"]]
    layer = _load_det(FeatureFusionLayer(d, nh, ff, 0.0), 14).train()
    a = lambda n, L_: G(det_input(n, (L_, B, d), 14, -1, 1)).transpose(0, 1).contiguous()
    s1, s2 = a("s1", Lt).requires_grad_(), a("s2", Ls).requires_grad_()
    # per-sample positional rows (R = B*L), as in the golden
    o1, o2 = layer(s1, s2, None, None, a("p1", Lt).reshape(-1, d), a("p2", Ls).reshape(-1, d))
    ((o1 * a("g1", Lt)).sum() + (o2 * a("g2", Ls)).sum()).backward()
    np.testing.assert_allclose(o1.detach().transpose(0, 1).cpu().numpy(), g["out1"], rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(o2.detach().transpose(0, 1).cpu().numpy(), g["out2"], rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(s1.grad.transpose(0, 1).cpu().numpy(), g["ds1"], rtol=2e-3, atol=2e-4)
    np.testing.assert_allclose(s2.grad.transpose(0, 1).cpu().numpy(), g["ds2"], rtol=2e-3, atol=2e-4)
    for k, p in layer.named_parameters():
        ref = g["grad." + k]
        tol = 2e-3 * max(1.0, float(np.abs(ref).max()))
        np.testing.assert_allclose(p.grad.cpu().numpy(), ref, rtol=2e-3, atol=tol, err_msg=k)


def test_fusion_network_golden(ops, golden):
    from spot_amd.fusion import FeatureFusionNetwork
    g = golden("g1_fusion_tiny")
    d, nh, nl, ff, B, ht, hs = [int(v) for v in g["dims"]]
    net = _load_det(FeatureFusionNetwork(d, nh, nl, ff, 0.1), 11).eval()
    a = lambda n, s: G(det_input(n, s, 11, -1, 1))
    with torch.no_grad():
        out = net(a("src_temp", (B, d, ht, ht)), G(g["mask_temp"]), a("src_search", (B, d, hs, hs)),
                  G(g["mask_search"]), a("pos_temp", (B, d, ht, ht)), a("pos_search", (B, d, hs, hs)))
    np.testing.assert_allclose(out.cpu().numpy(), g["hs"], rtol=1e-3, atol=1e-4)
    g = golden("g2_fusion_full")
    d, nh, nl, ff, B, ht, hs = [int(v) for v in g["dims"]]
    net = _load_det(FeatureFusionNetwork(d, nh, nl, ff, 0.1), 12).eval()
    a = lambda n, s: G(det_input(n, s, 12, -1, 1))
    z = lambda *s: torch.zeros(*s, dtype=torch.bool, device=DEV)
    with torch.no_grad():
        out = net(a("src_temp", (B, d, ht, ht)), z(B, ht, ht), a("src_search", (B, d, hs, hs)), z(B, hs, hs),
                  a("pos_temp", (B, d, ht, ht)), a("pos_search", (B, d, hs, hs))).cpu().numpy()
    np.testing.assert_allclose(out[:, :, ::16, ::8], g["hs_slice"], rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(np.linalg.norm(out, axis=-1), g["hs_rownorm"], rtol=1e-4)


# --------------------------------------------------------------------------- dropout (train mode)
def test_dropout_mask_statistics_and_gemm_epilogue(ops):
    rows, cols, p = 4096, 256, 0.1
    ones = torch.ones(rows, cols, device=DEV)
    m = ops.dropout_apply(ones, p, 1234)
    keep = (m > 0).float().mean().item()
    assert abs(keep - (1 - p)) < 3e-3
    assert torch.allclose(m[m > 0], torch.tensor(1 / (1 - p), device=DEV))
    assert torch.equal(m, ops.dropout_apply(ones, p, 1234))            # pure function of (seed,row,col)
    assert not torch.equal(m, ops.dropout_apply(ones, p, 1235))
    # per-row and per-column keep rates are unbiased too
    assert (m > 0).float().mean(0).sub(1 - p).abs().max().item() < 0.03
    # GEMM epilogue dropout == dropout applied to the plain GEMM result (relu before, residual after)
    A, W, b, res = G(det_input("A", (512, 128), 1, -1, 1)), G(det_input("W", (256, 128), 2, -1, 1)), \
        G(det_input("b", (256,), 3, -1, 1)), G(det_input("r", (512, 256), 4, -1, 1))
    plain = ops.linear_fwd(A, W, b, relu=True)
    fused = ops.linear_fwd(A, W, b, relu=True, residual=res, drop_p=0.25, drop_seed=77)
    assert torch.equal(fused, ops.dropout_apply(plain, 0.25, 77) + res)


def test_mha_core_dropout_matches_explicit_mask(ops):
    from spot_amd import lib as L
    B, nh, dh, Lq, Lk, p, seed = 2, 4, 32, 96, 160, 0.2, 4242
    C = nh * dh
    q, k, v = (det_input(n, (B, L_, C), 3, -2, 2) for n, L_ in (("q", Lq), ("k", Lk), ("v", Lk)))
    do = det_input("do", (B, Lq, C), 4, -1, 1)
    mask = torch.empty((B, nh, Lq, Lk), dtype=torch.uint8, device=DEV)
    L.call("spot_mha_dropout_mask", ops._p(mask), B, nh, Lq, Lk, p, seed, ops._st())
    mk = mask.cpu().double()
    assert abs(mk.mean().item() - (1 - p)) < 5e-3
    qg, kg, vg = G(q).view(-1, C), G(k).view(-1, C), G(v).view(-1, C)
    out, lse = ops.mha_core_fwd(qg, C, kg, C, vg, C, B, Lq, Lk, nh, dh, None, True, drop_p=p, drop_seed=seed)
    qc, kc, vc = (T(a).double().requires_grad_() for a in (q, k, v))
    qh = qc.view(B, Lq, nh, dh).transpose(1, 2) / math.sqrt(dh)
    kh, vh = kc.view(B, Lk, nh, dh).transpose(1, 2), vc.view(B, Lk, nh, dh).transpose(1, 2)
    pr = torch.softmax(qh @ kh.transpose(-1, -2), -1) * mk / (1 - p)
    ref = (pr @ vh).transpose(1, 2).reshape(B, Lq, C)
    np.testing.assert_allclose(out.view(B, Lq, C).cpu().numpy(), ref.detach().numpy(), rtol=1e-4, atol=1e-5)
    (ref * T(do).double()).sum().backward()
    dq, dk, dv = torch.empty_like(qg), torch.empty_like(kg), torch.empty_like(vg)
    delta = torch.empty((B, nh, Lq), device=DEV)
    P = ops._p
    L.call("spot_mha_bwd_f32", P(qg), C, P(kg), C, P(vg), C, P(out), C, P(G(do).view(-1, C)), C, P(lse), None, P(delta),
           P(dq), C, P(dk), C, P(dv), C, B, Lq, Lk, nh, dh, 1.0 / math.sqrt(dh), p, seed, None, ops._st())
    np.testing.assert_allclose(dq.view(B, Lq, C).cpu().numpy(), qc.grad.numpy(), rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(dk.view(B, Lk, C).cpu().numpy(), kc.grad.numpy(), rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(dv.view(B, Lk, C).cpu().numpy(), vc.grad.numpy(), rtol=2e-4, atol=2e-5)


@pytest.mark.parametrize("Lq,Lk,bf16", [(96, 160, False), (1024, 256, False), (256, 1024, False), (200, 77, False),
                                        (160, 96, True)])
def test_mha_keep_bits_from_the_forward_give_the_backward_of_the_regenerated_mask(ops, Lq, Lk, bf16):
    """A forward with dropout that also writes its keep bits returns the same output, the bits are the byte mask
    (one word per query and key tile, key k at bit k, queries in the backward's row order), and a backward that reads them equals the backward that
    regenerates the mask from the seed: dK and dV bit for bit, dQ up to the order of its atomic adds."""
    from spot_amd import lib as L
    B, nh, dh, p, seed = (2, 4, 32, 0.2, 977) if Lq * Lk < 200000 else (8, 8, 32, 0.1, 31)
    C = nh * dh
    q, k, v = (G(det_input(n, (B * L_, C), 5, -2, 2)) for n, L_ in (("q", Lq), ("k", Lk), ("v", Lk)))
    do = G(det_input("do", (B * Lq, C), 6, -1, 1))
    out0, lse0 = ops.mha_core_fwd(q, C, k, C, v, C, B, Lq, Lk, nh, dh, None, True, drop_p=p, drop_seed=seed, bf16=bf16)
    keep = ops.mha_drop_bits(B, nh, Lq, Lk, DEV)
    keep.fill_(-1 if Lq % 2 else 0)                  # whatever the buffer held before must not matter
    out1, lse1 = ops.mha_core_fwd(q, C, k, C, v, C, B, Lq, Lk, nh, dh, None, True, drop_p=p, drop_seed=seed, bf16=bf16,
                                  drop_bits=keep)
    assert torch.equal(out0, out1) and torch.equal(lse0, lse1)
    # the words against the byte mask
    mask = torch.empty((B, nh, Lq, Lk), dtype=torch.uint8, device=DEV)
    L.call("spot_mha_dropout_mask", ops._p(mask), B, nh, Lq, Lk, p, seed, ops._st())
    nqt, nkt = (Lq + 31) // 32, (Lk + 31) // 32
    words = keep.view(B, nh, nqt, nkt, 32).cpu().numpy().astype(np.uint32)
    r = np.arange(32)
    slot = 2 * ((r & 3) + 4 * (r >> 3)) + ((r >> 2) & 1)               # where the word of query r (within its tile) sits
    got = ((words[..., slot, None] >> r.astype(np.uint32)) & 1).astype(np.uint8)  # [B, nh, nqt, nkt, 32 q, 32 keys]
    got = got.transpose(0, 1, 2, 4, 3, 5).reshape(B, nh, nqt * 32, nkt * 32)[:, :, :Lq, :Lk]
    assert np.array_equal(got, mask.cpu().numpy())
    P = ops._p
    entry = "spot_mha_bwd_bf16" if bf16 else "spot_mha_bwd_f32"
    res = []
    for bits in (None, keep):
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        delta = torch.empty((B, nh, Lq), device=DEV)
        L.call(entry, P(q), C, P(k), C, P(v), C, P(out1), C, P(do), C, P(lse1), None, P(delta),
               P(dq), C, P(dk), C, P(dv), C, B, Lq, Lk, nh, dh, 1.0 / math.sqrt(dh), p, seed, P(bits), ops._st())
        res.append((dq, dk, dv))
    assert torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][2], res[1][2])
    assert float((res[0][0] - res[1][0]).abs().max()) <= 1e-5 * float(res[0][0].abs().max())


def test_ffn_dropout_forward_backward_vs_explicit_masks(ops):
    M, C, F_, p = 384, 64, 128, 0.1
    x = det_input("x", (M, C), 1, -1, 1)
    W1, b1 = det_input("W1", (F_, C), 2, -0.3, 0.3), det_input("b1", (F_,), 3, -0.1, 0.1)
    W2, b2 = det_input("W2", (C, F_), 4, -0.3, 0.3), det_input("b2", (C,), 5, -0.1, 0.1)
    dz = det_input("dz", (M, C), 6, -1, 1)
    ops.manual_seed(99)
    s1, s2 = ops.next_seed(), ops.next_seed()
    m1 = ops.dropout_apply(torch.ones(M, F_, device=DEV), p, s1).cpu().double()     # mask * 1/(1-p)
    m2 = ops.dropout_apply(torch.ones(M, C, device=DEV), p, s2).cpu().double()
    ops.manual_seed(99)
    gs = [G(a).requires_grad_() for a in (x, W1, b1, W2, b2)]
    z = ops.FFNFn.apply(gs[0][None], gs[1], gs[2], gs[3], gs[4], p)[0]
    (z * G(dz)).sum().backward()
    cs = [T(a).double().requires_grad_() for a in (x, W1, b1, W2, b2)]
    h = torch.relu(cs[0] @ cs[1].t() + cs[2]) * m1
    zr = cs[0] + (h @ cs[3].t() + cs[4]) * m2
    (zr * T(dz).double()).sum().backward()
    np.testing.assert_allclose(z.detach().cpu().numpy(), zr.detach().numpy(), rtol=1e-4, atol=1e-5)
    for g_, c_ in zip(gs, cs):
        np.testing.assert_allclose(g_.grad.cpu().numpy(), c_.grad.numpy(), rtol=1e-3, atol=1e-4)


def test_fusion_layer_train_mode_dropout_is_reproducible_and_unbiased(ops):
    from spot_amd.fusion import FeatureFusionLayer
    d, nh, ff, B, Lt, Ls = 64, 2, 128, 2, 16, 64
    layer = _load_det(FeatureFusionLayer(d, nh, ff, 0.1), 14).train()
    a = lambda n, L_: G(det_input(n, (B, L_, d), 14, -1, 1))
    s1, s2, p1, p2 = a("s1", Lt), a("s2", Ls), a("p1", Lt)[0], a("p2", Ls)[0]

    def run(seed):
        ops.manual_seed(seed)
        x1, x2 = s1.clone().requires_grad_(), s2.clone().requires_grad_()
        o1, o2 = layer(x1, x2, None, None, p1, p2)
        (o1.sum() + o2.square().sum()).backward()
        return o1.detach(), o2.detach(), x1.grad, x2.grad

    r1, r2, r3 = run(5), run(5), run(6)
    for u, w in zip(r1, r2):
        assert torch.equal(u, w)
    assert not torch.equal(r1[1], r3[1])
    assert all(torch.isfinite(t).all() for t in r1)
    layer.eval()
    with torch.no_grad():
        e1, e2 = layer(s1, s2, None, None, p1, p2)
    # dropout perturbs but does not bias: the train-mode output stays close to eval on average
    assert (r1[1] - e2).abs().mean().item() < 0.5 * e2.abs().mean().item()


def test_layernorm_backward_hands_the_masked_gradient_to_the_block_before_it(ops, monkeypatch):
    """The gradient behind a block's output dropout comes out of the following LayerNorm's backward kernel as a second
    output; the block's backward uses it instead of a pass of its own.  Same gradients either way, and the stand-alone
    dropout pass is not launched when the hand-over works."""
    from spot_amd.fusion import FeatureFusionLayer
    d, nh, ff, B, Lt, Ls = 64, 2, 128, 2, 32, 96
    layer = _load_det(FeatureFusionLayer(d, nh, ff, 0.1), 21).train()
    a = lambda n, L_: G(det_input(n, (B, L_, d), 21, -1, 1))
    s1, s2, p1, p2 = a("s1", Lt), a("s2", Ls), a("p1", Lt)[0], a("p2", Ls)[0]
    calls = []
    real = ops.dropout_apply
    monkeypatch.setattr(ops, "dropout_apply", lambda x2, p, seed: (calls.append(p), real(x2, p, seed))[1])

    def run(hand_over):
        ops.manual_seed(11)
        for q in layer.parameters():
            q.grad = None
        x1, x2 = s1.clone().requires_grad_(), s2.clone().requires_grad_()
        o1, o2 = layer(x1, x2, None, None, p1, p2)
        if not hand_over:
            monkeypatch.setattr(ops, "_dropped_grad", lambda dz2, p, seed: ops.dropout_apply(dz2, p, seed))
        del calls[:]
        (o1.square().sum() + o2.sum()).backward()
        n = len(calls)
        monkeypatch.undo()
        monkeypatch.setattr(ops, "dropout_apply", lambda x2, p, seed: (calls.append(p), real(x2, p, seed))[1])
        return [x1.grad, x2.grad] + [q.grad.clone() for q in layer.parameters()], n

    with_it, n_with = run(True)
    assert not ops._DROPPED                      # every masked copy was picked up
    without, n_without = run(False)
    ops._DROPPED.clear()                         # (the copies nobody asked for; a training step clears them too)
    assert n_without >= 6 and n_with == 0, (n_with, n_without)
    for u, w in zip(with_it, without):
        # the same masked values enter the same products; weight gradients are summed by split-K atomics, whose order
        # differs from run to run in the last bit
        assert float((u - w).abs().max()) <= 2e-6 * float(w.abs().max())


@pytest.mark.parametrize("downsample", [False, True])
def test_bottleneck_bn1_inside_the_input_transform_equals_the_separate_pass(ops, downsample):
    """Training-mode bn1 + relu taken by conv2's Winograd input transform (no normalised map is written): output, input
    gradient, every parameter gradient, running statistics and the deferred-statistics variant equal the
    batch_norm_act -> conv3x3 sequence bit for bit."""
    from spot_amd.backbone import Bottleneck
    import torch.nn as nn
    inpl, planes, B, H, W = (64, 32, 3, 24, 32) if downsample else (128, 32, 3, 24, 32)
    ds = nn.Sequential(nn.Conv2d(inpl, planes * 4, 1, bias=False), nn.BatchNorm2d(planes * 4)) if downsample else None
    blk = _load_det(Bottleneck(inpl, planes, 1, ds), 33).to(DEV).train()
    for m in blk.modules():
        if isinstance(m, nn.BatchNorm2d):
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.uniform_(-0.5, 0.5)
    x0 = G(det_input("x", (B, inpl, H, W), 33, -1, 1))
    up = G(det_input("up", (B, planes * 4, H, W), 34, -1, 1))
    start = {k: v.clone() for k, v in blk.state_dict().items()}

    def run(fused, deferred):
        blk.load_state_dict(start)
        for q in blk.parameters():
            q.grad = None
        ops.FUSE_BN_INTO_CONV3X3 = fused
        try:
            x = x0.clone().requires_grad_()
            if deferred:
                with ops.defer_running_stats() as pending:
                    y = blk(x)
                pending.apply()
            else:
                y = blk(x)
            (y * up).sum().backward()
        finally:
            ops.FUSE_BN_INTO_CONV3X3 = True
        return [y.detach(), x.grad] + [q.grad.clone() for q in blk.parameters()] + [v.clone() for v in blk.state_dict().values()]

    for deferred in (False, True):
        a, b = run(True, deferred), run(False, deferred)
        assert len(a) == len(b)
        n_conv = 0
        for i, (u, w) in enumerate(zip(a, b)):
            if i >= 2 and u.dim() == 4 and u.is_floating_point() and u.shape[1] > 1 and i < 2 + len(list(blk.parameters())):
                # convolution weight gradients: summed over the batch by split-K atomics, whose order is not fixed
                assert float((u - w).abs().max()) <= 1e-5 * float(w.abs().max()), i
                n_conv += 1
            else:
                assert torch.equal(u, w), i
        assert n_conv >= 3
    with torch.no_grad():                           # a pass that needs no gradient (the teacher in its train() mode)
        blk.load_state_dict(start)
        y1 = blk(x0)
        s1 = {k: v.clone() for k, v in blk.state_dict().items()}
        blk.load_state_dict(start)
        ops.FUSE_BN_INTO_CONV3X3 = False
        try:
            y2 = blk(x0)
        finally:
            ops.FUSE_BN_INTO_CONV3X3 = True
        assert torch.equal(y1, y2)
        for k, v in blk.state_dict().items():
            assert torch.equal(v, s1[k]), k


def test_xcorr_backward_golden(ops, golden):
    g = golden("g12_xcorr")
    for tag in ("ocean255", "cfg5"):
        B, C, hx, hz = [int(v) for v in g["dims_" + tag]]
        shp_x = [(hx - 2, hx - 2), (hx - 4, hx - 2), (hx - 2, hx - 4)]
        shp_z = [(hz - 2, hz - 2), (hz - 4, hz - 2), (hz - 2, hz - 4)]
        xs = [G(det_input("x%d%s" % (i, tag), (B, C) + s, 71, -1, 1)).requires_grad_() for i, s in enumerate(shp_x)]
        zs = [G(det_input("z%d%s" % (i, tag), (B, C) + s, 71, -1, 1)).requires_grad_() for i, s in enumerate(shp_z)]
        w = G(np.asarray(g["weight"], np.float32)).requires_grad_()
        s = ops.groupdw(zs, xs, w)
        up = G(det_input("g" + tag, tuple(s.shape), 71, -1, 1))
        (s * up).sum().backward()
        for i in range(3):
            np.testing.assert_allclose(xs[i].grad.cpu().numpy(), g["dx%d_%s" % (i, tag)], rtol=1e-4, atol=1e-5)
            np.testing.assert_allclose(zs[i].grad.cpu().numpy(), g["dz%d_%s" % (i, tag)], rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(w.grad.cpu().numpy(), g["dw_" + tag], rtol=1e-3, atol=1e-3)
    # single correlation, odd shapes, vs torch autograd on the CPU
    for (B, C, hx, wx, hz, wz) in ((1, 2, 9, 7, 3, 2), (2, 3, 20, 31, 7, 7), (1, 1, 8, 8, 8, 8)):
        x, z = det_input("x", (B, C, hx, wx), 8, -1, 1), det_input("z", (B, C, hz, wz), 8, -1, 1)
        xg, zg = G(x).requires_grad_(), G(z).requires_grad_()
        out = ops.xcorr_depthwise(xg, zg)
        up = det_input("up", tuple(out.shape), 8, -1, 1)
        (out * G(up)).sum().backward()
        xc, zc = T(x).double().requires_grad_(), T(z).double().requires_grad_()
        oc = torch.nn.functional.conv2d(xc.view(1, B * C, hx, wx), zc.view(B * C, 1, hz, wz), groups=B * C)
        (oc.view(B, C, *oc.shape[2:]) * T(up).double()).sum().backward()
        np.testing.assert_allclose(xg.grad.cpu().numpy(), xc.grad.numpy(), rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(zg.grad.cpu().numpy(), zc.grad.numpy(), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("N,C,H,W", [(3, 64, 16, 16), (5, 8, 6, 6), (2, 2048, 4, 4), (40, 16, 64, 64)])
def test_batch_norm_act_fwd_bwd_vs_torch(ops, N, C, H, W):
    """Fused BatchNorm (+ residual) (+ ReLU) in training and inference mode against nn.BatchNorm2d / F.relu in fp64."""
    x, r = det_input("x", (N, C, H, W), 3, -2, 3), det_input("r", (N, C, H, W), 4, -1, 1)
    gm, bt = det_input("g", (C,), 5, 0.5, 1.5), det_input("b", (C,), 6, -0.5, 0.5)
    up = det_input("up", (N, C, H, W), 7, -1, 1)
    for relu, use_res in ((True, False), (True, True), (False, False)):
        bn = torch.nn.BatchNorm2d(C).to(DEV)
        ref = torch.nn.BatchNorm2d(C).double()
        with torch.no_grad():
            for m in (bn, ref):
                m.weight.copy_(T(gm))
                m.bias.copy_(T(bt))
                m.running_mean.fill_(0.25)
                m.running_var.fill_(2.0)
        xg, rg = G(x).requires_grad_(), G(r).requires_grad_()
        y = ops.batch_norm_act(xg, bn, relu=relu, residual=rg if use_res else None)
        (y * G(up)).sum().backward()
        xc, rc = T(x).double().requires_grad_(), T(r).double().requires_grad_()
        yc = ref(xc) + (rc if use_res else 0)
        yc = torch.relu(yc) if relu else yc
        (yc * T(up).double()).sum().backward()
        np.testing.assert_allclose(y.detach().cpu().numpy(), yc.detach().numpy(), rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(xg.grad.cpu().numpy(), xc.grad.numpy(), rtol=1e-3, atol=2e-5)
        if use_res:
            np.testing.assert_allclose(rg.grad.cpu().numpy(), rc.grad.numpy(), rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(bn.weight.grad.cpu().numpy(), ref.weight.grad.numpy(), rtol=1e-3, atol=1e-3)
        np.testing.assert_allclose(bn.bias.grad.cpu().numpy(), ref.bias.grad.numpy(), rtol=1e-3, atol=1e-3)
        np.testing.assert_allclose(bn.running_mean.cpu().numpy(), ref.running_mean.numpy(), rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(bn.running_var.cpu().numpy(), ref.running_var.numpy(), rtol=1e-5, atol=1e-6)
        assert int(bn.num_batches_tracked) == 1
        bn.eval(), ref.eval()
        with torch.no_grad():
            ye = ops.batch_norm_act(G(x), bn, relu=relu, residual=G(r) if use_res else None)
            yr = ref(T(x).double()) + (T(r).double() if use_res else 0)
            yr = torch.relu(yr) if relu else yr
        np.testing.assert_allclose(ye.cpu().numpy(), yr.numpy(), rtol=1e-5, atol=1e-5)


def test_flat_adamw_matches_torch_adamw(ops):
    """One-launch AdamW over the flat buffer vs torch.optim.AdamW: two groups with different learning rates,
    odd-sized tensors, buffers in between that must stay untouched, three steps."""
    from spot_amd.arena import ParamArena
    from spot_amd.optim import FlatAdamW

    def make():
        torch.manual_seed(5)
        return torch.nn.Sequential(torch.nn.Conv2d(3, 7, 3), torch.nn.BatchNorm2d(7), torch.nn.Linear(5, 9),
                                   torch.nn.Linear(9, 4099)).to(DEV)

    a, b = make(), make()
    arena = ParamArena(a, with_grads=True)
    groups = lambda m: [{"params": [p for n, p in m.named_parameters() if not n.startswith("0.")]},
                        {"params": [p for n, p in m.named_parameters() if n.startswith("0.")], "lr": 3e-4}]
    oa = FlatAdamW(groups(a), arena, lr=2e-3, weight_decay=1e-2)
    ob = torch.optim.AdamW(groups(b), lr=2e-3, weight_decay=1e-2)
    rm0 = a[1].running_mean.clone()
    for step in range(3):
        for (n, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
            g = G(det_input(n, tuple(pa.shape), 10 + step, -1, 1))
            pa.grad.copy_(g)
            pb.grad = g.clone()
        oa.step()
        ob.step()
    for (n, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
        np.testing.assert_allclose(pa.detach().cpu().numpy(), pb.detach().cpu().numpy(), rtol=2e-6, atol=1e-7, err_msg=n)
    assert torch.equal(a[1].running_mean, rm0)
    oa.param_groups[0]["lr"] = 0.0            # schedulers act through param_groups
    oa.param_groups[1]["lr"] = 0.0
    oa.param_groups[0]["weight_decay"] = 0.0
    oa.param_groups[1]["weight_decay"] = 0.0
    before = arena.flat.clone()
    oa.step()
    assert torch.equal(before, arena.flat)


def test_flat_adamw_exchanges_checkpoints_with_torch_adamw(ops):
    """A torch.optim.AdamW state dict (what the reference's checkpoints hold) loads into FlatAdamW with its moments and
    step, the caller's dict is left alone, the next step matches torch's, and the state dict FlatAdamW writes loads back
    into torch.optim.AdamW.  Frozen parameters and diverging per-parameter steps are refused."""
    import copy
    from spot_amd.arena import ParamArena
    from spot_amd.optim import FlatAdamW

    def make():
        torch.manual_seed(6)
        return torch.nn.Sequential(torch.nn.Linear(5, 9), torch.nn.BatchNorm1d(9), torch.nn.Linear(9, 301)).to(DEV)

    def grads(models, step):
        for m in models:
            for n, p in m.named_parameters():
                g = G(det_input(n, tuple(p.shape), 30 + step, -1, 1))
                if p.grad is None:
                    p.grad = g.clone()
                else:
                    p.grad.copy_(g)

    a, b = make(), make()
    ob = torch.optim.AdamW(b.parameters(), lr=2e-3, weight_decay=1e-2)
    for step in range(2):
        grads([b], step)
        ob.step()
    a.load_state_dict(b.state_dict())
    arena = ParamArena(a, with_grads=True)
    oa = FlatAdamW(a.parameters(), arena, lr=2e-3, weight_decay=1e-2)
    ckpt = ob.state_dict()
    keys_before = (set(ckpt), set(ckpt["state"]))
    oa.load_state_dict(ckpt)
    assert (set(ckpt), set(ckpt["state"])) == keys_before
    assert oa._step == 2
    grads([a, b], 2)
    oa.step(), ob.step()
    for (n, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
        np.testing.assert_allclose(pa.detach().cpu().numpy(), pb.detach().cpu().numpy(), rtol=2e-6, atol=1e-7, err_msg=n)
    # and back: torch's optimizer continues from FlatAdamW's state
    c = make()
    c.load_state_dict(a.state_dict())
    oc = torch.optim.AdamW(c.parameters(), lr=2e-3, weight_decay=1e-2)
    oc.load_state_dict(copy.deepcopy(oa.state_dict()))
    grads([a, c], 3)
    oa.step(), oc.step()
    for (n, pa), (_, pc) in zip(a.named_parameters(), c.named_parameters()):
        np.testing.assert_allclose(pa.detach().cpu().numpy(), pc.detach().cpu().numpy(), rtol=2e-6, atol=1e-7, err_msg=n)
    bad = copy.deepcopy(ob.state_dict())
    bad["state"][0]["step"] = torch.tensor(7.0)
    with pytest.raises(ValueError, match="step counts differ"):
        oa.load_state_dict(bad)
    a[0].bias.requires_grad_(False)
    with pytest.raises(ValueError, match="requires_grad=False"):
        FlatAdamW(a.parameters(), arena, lr=1e-3)


@pytest.mark.parametrize("B,Ci,Co,H,W,stride", [(3, 64, 256, 16, 16, 1), (2, 256, 64, 8, 8, 1), (2, 96, 40, 6, 6, 1),
                                                (2, 128, 256, 16, 16, 2)])
def test_conv1x1_as_gemm_vs_torch(ops, B, Ci, Co, H, W, stride):
    conv = torch.nn.Conv2d(Ci, Co, 1, stride=stride, bias=False).to(DEV)
    x = det_input("x", (B, Ci, H, W), 3, -1, 1)
    with torch.no_grad():
        conv.weight.copy_(G(det_input("w", (Co, Ci, 1, 1), 4, -0.2, 0.2)))
    xg = G(x).requires_grad_()
    y = ops.conv1x1(xg, conv)
    up = det_input("up", tuple(y.shape), 5, -1, 1)
    (y * G(up)).sum().backward()
    xc = T(x).double().requires_grad_()
    wc = conv.weight.detach().cpu().double().requires_grad_()
    yc = torch.nn.functional.conv2d(xc, wc, None, stride=stride)
    (yc * T(up).double()).sum().backward()
    np.testing.assert_allclose(y.detach().cpu().numpy(), yc.detach().numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(xg.grad.cpu().numpy(), xc.grad.numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(conv.weight.grad.cpu().numpy(), wc.grad.numpy(), rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("B,Ci,Co,H,W,dil", [(2, 8, 12, 8, 12, 1), (3, 64, 64, 16, 16, 1), (1, 5, 3, 4, 4, 1),
                                             (2, 128, 96, 32, 32, 1), (2, 8, 12, 8, 16, 2), (3, 64, 48, 32, 32, 2)])
def test_conv3x3_winograd_vs_direct(ops, B, Ci, Co, H, W, dil):
    """Forward, input gradient (Winograd) and weight gradient against torch's direct convolution in float64."""
    import torch.nn.functional as F
    x = det_input("x", (B, Ci, H, W), 1, -1, 1)
    w = det_input("w", (Co, Ci, 3, 3), 2, -0.2, 0.2)
    dy = det_input("dy", (B, Co, H, W), 3, -1, 1)
    conv = torch.nn.Conv2d(Ci, Co, 3, padding=dil, dilation=dil, bias=False).to(DEV)
    with torch.no_grad():
        conv.weight.copy_(G(w))
    xg = G(x).requires_grad_()
    y = ops.conv3x3(xg, conv)
    assert type(y.grad_fn).__name__.startswith("Conv3x3Fn")
    y.backward(G(dy))
    xr = T(x).double().requires_grad_()
    wr = T(w).double().requires_grad_()
    yr = F.conv2d(xr, wr, None, 1, dil, dil)
    yr.backward(T(dy).double())
    scale = float(yr.detach().abs().max())
    np.testing.assert_allclose(y.detach().cpu().double().numpy(), yr.detach().numpy(), rtol=0, atol=2e-5 * scale)
    np.testing.assert_allclose(xg.grad.cpu().double().numpy(), xr.grad.numpy(), rtol=0,
                               atol=2e-5 * float(xr.grad.abs().max()))
    np.testing.assert_allclose(conv.weight.grad.cpu().double().numpy(), wr.grad.numpy(), rtol=0,
                               atol=1e-4 * float(wr.grad.abs().max()))
    # strided / biased convolutions stay with the convolution library
    other = torch.nn.Conv2d(Ci, Co, 3, padding=1, stride=2, bias=False).to(DEV)
    assert not type(ops.conv3x3(xg, other).grad_fn).__name__.startswith("Conv3x3Fn")


@pytest.mark.parametrize("stride,with_res", [(1, False), (1, True), (2, False)])
def test_conv1x1_with_folded_inference_batchnorm(ops, stride, with_res):
    """eval()-mode BatchNorm behind a 1x1 convolution rides in the GEMM epilogue (with the residual add and the ReLU
    after it): same result as convolution -> nn.BatchNorm2d.eval() -> (+ residual) -> relu in float64."""
    B, Ci, Co, H, W = 3, 48, 80, 16, 24
    x = det_input("x", (B, Ci, H, W), 1, -1, 1)
    conv = torch.nn.Conv2d(Ci, Co, 1, stride=stride, bias=False).to(DEV)
    bn = torch.nn.BatchNorm2d(Co).to(DEV).eval()
    with torch.no_grad():
        conv.weight.copy_(G(det_input("w", (Co, Ci, 1, 1), 2, -0.3, 0.3)))
        bn.weight.copy_(G(det_input("g", (Co,), 3, 0.5, 1.5)))
        bn.bias.copy_(G(det_input("b", (Co,), 4, -0.5, 0.5)))
        bn.running_mean.copy_(G(det_input("m", (Co,), 5, -0.5, 0.5)))
        bn.running_var.copy_(G(det_input("v", (Co,), 6, 0.5, 2.0)))
    res = det_input("r", (B, Co, H // stride, W // stride), 7, -1, 1) if with_res else None
    with torch.no_grad():
        y = ops.conv1x1_bn_act(G(x), conv, bn, relu=True, residual=G(res) if with_res else None)
        two_pass = ops.batch_norm_act(ops.conv1x1(G(x), conv), bn, relu=True, residual=G(res) if with_res else None)
    ref = torch.nn.functional.conv2d(T(x).double(), conv.weight.detach().cpu().double(), None, stride)
    a = bn.weight.detach().cpu().double() / torch.sqrt(bn.running_var.cpu().double() + bn.eps)
    ref = ref * a.view(1, -1, 1, 1) + (bn.bias.detach().cpu().double() - bn.running_mean.cpu().double() * a).view(1, -1, 1, 1)
    if with_res:
        ref = ref + T(res).double()
    ref = torch.relu(ref)
    np.testing.assert_allclose(y.cpu().double().numpy(), ref.numpy(), rtol=1e-5, atol=1e-5)
    assert torch.equal(y, two_pass)          # the fused epilogue repeats the two-pass arithmetic exactly
    # in training mode (or with gradients wanted) the same call is convolution -> batch-statistics BatchNorm
    bn.train()
    yt = ops.conv1x1_bn_act(G(x).requires_grad_(), conv, bn, relu=True, residual=G(res) if with_res else None)
    assert yt.requires_grad


@pytest.mark.parametrize("B,C,H,W", [(2, 3, 8, 16), (3, 64, 128, 128), (1, 5, 6, 8), (2, 16, 61, 61), (1, 3, 7, 12),
                                     (2, 2, 1, 5), (1, 4, 10, 9)])
def test_maxpool3x3s2_vs_torch(ops, B, C, H, W):
    """Stem max-pool: values, tie-breaking and gradient equal nn.MaxPool2d(3, 2, 1) exactly, on the vectorised kernel's
    shapes and on odd / small maps (the Siamese variant's 61x61)."""
    pool = torch.nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
    x = det_input("x", (B, C, H, W), 1, -1, 1)
    xt = T(x).clone()
    xt[:, :, ::3, ::2] = 0.25                      # plateaus: ties inside many windows
    xt[0, 0, :4, :8] = -3.0                        # a constant block at the padded corner
    dy = det_input("dy", (B, C, (H - 1) // 2 + 1, (W - 1) // 2 + 1), 2, -1, 1)
    xg = xt.to(DEV).requires_grad_()
    y = ops.max_pool_3x3_s2(xg, pool)
    assert type(y.grad_fn).__name__.startswith("MaxPool3x3s2Fn")
    y.backward(G(dy))
    xr = xt.clone().requires_grad_()
    yr = pool(xr)
    yr.backward(T(dy))
    assert torch.equal(y.detach().cpu(), yr.detach())
    np.testing.assert_allclose(xg.grad.cpu().numpy(), xr.grad.numpy(), rtol=0, atol=1e-6)   # <= 4 addends, order free
    with torch.no_grad():
        assert torch.equal(ops.max_pool_3x3_s2(xt.to(DEV), pool).cpu(), yr.detach())


@pytest.mark.parametrize("H,W", [(16, 16), (9, 11)])
def test_maxpool3x3s2_propagates_nan_as_torch(ops, H, W):
    """A NaN anywhere in a window reaches the output (the trainer's skip logic relies on non-finite values arriving at
    the loss), and the gradient goes where nn.MaxPool2d sends it; both the vectorised and the any-size kernel."""
    pool = torch.nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
    xt = T(det_input("x", (2, 3, H, W), 3, -1, 1)).clone()
    xt[0, 0, 4, 5] = float("nan")
    xt[0, 1, 0, 0] = float("nan")
    xt[1, 2, 3, 3] = float("nan")
    xt[1, 2, 3, 4] = float("nan")                   # two in one window: the later one is the recorded winner
    xt[1, 0, H - 1, W - 1] = float("inf")
    dy = det_input("dy", (2, 3, (H - 1) // 2 + 1, (W - 1) // 2 + 1), 4, -1, 1)
    xg = xt.to(DEV).requires_grad_()
    y = ops.max_pool_3x3_s2(xg, pool)
    y.backward(G(dy))
    xr = xt.clone().requires_grad_()
    yr = pool(xr)
    yr.backward(T(dy))
    assert torch.isnan(yr).sum() > 3
    assert torch.equal(torch.isnan(y).cpu(), torch.isnan(yr))
    assert torch.equal(torch.nan_to_num(y.detach(), nan=7.0).cpu(), torch.nan_to_num(yr.detach(), nan=7.0))
    np.testing.assert_allclose(xg.grad.cpu().numpy(), xr.grad.numpy(), rtol=0, atol=1e-6)


def test_in_network_transforms_golden(golden):
    """Batched GrayScale / EllipseRotationAffine kernels against the reference's per-sample loop (golden draws)."""
    from spot_amd.transforms_in_net import EllipseRotationAffine, GrayScale
    g = golden("g16_transforms")
    seed = int(g["seed"])
    B, C, H, W = 8, 3, 32, 48
    img, tmpl = det_input("img", (B, C, H, W), seed, -2, 2), det_input("tmpl", (B, C, 16, 24), seed + 1, -2, 2)
    avg, avg_t = det_input("avg", (B, 3), seed, -1, 1), det_input("avg_t", (B, 3), seed + 1, -1, 1)
    era = EllipseRotationAffine(max_angle=10, lr_flip_prob=0.35, ud_flip_prob=0.25)
    im, bb = era.apply(G(img), G(g["bbox"]), G(avg), G(g["mask"]), G(g["angle"]), G(g["lr_flip"]))
    # nearest sampling: a source coordinate within rounding of a pixel boundary may pick the neighbour
    differ = (im.cpu() != T(g["era_img"])).float().mean().item()
    assert differ <= 2e-4, differ
    np.testing.assert_allclose(bb.cpu().numpy(), g["era_bbox"], rtol=0, atol=2e-3)
    unmasked = ~T(g["mask"])
    assert torch.equal(im.cpu()[unmasked], T(img)[unmasked])
    gs = GrayScale(0.4)
    t_dev, i_dev = G(tmpl), G(img)
    groups, avgs = gs.apply([t_dev, i_dev], [G(avg_t), G(avg)], G(g["gray_tag"]))
    assert torch.equal(groups[0].cpu(), T(g["gray_tmpl"])) and torch.equal(groups[1].cpu(), T(g["gray_img"]))
    np.testing.assert_allclose(avgs[0].cpu().numpy(), g["gray_avg_t"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(avgs[1].cpu().numpy(), g["gray_avg"], rtol=0, atol=1e-6)
    # full-size crops, all angles of the training configuration incl. the 180-degree turn: against the oracle
    from oracle import spot_oracle as O
    Bf = 6
    big = det_input("big", (Bf, 3, 256, 256), 5, -2, 2)
    ang = torch.tensor([-10., 7., 180., 173., 0., 190.])
    lr = torch.tensor([False, True, True, False, True, False])
    msk = torch.tensor([True, True, True, True, False, True])
    box = torch.tensor([[100., 90., 60., 40.]] * Bf)
    favg = T(det_input("favg", (Bf, 3), 6, -1, 1))
    im2, bb2 = era.apply(G(big), box.to(DEV), favg.to(DEV), msk.to(DEV), ang.to(DEV), lr.to(DEV))
    ro, rb = O.ellipse_rotation_affine(T(big), box.clone(), favg, msk, ang, lr)
    assert (im2.cpu() != ro).float().mean().item() <= 2e-4
    np.testing.assert_allclose(bb2.cpu().numpy(), rb.numpy(), rtol=0, atol=2e-3)


@pytest.mark.parametrize("B,Ci,Co,H,W,k,s,p,d", [(3, 3, 64, 32, 32, 7, 2, 3, 1), (2, 32, 48, 16, 16, 3, 2, 1, 1),
                                                 (2, 8, 16, 16, 32, 3, 1, 1, 1), (1, 3, 64, 64, 64, 7, 2, 3, 1)])
def test_conv_im2col_vs_direct(ops, B, Ci, Co, H, W, k, s, p, d):
    """Patch matrix + GEMM convolution (stem 7x7 / stride 2, strided 3x3): forward and both gradients against
    torch's direct convolution in float64."""
    import torch.nn.functional as F
    x = det_input("x", (B, Ci, H, W), 1, -1, 1)
    w = det_input("w", (Co, Ci, k, k), 2, -0.2, 0.2)
    conv = torch.nn.Conv2d(Ci, Co, k, stride=s, padding=p, dilation=d, bias=False).to(DEV)
    with torch.no_grad():
        conv.weight.copy_(G(w))
    xg = G(x).requires_grad_()
    y = ops.conv_general(xg, conv)
    assert type(y.grad_fn).__name__.startswith("ConvIm2colFn")
    dy = det_input("dy", tuple(y.shape), 3, -1, 1)
    y.backward(G(dy))
    xr, wr = T(x).double().requires_grad_(), T(w).double().requires_grad_()
    yr = F.conv2d(xr, wr, None, s, p, d)
    yr.backward(T(dy).double())
    for got, ref in ((y.detach(), yr.detach()), (xg.grad, xr.grad), (conv.weight.grad, wr.grad)):
        np.testing.assert_allclose(got.cpu().double().numpy(), ref.numpy(), rtol=0, atol=2e-5 * float(ref.abs().max()))


@pytest.mark.parametrize("dil", [1, 2])
def test_conv3x3_with_folded_inference_batchnorm_and_kept_weights(ops, dil):
    """eval()-mode BatchNorm + ReLU in the Winograd output transform == convolution -> inference BatchNorm kernel,
    and with frozen weights the transformed weights are reused across calls (and dropped by clear())."""
    B, Ci, Co, H, W = 2, 32, 48, 16, 16
    x = G(det_input("x", (B, Ci, H, W), 1, -1, 1))
    conv = torch.nn.Conv2d(Ci, Co, 3, padding=dil, dilation=dil, bias=False).to(DEV)
    bn = torch.nn.BatchNorm2d(Co).to(DEV).eval()
    with torch.no_grad():
        conv.weight.copy_(G(det_input("w", (Co, Ci, 3, 3), 2, -0.2, 0.2)))
        bn.weight.copy_(G(det_input("g", (Co,), 3, 0.5, 1.5)))
        bn.bias.copy_(G(det_input("b", (Co,), 4, -0.5, 0.5)))
        bn.running_mean.copy_(G(det_input("m", (Co,), 5, -0.5, 0.5)))
        bn.running_var.copy_(G(det_input("v", (Co,), 6, 0.5, 2.0)))
        two_pass = ops.batch_norm_act(ops.conv3x3(x, conv), bn, relu=True)
        fused = ops.conv3x3_bn_act(x, conv, bn, relu=True)
        assert torch.equal(fused, two_pass)
        ops.frozen_weights.clear()
        with ops.frozen_weights():
            first = ops.conv3x3_bn_act(x, conv, bn, relu=True)
            assert len(ops.frozen_weights._tables) == 1
            again = ops.conv3x3_bn_act(x, conv, bn, relu=True)          # transformed weights not recomputed
            conv.weight.mul_(2.0)                                       # a weight update inside the context is NOT seen
            stale = ops.conv3x3_bn_act(x, conv, bn, relu=True)
        assert torch.equal(first, two_pass) and torch.equal(again, two_pass) and torch.equal(stale, two_pass)
        ops.frozen_weights.clear()
        with ops.frozen_weights():
            fresh = ops.conv3x3_bn_act(x, conv, bn, relu=True)
        assert not torch.equal(fresh, two_pass)
        ops.frozen_weights.clear()



@pytest.mark.parametrize("Lq,Lk,nsplit", [(256, 256, 4), (1024, 1024, 8), (256, 1024, 5), (50, 77, 2), (50, 77, 8), (16, 64, 3)])
def test_mha_forward_with_split_keys(ops, Lq, Lk, nsplit):
    """The inference forward for small grids: key ranges handled by separate workgroups and merged; ragged sizes, a
    padding mask that empties one whole range, more ranges than key tiles."""
    from spot_amd import lib as L
    B, nh, dh = 1, 4, 32
    C = nh * dh
    q, k, v = (det_input(n, (B, L_, C), 3, -2, 2) for n, L_ in (("q", Lq), ("k", Lk), ("v", Lk)))
    kpm = np.zeros((B, Lk), bool)
    kpm[0, :min(64, Lk - 1)] = True              # the first 64-key tile is masked out entirely
    qg, kg, vg, kp = G(q).view(-1, C), G(k).view(-1, C), G(v).view(-1, C), G(kpm.astype(np.uint8))
    ref = _ref_attn(T(q).double(), T(k).double(), T(v).double(), T(kpm), nh)
    out = torch.empty((B * Lq, C), device=DEV)
    ws = torch.empty((L.load().spot_mha_fwd_splitk_ws_floats(B, Lq, nh, nsplit),), device=DEV)
    P = ops._p
    L.call("spot_mha_fwd_splitk_f32", P(qg), C, P(kg), C, P(vg), C, P(out), C, P(kp), P(ws), nsplit, B, Lq, Lk, nh, dh,
           1.0 / math.sqrt(dh), ops._st())
    np.testing.assert_allclose(out.view(B, Lq, C).cpu().numpy(), ref.numpy(), rtol=1e-4, atol=1e-5)
    one, _ = ops.mha_core_fwd(qg, C, kg, C, vg, C, B, Lq, Lk, nh, dh, kp, True)       # single-range kernel
    np.testing.assert_allclose(out.cpu().numpy(), one.cpu().numpy(), rtol=1e-5, atol=1e-6)
    auto, lse = ops.mha_core_fwd(qg, C, kg, C, vg, C, B, Lq, Lk, nh, dh, kp, False)   # the wrapper's own choice
    assert lse is None
    np.testing.assert_allclose(auto.cpu().numpy(), one.cpu().numpy(), rtol=1e-5, atol=1e-6)


def test_two_problem_entry_points_equal_separate_calls(ops):
    """The two-problem launches of one-clip inference (GEMM with 32x32 k-split tiles, LayerNorm, split-key attention +
    merge) give, bit for bit, what the single-problem entry points give for either problem; a pair the library cannot
    put in one launch (different tiles) still runs both."""
    from spot_amd import lib as L
    P = ops._p
    # --- GEMM pair: x W^T + bias + residual, template rows (256) beside search rows (1024)
    outs = {}
    for tag, paired in (("single", False), ("pair", True)):
        xa, wa, ba, ra = (G(det_input(n, sh, 5, -1, 1)) for n, sh in (("xa", (256, 256)), ("wa", (256, 256)), ("ba", (256,)), ("ra", (256, 256))))
        xb, wb, bb, rb = (G(det_input(n, sh, 6, -1, 1)) for n, sh in (("xb", (1024, 256)), ("wb", (512, 256)), ("bb", (512,)), ("rb", (1024, 512))))
        ya, yb = torch.empty(256, 256, device=DEV), torch.empty(1024, 512, device=DEV)
        if paired:
            with L.record() as r0:
                ops.gemm(xa, wa, ya, 256, 256, 256, bias=ba, residual=ra, ldres=256, tile=32)
            with L.record() as r1:
                ops.gemm(xb, wb, yb, 1024, 512, 256, bias=bb, residual=rb, ldres=512, relu=True, tile=32)
            assert len(r0.calls) == len(r1.calls) == 1
            L.run_paired(r0.calls, r1.calls)
        else:
            ops.gemm(xa, wa, ya, 256, 256, 256, bias=ba, residual=ra, ldres=256, tile=32)
            ops.gemm(xb, wb, yb, 1024, 512, 256, bias=bb, residual=rb, ldres=512, relu=True, tile=32)
        outs[tag] = (ya.clone(), yb.clone())
    assert torch.equal(outs["single"][0], outs["pair"][0]) and torch.equal(outs["single"][1], outs["pair"][1])
    ref = T(det_input("xa", (256, 256), 5, -1, 1)).double() @ T(det_input("wa", (256, 256), 5, -1, 1)).double().T
    ref = ref + T(det_input("ba", (256,), 5, -1, 1)).double() + T(det_input("ra", (256, 256), 5, -1, 1)).double()
    np.testing.assert_allclose(outs["pair"][0].cpu().double().numpy(), ref.numpy(), rtol=0, atol=2e-4)
    # a pair of different tiles falls back to two launches
    xa, wa = G(det_input("xa", (256, 256), 5, -1, 1)), G(det_input("wa", (256, 256), 5, -1, 1))
    y32, y64, y1, y2 = (torch.empty(256, 256, device=DEV) for _ in range(4))
    with L.record() as r0:
        ops.gemm(xa, wa, y32, 256, 256, 256, tile=32)
    with L.record() as r1:
        ops.gemm(xa, wa, y64, 256, 256, 256, tile=64)
    L.run_paired(r0.calls, r1.calls)
    ops.gemm(xa, wa, y1, 256, 256, 256, tile=32)
    ops.gemm(xa, wa, y2, 256, 256, 256, tile=64)
    assert torch.equal(y32, y1) and torch.equal(y64, y2)
    # --- LayerNorm pair
    xa, xb = G(det_input("lna", (256, 256), 7, -3, 3)), G(det_input("lnb", (1021, 256), 8, -3, 3))
    ga, gb, ba, bb = (G(det_input(n, (256,), 9, 0.5, 1.5)) for n in ("ga", "gb", "ba", "bb"))
    with torch.no_grad():
        sa, sb = ops.layer_norm(xa, ga, ba, 1e-5), ops.layer_norm(xb, gb, bb, 1e-5)
        with L.record() as r0:
            pa = ops.layer_norm(xa, ga, ba, 1e-5)
        with L.record() as r1:
            pb = ops.layer_norm(xb, gb, bb, 1e-5)
        L.run_paired(r0.calls, r1.calls)
    assert torch.equal(sa, pa) and torch.equal(sb, pb)
    # --- split-key attention pair (one clip each)
    nh, dh = 8, 32
    C = nh * dh
    res = {}
    for tag in ("single", "pair"):
        got = []
        recs = []
        for (Lq, Lk, ns, seed) in ((256, 1024, 4, 11), (1024, 256, 3, 12)):
            q, k, v = (G(det_input(n, (L_, C), seed, -2, 2)) for n, L_ in (("q", Lq), ("k", Lk), ("v", Lk)))
            out = torch.empty((Lq, C), device=DEV)
            ws = torch.empty((L.load().spot_mha_fwd_splitk_ws_floats(1, Lq, nh, ns),), device=DEV)
            args = (P(q), C, P(k), C, P(v), C, P(out), C, None, P(ws), ns, 1, Lq, Lk, nh, dh, 1.0 / math.sqrt(dh), ops._st())
            if tag == "single":
                L.call("spot_mha_fwd_splitk_f32", *args)
            else:
                recs.append([("spot_mha_fwd_splitk_f32", args)])
            got.append((out, q, k, v, ws))
        if tag == "pair":
            L.run_paired(recs[0], recs[1])
        res[tag] = [g[0].clone() for g in got]
    assert torch.equal(res["single"][0], res["pair"][0]) and torch.equal(res["single"][1], res["pair"][1])


def test_collected_launches_keep_their_buffers(ops):
    """Launches collected for pairing address buffers that the collecting code has already let go of: the collection
    must keep them, or two collected launches end up on the same memory."""
    from spot_amd import lib as L
    x = G(det_input("x", (512, 256), 3, -1, 1))
    w1, w2 = G(det_input("w1", (256, 256), 4, -0.2, 0.2)), G(det_input("w2", (256, 256), 5, -0.2, 0.2))

    def chain(w):                               # two dependent products; the intermediate is released on return
        h = ops.linear_fwd(x, w)
        return ops.linear_fwd(h, w)
    with torch.no_grad():
        ref1, ref2 = chain(w1), chain(w2)
        with L.record() as r0:
            o1 = chain(w1)
        with L.record() as r1:
            o2 = chain(w2)
        assert len(r0.tensors) > 0
        L.run_paired(r0.calls, r1.calls)
    assert torch.equal(o1, ref1) and torch.equal(o2, ref2)


# --------------------------------------------------------------------------- position encoding, matcher, criterion
def test_position_encoding_golden(golden):
    """The sine position encoding kernel against the reference's PositionEmbeddingSine: the unmasked 16x16 and 32x32
    maps of the template / search tokens (as the cached token rows the network reads) and a masked 8x8 map."""
    from spot_amd.backbone import PositionEmbeddingSine
    from spot_amd.tensordict import NestedTensor
    g = golden("g14_pre_pos")
    pe = PositionEmbeddingSine(128, normalize=True)
    for hw in (16, 32):
        rows = pe.rows(hw, hw, torch.device(DEV))                        # [H*W, 256]
        ref = g["pos%d" % hw][0].reshape(256, hw * hw).T
        np.testing.assert_allclose(rows.cpu().numpy(), ref, rtol=1e-5, atol=2e-6)
        assert pe.rows(hw, hw, torch.device(DEV)) is rows                # kept per (H, W, device)
    mask = G(g["mask8"])
    out = pe(NestedTensor(torch.zeros(1, 1, 8, 8, device=DEV), mask))
    np.testing.assert_allclose(out.cpu().numpy(), g["pos8_masked"], rtol=1e-5, atol=2e-6)
    both = pe(NestedTensor(None, torch.cat([mask, torch.zeros_like(mask)])))     # clips do not interact
    assert torch.equal(both[0], out[0])
    np.testing.assert_allclose(both[1].cpu().numpy(), O.position_embedding_sine(torch.zeros(1, 8, 8, dtype=torch.bool))[0].numpy(),
                               rtol=1e-5, atol=2e-6)


@pytest.mark.parametrize("name,seed", [("g9_matcher_criterion", 51), ("g22_matcher_offmap", 53)])
def test_matcher_masks_and_criterion_golden(golden, name, seed):
    """spot_match_masks on the reference's border cases (boxes at the map's edge, beyond it on every side, empty cell
    ranges, the whole map): token sets identical to the reference matcher's index lists; the losses built on them and
    their gradients against the reference criterion."""
    from spot_amd.criterion import transt_loss
    from spot_amd.synthetic import Settings
    g = golden(name)
    gt = g["gt"]
    B, L = gt.shape[0], 1024
    crit = transt_loss(Settings()).to(DEV)
    logits, boxes = det_input("logits", (B, L, 2), seed, -2, 2), det_input("boxes", (B, L, 4), seed, 0.05, 0.6)
    ious = det_input("ious", (B, L, 1), seed, 0.0, 1.0)
    wd = Settings().weight_dict
    for tag in ("sup", "unsup"):
        o = {"logits_" + tag: G(logits).requires_grad_(), "boxes_" + tag: G(boxes).requires_grad_()}
        if tag == "sup":
            o["ious_" + tag] = G(ious).requires_grad_()
        pos, neg = crit.matcher(o, G(gt), tag=tag)
        assert np.array_equal(pos.cpu().numpy(), g["pos_mask_" + tag])
        if tag == "sup":
            assert np.array_equal(neg.cpu().numpy().astype(np.int32), g["neg_count_" + tag])
        else:
            assert neg is None
        losses = crit(o, G(gt), tag=tag)
        for k, v in losses.items():
            np.testing.assert_allclose(v.item(), g["%s_%s" % (k, tag)], rtol=2e-5, err_msg=k)
        total = sum(losses[k] * wd[k + "_" + tag] for k in losses if k + "_" + tag in wd)
        np.testing.assert_allclose(total.item(), g["total_" + tag], rtol=2e-5)
        total.backward()
        for k, v in o.items():
            ref = g["grad_" + k]
            np.testing.assert_allclose(v.grad.cpu().numpy(), ref, rtol=1e-4, atol=1e-6 * float(np.abs(ref).max()), err_msg=k)


# --------------------------------------------------------------------------- set criterion, heads, small reductions
@pytest.mark.parametrize("tag", ["sup", "unsup"])
def test_set_criterion_kernels_against_oracle(ops, tag):
    """spot_transt_loss_f32 / _bwd_f32 against the oracle's criterion (autograd on the CPU): every loss term and the
    gradients of a weighted total, with zero-size boxes on tokens outside both sets (their GIoU is NaN and must not leak)
    and a prediction that equals its target exactly (sign(0) = 0)."""
    from spot_amd.criterion import transt_loss
    from spot_amd.synthetic import Settings
    B, L = 6, 1024
    rng = np.random.default_rng(5)
    logits = rng.uniform(-3, 3, (B, L, 2)).astype(np.float32)
    boxes = rng.uniform(0.05, 0.7, (B, L, 4)).astype(np.float32)
    ious = rng.uniform(0, 1, (B, L, 1)).astype(np.float32)
    gt = np.stack([rng.uniform(0.3, 0.7, B), rng.uniform(0.3, 0.7, B), rng.uniform(0.05, 0.5, B),
                   rng.uniform(0.05, 0.5, B)], 1).astype(np.float32)
    pos_o, neg_o = O.matcher_masks(gt, L, with_neg=True)
    outside = ~(pos_o | (neg_o > 0))
    boxes[outside, 2:] = 0.0
    boxes[0, np.flatnonzero(pos_o[0])[0]] = gt[0]                       # |box - target| = 0 on one positive token
    wd = Settings().weight_dict
    ref_in = [T(a).requires_grad_() for a in (logits, boxes, ious)]
    ref = O.criterion(ref_in[0], ref_in[1], ref_in[2] if tag == "sup" else None, T(gt), tag=tag)
    sum(ref[k] * wd[k + "_" + tag] for k in ref if k + "_" + tag in wd).backward()
    crit = transt_loss(Settings()).to(DEV)
    o = {"logits_" + tag: G(logits).requires_grad_(), "boxes_" + tag: G(boxes).requires_grad_()}
    if tag == "sup":
        o["ious_" + tag] = G(ious).requires_grad_()
    got = crit(o, G(gt), tag=tag)
    assert set(got) == set(ref)
    for k in ref:
        np.testing.assert_allclose(got[k].item(), ref[k].item(), rtol=1e-5, err_msg=k)
    assert not got["iou"].requires_grad
    sum(got[k] * wd[k + "_" + tag] for k in got if k + "_" + tag in wd).backward()
    for name, r in zip(("logits_", "boxes_", "ious_"), ref_in):
        if name + tag not in o:
            continue
        want = r.grad.numpy()
        np.testing.assert_allclose(o[name + tag].grad.cpu().numpy(), want, rtol=1e-4, atol=1e-6 * float(np.abs(want).max()),
                                   err_msg=name)
    again = crit({k: v.detach() for k, v in o.items()}, G(gt), tag=tag)
    assert all(torch.equal(again[k], got[k]) for k in got)               # fixed reduction order: the same bits


def test_mlp_head_against_torch(ops):
    """Linear-ReLU-Linear-ReLU-Linear through MLPFn (ReLU in the GEMM epilogues, its derivative as the mask operand of the
    input-gradient GEMMs) against torch on the CPU, with pre-activations that hit zero exactly."""
    torch.manual_seed(3)
    lin = [torch.nn.Linear(256, 256), torch.nn.Linear(256, 256), torch.nn.Linear(256, 4)]
    with torch.no_grad():
        lin[0].weight[:7].zero_(), lin[0].bias[:7].zero_()               # seven hidden units sit at exactly 0
    x = torch.randn(3, 130, 256)
    xr = x.clone().requires_grad_()
    h = xr
    for i, l in enumerate(lin):
        h = l(h)
        if i < 2:
            h = torch.relu(h)
    dy = torch.randn_like(h)
    h.backward(dy)
    dev = [torch.nn.Linear(l.in_features, l.out_features).to(DEV) for l in lin]
    for d, l in zip(dev, lin):
        d.load_state_dict(l.state_dict())
    xg = x.to(DEV).requires_grad_()
    y = ops.mlp(xg, dev)
    np.testing.assert_allclose(y.detach().cpu().numpy(), h.detach().numpy(), rtol=1e-4, atol=1e-5)
    y.backward(dy.to(DEV))
    np.testing.assert_allclose(xg.grad.cpu().numpy(), xr.grad.numpy(), rtol=1e-4, atol=1e-5)
    for d, l in zip(dev, lin):
        np.testing.assert_allclose(d.weight.grad.cpu().numpy(), l.weight.grad.numpy(), rtol=1e-4, atol=2e-4)
        np.testing.assert_allclose(d.bias.grad.cpu().numpy(), l.bias.grad.numpy(), rtol=1e-4, atol=2e-4)
    with torch.no_grad():
        assert torch.equal(ops.mlp(xg, dev), y)
    lone = ops.linear(xg.detach().requires_grad_(), dev[0].weight, dev[0].bias)
    np.testing.assert_allclose(lone.detach().cpu().numpy(), lin[0](x).detach().numpy(), rtol=1e-4, atol=1e-5)


def test_batch_channel_and_mixing_weight_reductions(ops):
    rng = np.random.default_rng(9)
    wide = G(rng.standard_normal((5 * 24, 96)).astype(np.float32))
    win = wide[:, 32:64]                                                  # a column window with the wide row stride
    np.testing.assert_allclose(ops.batch_sum(win, 24).cpu().numpy(), win.reshape(5, 24, 32).sum(0).cpu().numpy(),
                               rtol=1e-6, atol=1e-6)
    assert ops.batch_sum(win, 120) is win
    x4 = G(rng.standard_normal((3, 7, 5, 9)).astype(np.float32))
    np.testing.assert_allclose(ops.channel_sum(x4).cpu().numpy(), x4.sum((0, 2, 3)).cpu().numpy(), rtol=1e-5, atol=1e-5)
    # GroupDW mixing logits: gradient through softmax of the per-plane sums
    part, w = rng.standard_normal((300, 3)).astype(np.float32), np.array([0.3, -1.2, 0.8], np.float32)
    wt = T(w).requires_grad_()
    (torch.softmax(wt, 0) * T(part).sum(0)).sum().backward()
    dw, part_d, w_d = torch.empty(3, device=DEV), G(part), G(w)         # named: the launch reads them after this line
    from spot_amd import lib
    lib.call("spot_groupdw_weight_grad_f32", ops._p(part_d), 300, ops._p(w_d), ops._p(dw), ops._st())
    np.testing.assert_allclose(dw.cpu().numpy(), wt.grad.numpy(), rtol=1e-5, atol=1e-6)


def test_gemm_bf16x3_pieces_against_float64(ops):
    """The bf16x3 experiment (spot_gemm_desc.precision = 1): operands cut into three bf16 pieces, six piece products on
    the bf16 matrix instruction.  Its error against float64 must stay within 4x the exact-fp32 kernel's on the same
    operands, for all four operand layouts, both workgroup tiles, split-K, batches and the epilogue terms; shapes it
    cannot take are refused by the library, and passed to the exact kernel when the mode is only a preference."""
    rng = np.random.default_rng(21)
    for ta, tb, M, N, K, tile, sk, batch in ((0, 0, 256, 256, 256, 64, 1, 1), (0, 0, 1024, 768, 256, 128, 1, 1),
                                              (0, 0, 384, 128, 2048, 0, 1, 1), (0, 1, 512, 256, 768, 0, 1, 1),
                                              (0, 1, 256, 384, 128, 128, 1, 1), (1, 1, 256, 128, 4096, 0, 4, 1),
                                              (1, 1, 128, 256, 1024, 64, 1, 1), (1, 0, 256, 256, 512, 0, 1, 3),
                                              (0, 1, 128, 1024, 64, 0, 1, 2)):
        scale = np.exp(rng.uniform(-6, 6, (M, 1)))
        a64 = rng.standard_normal((M, K)) * scale
        A = G((a64.T if ta else a64).astype(np.float32).copy())
        b64 = rng.standard_normal((batch, K, N))
        Bm = G((b64 if tb else b64.transpose(0, 2, 1)).astype(np.float32).copy())
        Ad = (A.T if ta else A).double()
        Bd = (Bm if tb else Bm.transpose(1, 2)).double()                  # [batch, K, N]
        bias, res = G(rng.standard_normal(N).astype(np.float32)), G(rng.standard_normal((batch, M, N)).astype(np.float32))
        plain = Ad @ Bd + bias.double()
        ref = plain if sk > 1 else torch.relu(plain) + res.double()
        err = {}
        for prec in (0, 1):
            out = torch.zeros(batch, M, N, device=DEV)
            kw = dict(transA=bool(ta), transB=bool(tb), bias=bias, tile=tile, precision=prec, batch=batch,
                      strideB=K * N if batch > 1 else 0, strideC=M * N if batch > 1 else 0)
            if sk > 1:
                kw.update(splitk=sk, accumulate=True)
            else:
                kw.update(relu=True, residual=res, ldres=N, strideRes=M * N if batch > 1 else 0)
            ops.gemm(A, Bm, out, M, N, K, **kw)
            err[prec] = float(((out.double() - ref).abs() / (Ad.abs() @ Bd.abs() + 1)).max())
        assert err[1] <= 4 * err[0] + 1e-9, (ta, tb, M, N, K, err)
        assert err[1] < 1e-6, (ta, tb, M, N, K, err)
    A, W, out = G(np.ones((100, 64), np.float32)), G(np.ones((128, 64), np.float32)), torch.empty(100, 128, device=DEV)
    with pytest.raises(RuntimeError, match="bf16x3"):
        ops.gemm(A, W, out, 100, 128, 64, precision=1)
    ops.gemm(A, W, out, 100, 128, 64, precision="try")                     # a preference: the exact kernel takes it
    assert float(out.min()) == 64.0 and float(out.max()) == 64.0


def test_gemm_small_grid_kernels_against_float64(ops):
    """tile = 32 (the 32x32-tile kernel whose four waves split the k step, asked for by the online trackers) and, for
    the larger shapes of the list, the deep-prefetch loop small grids take on their own: all four operand layouts,
    batches, every epilogue term and accumulation against float64; two runs give the same bits (the partial tiles are
    summed in a fixed order); a shape the kernel cannot take is refused."""
    rng = np.random.default_rng(33)
    for ta, tb, M, N, K, batch in ((0, 0, 1024, 256, 256, 1), (0, 0, 256, 256, 2048, 1), (0, 1, 256, 1024, 1024, 1),
                                   (1, 1, 512, 256, 384, 2), (1, 0, 1024, 256, 1024, 1), (0, 1, 64, 4096, 256, 1),
                                   (0, 0, 1024, 1024, 256, 1), (0, 1, 2048, 512, 256, 1)):
        a64 = rng.standard_normal((M, K))
        A = G((a64.T if ta else a64).astype(np.float32).copy())
        b64 = rng.standard_normal((batch, K, N))
        Bm = G((b64 if tb else b64.transpose(0, 2, 1)).astype(np.float32).copy())
        Ad = (A.T if ta else A).double()
        Bd = (Bm if tb else Bm.transpose(1, 2)).double()
        bias = G(rng.standard_normal(N).astype(np.float32))
        rowb = G(rng.standard_normal((128, N)).astype(np.float32))
        res = G(rng.standard_normal((batch, M, N)).astype(np.float32))
        mask = G(rng.standard_normal((M, N)).astype(np.float32))
        start = G(rng.standard_normal((batch, M, N)).astype(np.float32))
        idx = torch.arange(M, device=DEV) % 128
        ref = torch.relu(0.5 * (Ad @ Bd) + bias.double() + rowb.double()[idx])
        ref = torch.where(mask.double() > 0, ref, torch.zeros((), dtype=torch.float64, device=DEV)) + res.double() + start.double()
        outs = []
        for _ in range(2):
            out = start.clone()
            ops.gemm(A, Bm, out, M, N, K, transA=bool(ta), transB=bool(tb), alpha=0.5, bias=bias, rowbias=rowb, rb_rows=128,
                     ldrb=N, relu=True, mask=mask, ldmask=N, residual=res, ldres=N, strideRes=M * N if batch > 1 else 0,
                     batch=batch, strideB=K * N if batch > 1 else 0, strideC=M * N if batch > 1 else 0, accumulate=True,
                     tile=32 if M * N * batch <= 128 * 4096 else 0)
            outs.append(out)
        assert torch.equal(outs[0], outs[1])
        err = float((outs[0].double() - ref).abs().max())
        assert err < 2e-5 * math.sqrt(K), (ta, tb, M, N, K, batch, err)
    # split-K whose last range is empty (9 k tiles over 4 ranges of 3): that workgroup must not touch the operands
    A, W = G(rng.standard_normal((128, 288)).astype(np.float32)), G(rng.standard_normal((128, 288)).astype(np.float32))
    out = torch.zeros(128, 128, device=DEV)
    ops.gemm(A, W, out, 128, 128, 288, splitk=4, accumulate=True)
    np.testing.assert_allclose(out.cpu().double().numpy(), (A.double() @ W.double().T).cpu().numpy(), rtol=1e-5, atol=2e-4)
    # the same empty range in the bf16x3 loop (its own prefetch code)
    out3 = torch.zeros(128, 128, device=DEV)
    ops.gemm(A, W, out3, 128, 128, 288, splitk=4, accumulate=True, precision=1)
    np.testing.assert_allclose(out3.cpu().double().numpy(), (A.double() @ W.double().T).cpu().numpy(), rtol=1e-5, atol=2e-4)
    # tile = 32 re-implements the epilogue: inference BatchNorm over the rows, dropout and K segments against tile = 64
    M, N, K = 128, 256, 512
    A2 = G(rng.standard_normal((2, M, K // 2)).astype(np.float32))          # two K segments, 256 apart in k
    W2 = G(rng.standard_normal((N, K)).astype(np.float32))
    bn = [G(rng.standard_normal(M).astype(np.float32)) for _ in range(3)] + [G((rng.random(M) + 0.5).astype(np.float32))]
    outs = {}
    for tile in (32, 64):
        o = torch.empty(M, N, device=DEV)
        ops.gemm(A2, W2, o, M, N, K, lda=K // 2, kseg=K // 2, segA=M * (K // 2), segB=K // 2, relu=True, drop_p=0.25, drop_seed=99,
                 bn=(bn[0], bn[1], bn[2], bn[3], 1e-5), tile=tile)
        outs[tile] = o
    assert float((outs[32] == 0).float().mean()) > 0.3                        # relu and dropout both took effect
    assert torch.equal(outs[32] == 0, outs[64] == 0)                          # same mask
    np.testing.assert_allclose(outs[32].cpu().numpy(), outs[64].cpu().numpy(), rtol=1e-5, atol=1e-4)
    ref = torch.cat([A2[0], A2[1]], 1).double() @ W2.double().T
    sc = bn[0].double() / torch.sqrt(bn[3].double() + 1e-5)
    ref = torch.relu(ref * sc[:, None] + (bn[1].double() - bn[2].double() * sc)[:, None]) / 0.75
    kept = outs[64] != 0
    np.testing.assert_allclose(outs[64][kept].cpu().double().numpy(), ref[kept].cpu().numpy(), rtol=1e-4, atol=2e-4)
    A, W, out = G(np.ones((64, 96), np.float32)), G(np.ones((64, 96), np.float32)), torch.empty(64, 64, device=DEV)
    with pytest.raises(RuntimeError, match="tile=32"):
        ops.gemm(A, W, out, 64, 64, 96, tile=32)
