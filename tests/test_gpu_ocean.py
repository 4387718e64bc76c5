"""The Siamese (SP-Ocean) head on the GPU against the golden vectors of the reference's own modules (g17) and
against the CPU oracle: odd-sized maps through the patch-matrix convolution and the BatchNorm kernels, the towers,
box decoding, label generator and criterion.  Needs a real MI355X."""
import types

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import spot_oracle as O
from spot_amd.detinit import det_fill_positive_norm_, det_input

pytestmark = pytest.mark.gpu
DEV = "cuda"


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def G(a):
    return T(a).to(DEV)


@pytest.fixture(scope="module")
def ops():
    from spot_amd import lib, ops as _ops
    lib.load()
    return _ops


def _close(got, want, rtol, name=""):
    want = np.asarray(want)
    np.testing.assert_allclose(got.detach().cpu().numpy(), want, rtol=0, atol=rtol * max(float(np.abs(want).max()), 1e-6),
                               err_msg=name)


@pytest.mark.parametrize("hw,k,pad,dil,bias", [((31, 31), 3, 0, (1, 1), False), ((31, 31), 3, 0, (2, 1), False),
                                               ((7, 7), 3, 0, (1, 2), False), ((25, 25), 3, 1, (1, 1), True),
                                               ((31, 31), 1, 0, (1, 1), False), ((16, 24), 3, 1, (1, 1), True)])
def test_conv_on_odd_maps_matches_aten(ops, hw, k, pad, dil, bias):
    """Patch matrix + GEMM at a padded row pitch: forward, input, weight and bias gradients against ATen on the CPU
    (float64), for map sizes whose position count is not a multiple of 4, per-axis dilation, with and without bias."""
    B, Ci, Co = 3, 16, 12
    conv = torch.nn.Conv2d(Ci, Co, k, padding=pad, dilation=dil, bias=bias)
    x = T(det_input("x", (B, Ci) + hw, 5, -1, 1))
    ref = torch.nn.Conv2d(Ci, Co, k, padding=pad, dilation=dil, bias=bias).double()
    ref.load_state_dict({n: v.double() for n, v in conv.state_dict().items()})
    xr = x.double().requires_grad_()
    yr = ref(xr)
    up = T(det_input("up", tuple(yr.shape), 6, -1, 1))
    (yr * up.double()).sum().backward()
    conv = conv.to(DEV)
    xg = x.to(DEV).requires_grad_()
    y = ops.conv_general(xg, conv)
    assert y.shape == yr.shape and y.is_contiguous()
    (y * up.to(DEV)).sum().backward()
    _close(y, yr.detach().numpy(), 2e-6, "y")
    _close(xg.grad, xr.grad.numpy(), 2e-6, "dx")
    _close(conv.weight.grad, ref.weight.grad.numpy(), 2e-6, "dw")
    if bias:
        _close(conv.bias.grad, ref.bias.grad.numpy(), 2e-6, "db")


@pytest.mark.parametrize("hw", [(25, 25), (5, 3), (29, 27)])
@pytest.mark.parametrize("relu", [False, True])
def test_batchnorm_on_odd_maps_matches_aten(ops, hw, relu):
    """The element-wise walk of the BatchNorm kernels (H*W not a multiple of 4): training forward, running
    statistics, backward, and the inference pass."""
    B, Cn = 5, 12
    bn = det_fill_positive_norm_(torch.nn.Sequential(torch.nn.BatchNorm2d(Cn)), 3)[0]
    x = T(det_input("x", (B, Cn) + hw, 7, -2, 2))
    up = T(det_input("up", (B, Cn) + hw, 8, -1, 1))
    ref = torch.nn.BatchNorm2d(Cn).double()
    ref.load_state_dict({n: (v.double() if v.is_floating_point() else v) for n, v in bn.state_dict().items()})
    xr = x.double().requires_grad_()
    yr = ref(xr)
    yr = F.relu(yr) if relu else yr
    (yr * up.double()).sum().backward()
    bn = bn.to(DEV).train()
    xg = x.to(DEV).requires_grad_()
    y = ops.batch_norm_act(xg, bn, relu=relu)
    (y * up.to(DEV)).sum().backward()
    _close(y, yr.detach().numpy(), 3e-6, "y")
    _close(xg.grad, xr.grad.numpy(), 2e-5, "dx")
    _close(bn.weight.grad, ref.weight.grad.numpy(), 2e-5, "dgamma")
    _close(bn.bias.grad, ref.bias.grad.numpy(), 2e-5, "dbeta")
    _close(bn.running_mean, ref.running_mean.numpy(), 2e-6, "running_mean")
    _close(bn.running_var, ref.running_var.numpy(), 2e-6, "running_var")
    bn.eval(), ref.eval()
    with torch.no_grad():
        ye = ops.batch_norm_act(x.to(DEV), bn, relu=relu)
        yre = F.relu(ref(x.double())) if relu else ref(x.double())
    _close(ye, yre.numpy(), 3e-6, "eval")


def _head(g):
    from spot_amd import ocean
    B, Cb, Ch, S = (int(v) for v in g["dims"])
    seed = int(g["seed"])
    neck = det_fill_positive_norm_(ocean.AdjustLayer(Cb, Ch), seed).to(DEV)
    tower = det_fill_positive_norm_(ocean.box_tower(inchannels=Ch, outchannels=Ch, towernum=1, iou_prediction=True),
                                    seed + 1).to(DEV)
    xf, zf = G(det_input("xf", (B, Cb, 31, 31), seed, -1, 1)), G(det_input("zf", (B, Cb, 15, 15), seed, -1, 1))
    return neck, tower, xf, zf


def test_ocean_head_golden(ops, golden):
    """Neck + box_tower through the library's kernels == the reference's modules: train-mode forward, gradients of
    inputs and parameters, running statistics, eval-mode forward, box decoding and align offsets."""
    from spot_amd import ocean
    g = golden("g17_ocean")
    neck, tower, xf, zf = _head(g)
    seed, B, S = int(g["seed"]), int(g["dims"][0]), int(g["dims"][3])
    neck.train(), tower.train()
    xf.requires_grad_(), zf.requires_grad_()
    _, z = neck(zf, crop=True)
    x = neck(xf, crop=False)
    bbox, cls, cls_feat, reg_feat, extra = tower(x, z, iou_prediction=True, batch_iou=2)
    for name, got in (("train_bbox", bbox), ("train_cls", cls), ("train_cls_feat", cls_feat), ("train_reg_feat", reg_feat),
                      ("train_ious", extra["pred_ious"])):
        _close(got, g[name], 2e-5, name)
    ups = {k: G(det_input("up_" + k, tuple(v.shape), seed, -1, 1))
           for k, v in (("bbox", bbox), ("cls", cls), ("iou", extra["pred_ious"]), ("reg_feat", reg_feat))}
    ((bbox * ups["bbox"]).sum() + (cls * ups["cls"]).sum() + (extra["pred_ious"] * ups["iou"]).sum()
     + (reg_feat * ups["reg_feat"]).sum()).backward()
    params = dict(tower.named_parameters())
    for key in (k for k in g.files if k.startswith("dtower.")):
        got = params[key[len("dtower."):]].grad
        if key.endswith("tower.0.bias"):        # a bias in front of batch-statistics BatchNorm: zero up to rounding
            assert float(got.abs().max()) < 2e-4 and float(np.abs(g[key]).max()) < 2e-4
        else:
            _close(got, g[key], 5e-4, key)
    _close(xf.grad, g["d_xf"], 5e-4, "d_xf")
    _close(zf.grad, g["d_zf"], 5e-4, "d_zf")
    _close(neck.downsample[0].weight.grad, g["dneck.downsample.0.weight"], 5e-4, "dneck")
    _close(neck.downsample[1].running_mean, g["neck_running_mean"], 1e-5, "neck running mean")
    _close(tower.bbox_tower[1].running_var, g["tower_running_var"], 1e-5, "tower running var")
    neck.eval(), tower.eval()
    with torch.no_grad():
        _, z = neck(zf.detach(), crop=True)
        x = neck(xf.detach(), crop=False)
        bbox_e, cls_e, _, reg_e, extra_e = tower(x, z, iou_prediction=False)
        assert "pred_ious" not in extra_e
        for name, got in (("eval_bbox", bbox_e), ("eval_cls", cls_e), ("eval_reg_feat", reg_e)):
            _close(got, g[name], 2e-5, name)
        gx, gy = ocean.response_grid(S, 255)
        p2i = ocean.pred_to_image(G(g["eval_bbox"]), torch.tensor(gx, dtype=torch.float32), torch.tensor(gy, dtype=torch.float32))
        np.testing.assert_allclose(p2i.cpu().numpy(), g["pred_to_image"], rtol=0, atol=1e-5)
        offs = ocean.align_offsets(G(g["pred_to_image"]).permute(0, 2, 3, 1).reshape(B, -1, 4), (S, S), 255)
        np.testing.assert_allclose(offs.cpu().numpy(), g["offsets"], rtol=1e-6, atol=1e-5)


def _settings(S=25):
    return types.SimpleNamespace(stride=8, output_sz=S, batch_size=8, search_sz=255, search_feature_sz=S, weight_dict={},
                                 device=DEV)


def test_ocean_labels_golden_bit_exact(ops, golden):
    """One launch writes the labels of the whole batch: identical to the reference's, regression labels in float64."""
    from spot_amd import ocean
    g = golden("g17_ocean")
    gen = ocean.OceanLabelGenerator(_settings())
    anno = G(g["anno"])
    cls = gen({"anno": anno}, tag="cls")
    reg, reg_w = gen({"anno": anno}, tag="reg")
    iou_w = gen({"anno": anno}, tag="iou_w")
    assert reg.dtype == torch.float64
    for got, key in ((cls, "cls_label"), (reg, "reg_label"), (reg_w, "reg_weight"), (iou_w, "iou_weight")):
        assert np.array_equal(got.cpu().numpy(), g[key]), key
    # any batch size, any box: against the oracle on random boxes, including degenerate and off-patch ones
    rng = np.random.default_rng(5)
    boxes = np.concatenate([rng.uniform(-60, 300, (257, 2)), rng.uniform(-5, 200, (257, 2))], 1).astype(np.float32)
    c2, r2, rw2, iw2 = O.ocean_labels(T(boxes), 25, 8, 255)
    b = G(boxes)
    assert torch.equal(gen({"anno": b}, tag="cls").cpu(), c2)
    rg, rwg = gen({"anno": b}, tag="reg")
    assert torch.equal(rg.cpu(), r2) and torch.equal(rwg.cpu(), rw2)
    assert torch.equal(gen({"anno": b}, tag="iou_w").cpu(), iw2)
    assert gen({"anno": b[:0]}, tag="cls").shape == (0, 25, 25)


def test_ocean_criterion_golden(ops, golden):
    """Mask-based losses == the reference's index-selected ones, values and gradients; the single-cell rule."""
    from spot_amd import ocean
    g = golden("g17_ocean")
    S, seed = int(g["dims"][3]), int(g["seed"])
    n = g["anno"].shape[0]
    gen = ocean.OceanLabelGenerator(_settings())
    crit = ocean.ocean_loss(types.SimpleNamespace(**vars(_settings())))
    pred = G(det_input("pred_ltrb", (n, 4, S, S), seed, 1, 90)).requires_grad_()
    logits = G(det_input("logits", (n, 1, S, S), seed, -3, 3)).requires_grad_()
    align_logits = G(det_input("align_logits", (n, 1, S, S), seed, -3, 3)).requires_grad_()
    pred_ious = G(det_input("pred_ious", (n, 1, S, S), seed, 0, 1)).requires_grad_()
    anno = G(g["anno"])
    reg, reg_w = gen({"anno": anno}, tag="reg")
    align = gen({"bbox_pred": pred, "reg_label": reg, "reg_weight": reg_w}, tag="align")
    np.testing.assert_allclose(align.detach().cpu().numpy(), g["align_label"], rtol=1e-12, atol=0)
    outputs = {"pred_logits_sup": logits, "cls_label_sup": gen({"anno": anno}, tag="cls"), "align_label_sup": align,
               "align_logits_sup": align_logits, "pred_boxes_sup": pred, "reg_label_sup": reg, "reg_weight_sup": reg_w,
               "pred_ious_sup": pred_ious, "iou_weight_sup": gen({"anno": anno}, tag="iou_w")}
    losses = crit(outputs, None, tag="sup")
    for k in ("loss_ce", "loss_align", "loss_bbox", "iou", "loss_iou_pred"):
        np.testing.assert_allclose(losses[k].detach().cpu().numpy(), g["loss." + k], rtol=2e-6, atol=1e-7, err_msg=k)
        assert losses[k].dtype == torch.from_numpy(g["loss." + k]).dtype, k
    (losses["loss_ce"] + 1.2 * losses["loss_align"] + 1.5 * losses["loss_bbox"] + 0.7 * losses["loss_iou_pred"]).backward()
    for got, key in ((pred.grad, "d_pred"), (logits.grad, "d_logits"), (align_logits.grad, "d_align_logits"),
                     (pred_ious.grad, "d_pred_ious")):
        np.testing.assert_allclose(got.cpu().numpy(), g[key], rtol=2e-5, atol=1e-9, err_msg=key)
    one = torch.zeros(1, S, S, device=DEV)
    one[0, 3, 4] = 1.0
    lg = G(det_input("one_logits", (1, 1, S, S), seed, -3, 3))
    np.testing.assert_allclose(crit._weighted_BCE(lg, one).cpu().numpy(), g["loss_ce_single_positive"], rtol=2e-6)
    # no positive cell at all: NaN in the reference's sum, which loss_cls / add_iouloss turn into 0
    none = torch.zeros(1, S, S, device=DEV)
    assert float(crit.loss_cls({"pred_logits_sup": lg, "cls_label_sup": none}, None, "sup")["loss_ce"]) == 0.0
    lb, iou = crit.add_iouloss(pred[:1], reg[:1], torch.zeros_like(reg_w[:1]), return_iou=True)
    assert float(lb.detach()) == 0.0 and float(iou.detach()) == 0.0


@pytest.mark.parametrize("hw,C,Co,dg", [((25, 25), 16, 12, 1), ((9, 13), 8, 5, 2), ((16, 16), 32, 32, 1)])
def test_deform_conv_matches_oracle(ops, hw, C, Co, dg):
    """Deformable convolution v1: forward, input, offset and weight gradients against the float64 restatement, with
    offsets that reach outside the map (samples beyond one pixel outside are zero) and several deformable groups.
    Parity is against the restated algorithm only: the reference holds no fixture for this operator."""
    B, H, W = 3, hw[0], hw[1]
    x = T(det_input("x", (B, C, H, W), 11, -1, 1))
    off = T(det_input("off", (B, dg * 18, H, W), 12, -3, 3))
    off[0, :, 0, 0] = 40.0                      # far outside
    off[1, 0::2, 2, 3] = -30.0
    off = off + 1e-3                            # keep sample points off the pixel rows / columns (kinks of the bilinear map)
    w = T(det_input("w", (Co, C, 3, 3), 13, -0.3, 0.3))
    up = T(det_input("up", (B, Co, H, W), 14, -1, 1))
    xr, orf, wr = (t.double().requires_grad_() for t in (x, off, w))
    yr = O.deform_conv(xr, orf, wr, 1, 1, 1, dg)
    (yr * up.double()).sum().backward()
    xg, og, wg = (t.to(DEV).requires_grad_() for t in (x, off, w))
    y = ops.deform_conv(xg, og, wg, 1, 1, 1, 1, dg)
    (y * up.to(DEV)).sum().backward()
    _close(y, yr.detach().numpy(), 3e-6, "y")
    _close(xg.grad, xr.grad.numpy(), 3e-6, "dx")
    _close(og.grad, orf.grad.numpy(), 1e-5, "doffset")
    _close(wg.grad, wr.grad.numpy(), 3e-6, "dw")
    assert float(og.grad[0, :, 0, 0].abs().max()) == 0.0          # invalid samples carry no gradient


def test_deform_conv_reduces_to_ordinary_convolution_at_head_size(ops):
    """Size-independent anchors of the deformable convolution at the aligned head's size (256 channels, 25 x 25,
    32 clips), against aten's convolution on the same device: zero offsets; one tap moved by whole pixels (rows on
    offset channel 2k, columns on 2k+1, ltr/models/neck/dcn/src/deform_conv_cuda_kernel.cu:220-227); half a pixel on
    every tap = the mean of two translations; with zero offsets the input and weight gradients are aten's."""
    from tests.test_oracle_golden import _conv3x3_of_translated
    B, C, Co, S = 32, 256, 256, 25
    x = G(det_input("x", (B, C, S, S), 41, -1, 1))
    w = G(det_input("w", (Co, C, 3, 3), 42, -0.05, 0.05))
    zero = torch.zeros(B, 18, S, S, device=DEV)
    ref = F.conv2d(x.double(), w.double(), None, 1, 1)
    _close(ops.deform_conv(x, zero, w, 1, 1, 1, 1, 1), ref.cpu().numpy(), 3e-6, "zero offsets")
    tap, dy, dx = 5, -2, 1
    off = zero.clone()
    off[:, 2 * tap], off[:, 2 * tap + 1] = dy, dx
    w_rest, w_tap = w.double().clone(), torch.zeros_like(w, dtype=torch.float64)
    w_rest[:, :, 1, 2] = 0
    w_tap[:, :, 1, 2] = w[:, :, 1, 2].double()
    want = F.conv2d(x.double(), w_rest, None, 1, 1) + _conv3x3_of_translated(x.double(), w_tap, dy, dx)
    _close(ops.deform_conv(x, off, w, 1, 1, 1, 1, 1), want.cpu().numpy(), 3e-6, "one tap moved")
    off = zero.clone()
    off[:, 1::2] = 0.5
    want = 0.5 * (ref + _conv3x3_of_translated(x.double(), w.double(), 0, 1))
    _close(ops.deform_conv(x, off, w, 1, 1, 1, 1, 1), want.cpu().numpy(), 3e-6, "half a pixel")
    # gradients: with zero offsets the input and weight gradients are aten's
    up = G(det_input("up", (B, Co, S, S), 43, -1, 1))
    xg, wg, og = x.clone().requires_grad_(), w.clone().requires_grad_(), zero.clone().requires_grad_()
    (ops.deform_conv(xg, og, wg, 1, 1, 1, 1, 1) * up).sum().backward()
    xr, wr = x.double().requires_grad_(), w.double().requires_grad_()
    (F.conv2d(xr, wr, None, 1, 1) * up.double()).sum().backward()
    _close(xg.grad, xr.grad.cpu().numpy(), 3e-6, "dx")
    _close(wg.grad, wr.grad.cpu().numpy(), 3e-6, "dw")


def test_align_head_and_fusion_pipeline(ops, golden):
    """The aligned head on the golden tower outputs: zero offsets make the deformable convolution an ordinary 3x3
    convolution; the whole pipeline (tower, box decoding, offsets, aligned head) equals the oracle's composition and
    carries gradients back into the regression branch through the offsets."""
    from spot_amd import ocean
    g = golden("g17_ocean")
    neck, tower, xf, zf = _head(g)
    B, Ch, S = int(g["dims"][0]), int(g["dims"][2]), int(g["dims"][3])
    head = det_fill_positive_norm_(ocean.AlignHead(Ch, Ch), 77).to(DEV)
    feat = G(g["eval_reg_feat"])
    with torch.no_grad():
        zero = torch.zeros(B, S * S, 18, device=DEV)
        plain = F.conv2d(torch.relu(F.conv2d(feat, head.rpn_conv.conv.weight, padding=1)), head.rpn_cls.weight, head.rpn_cls.bias)
        _close(head(feat, zero), plain.cpu().numpy(), 3e-6, "zero offsets")
    neck.eval(), tower.train()
    gx, gy = (torch.tensor(a, dtype=torch.float32) for a in ocean.response_grid(S, 255))
    with torch.no_grad():
        _, z = neck(zf, crop=True)
        x = neck(xf, crop=False)
    out = ocean.fusion_pipeline(tower, head, x, z, gx, gy, 255, iou_prediction=True, b_iou=B, pred_to_img_tag=True)
    assert set(out) == {"pred_logits", "pred_boxes", "align_logits", "pred_ious", "bbox_pred_to_img"}
    assert out["align_logits"].shape == (B, 1, S, S)
    sd = {k: v.detach().cpu().double() for k, v in tower.state_dict().items()}
    # the oracle's train-mode pass updates running statistics in its own copy; feed it the pre-update values
    ref_boxes = O.ocean_pred_to_image(out["pred_boxes"].detach().cpu(), 255)
    np.testing.assert_allclose(out["bbox_pred_to_img"].cpu().numpy(), ref_boxes.numpy(), rtol=0, atol=1e-4)
    offs = O.ocean_align_offsets(ref_boxes.permute(0, 2, 3, 1).reshape(B, -1, 4), (S, S), 255)
    hw = {k: v.detach().cpu().double() for k, v in head.state_dict().items()}
    # reg_feature is the tower's fourth output; recompute it for the oracle from the same parameters
    with torch.no_grad():
        _, _, _, reg_feat, _ = tower(x, z, iou_prediction=False)
    a = O.deform_conv(reg_feat.cpu().double(), offs.double().permute(0, 2, 1).reshape(B, 18, S, S), hw["rpn_conv.conv.weight"], 1, 1, 1)
    want = F.conv2d(torch.relu(a), hw["rpn_cls.weight"], hw["rpn_cls.bias"])
    _close(out["align_logits"], want.numpy(), 2e-4, "align logits")
    out["align_logits"].sum().backward()
    assert float(tower.bbox_pred.weight.grad.abs().max()) > 0        # through the offsets into the regression head
    assert float(tower.cls_pred.weight.grad.abs().max() if tower.cls_pred.weight.grad is not None else 0.0) == 0.0


@pytest.mark.parametrize("hw,pad,dil,bias,B", [((31, 31), 0, (1, 1), False, 3), ((25, 25), 1, (1, 1), True, 3),
                                               ((7, 7), 0, (1, 1), False, 5), ((3, 3), 0, (1, 1), False, 2),
                                               ((63, 63), 1, (1, 1), False, 2), ((29, 27), 1, (1, 1), True, 1),
                                               ((15, 15), 2, (1, 1), False, 4), ((16, 16), 1, (1, 1), False, 2),
                                               ((31, 31), 0, (2, 1), False, 3), ((29, 31), 0, (1, 2), False, 2),
                                               ((7, 7), 0, (2, 1), False, 4), ((7, 7), 0, (1, 2), True, 4),
                                               ((31, 31), 1, (2, 2), False, 2), ((15, 15), 1, (2, 2), False, 3),
                                               ((32, 32), 1, (2, 2), False, 1)])
def test_winograd_any_size_matches_aten(ops, hw, pad, dil, bias, B):
    """Winograd F(4x4, 3x3) with masked partial tiles: forward, input, weight and bias gradients against float64 ATen
    for map sides that are not multiples of 4, padding 0 / 1 / 2 (times the dilation), dilation 2 along one or both
    axes (polyphase sub-maps of unequal size on odd maps), tile counts that are not multiples of 32."""
    Ci, Co = 24, 20
    padding = (pad * dil[0], pad * dil[1])
    assert padding[0] == padding[1]
    conv = torch.nn.Conv2d(Ci, Co, 3, padding=padding, dilation=dil, bias=bias)
    x = T(det_input("x", (B, Ci) + hw, 15, -1, 1))
    ref = torch.nn.Conv2d(Ci, Co, 3, padding=padding, dilation=dil, bias=bias).double()
    ref.load_state_dict({n: v.double() for n, v in conv.state_dict().items()})
    xr = x.double().requires_grad_()
    yr = ref(xr)
    up = T(det_input("up", tuple(yr.shape), 16, -1, 1))
    (yr * up.double()).sum().backward()
    conv = conv.to(DEV)
    xg = x.to(DEV).requires_grad_()
    y = ops.Conv3x3GenFn.apply(xg, conv.weight, conv.bias, pad, dil)
    assert y.shape == yr.shape
    (y * up.to(DEV)).sum().backward()
    _close(y, yr.detach().numpy(), 3e-5, "y")
    _close(xg.grad, xr.grad.numpy(), 3e-5, "dx")
    _close(conv.weight.grad, ref.weight.grad.numpy(), 3e-5, "dw")
    if bias:
        _close(conv.bias.grad, ref.bias.grad.numpy(), 2e-6, "db")
    # conv_general picks this path for such a convolution; the patch-matrix path is a separate implementation
    with torch.no_grad():
        a = ops.conv_general(x.to(DEV), conv)
        b = ops.ConvIm2colFn.apply(x.to(DEV), conv.weight, 1, padding[0], dil, conv.bias)
    assert torch.equal(a, y.detach())
    _close(a, b.cpu().numpy(), 3e-5, "two paths")


def test_empty_batches(ops):
    """Zero clips: convolution, deformable convolution and the label generator return empty results of the right
    shape without launching anything."""
    from spot_amd import ocean
    conv = torch.nn.Conv2d(16, 16, 3, padding=1).to(DEV)
    assert ops.conv_general(torch.empty(0, 16, 25, 25, device=DEV), conv).shape == (0, 16, 25, 25)
    w = torch.randn(8, 16, 3, 3, device=DEV)
    y = ops.deform_conv(torch.empty(0, 16, 9, 9, device=DEV), torch.empty(0, 18, 9, 9, device=DEV), w, 1, 1, 1, 1, 1)
    assert y.shape == (0, 8, 9, 9)
    gen = ocean.OceanLabelGenerator(_settings())
    reg, reg_w = gen({"anno": torch.empty(0, 4, device=DEV)}, tag="reg")
    assert reg.shape == (0, 25, 25, 4) and reg_w.shape == (0, 25, 25)
