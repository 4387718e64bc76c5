"""Pins the CPU oracle against golden vectors produced by running the reference
(tests/golden/make_golden.py).  CPU only."""
import numpy as np
import torch

from oracle import spot_oracle as O
from spot_amd.detinit import det_array, det_input


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def sd_for(keys_shapes, seed):
    return {k: T(det_array(k, s, seed)).float() for k, s in keys_shapes.items()}


def fusion_shapes(d, ff, nl, prefix=""):
    sh = {}
    p = prefix + "." if prefix else ""

    def mha(n):
        sh[n + ".in_proj_weight"] = (3 * d, d)
        sh[n + ".in_proj_bias"] = (3 * d,)
        sh[n + ".out_proj.weight"] = (d, d)
        sh[n + ".out_proj.bias"] = (d,)

    def lin(n, o, i):
        sh[n + ".weight"] = (o, i)
        sh[n + ".bias"] = (o,)

    def ln(n):
        sh[n + ".weight"] = (d,)
        sh[n + ".bias"] = (d,)

    for i in range(nl):
        b = "%sencoder.layers.%d" % (p, i)
        for a in ("self_attn1", "self_attn2", "multihead_attn1", "multihead_attn2"):
            mha(b + "." + a)
        lin(b + ".linear11", ff, d); lin(b + ".linear12", d, ff)
        lin(b + ".linear21", ff, d); lin(b + ".linear22", d, ff)
        for n in ("norm11", "norm12", "norm13", "norm21", "norm22", "norm23"):
            ln(b + "." + n)
    b = p + "decoder.layers.0"
    mha(b + ".multihead_attn")
    lin(b + ".linear1", ff, d); lin(b + ".linear2", d, ff)
    ln(b + ".norm1"); ln(b + ".norm2"); ln(p + "decoder.norm")
    return sh


def test_fusion_tiny(golden):
    g = golden("g1_fusion_tiny")
    d, nh, nl, ff, B, ht, hs = [int(v) for v in g["dims"]]
    sd = sd_for(fusion_shapes(d, ff, nl), 11)
    a = lambda n, s: T(det_input(n, s, 11, -1, 1))
    out = O.fusion_network(a("src_temp", (B, d, ht, ht)), T(g["mask_temp"]),
                           a("src_search", (B, d, hs, hs)), T(g["mask_search"]),
                           a("pos_temp", (B, d, ht, ht)), a("pos_search", (B, d, hs, hs)),
                           sd, "", nh, nl)
    np.testing.assert_allclose(out.numpy(), g["hs"], rtol=1e-4, atol=2e-5)


def test_fusion_full(golden):
    g = golden("g2_fusion_full")
    d, nh, nl, ff, B, ht, hs = [int(v) for v in g["dims"]]
    sd = sd_for(fusion_shapes(d, ff, nl), 12)
    a = lambda n, s: T(det_input(n, s, 12, -1, 1))
    z = lambda *s: torch.zeros(*s, dtype=torch.bool)
    out = O.fusion_network(a("src_temp", (B, d, ht, ht)), z(B, ht, ht), a("src_search", (B, d, hs, hs)),
                           z(B, hs, hs), a("pos_temp", (B, d, ht, ht)), a("pos_search", (B, d, hs, hs)),
                           sd, "", nh, nl).numpy()
    np.testing.assert_allclose(out[:, :, ::16, ::8], g["hs_slice"], rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(np.linalg.norm(out, axis=-1), g["hs_rownorm"], rtol=1e-4)


def test_mha(golden):
    g = golden("g3_mha")
    d, nh, B = 256, 8, 2
    sd = sd_for({"m.in_proj_weight": (3 * d, d), "m.in_proj_bias": (3 * d,),
                 "m.out_proj.weight": (d, d), "m.out_proj.bias": (d,)}, 13)
    # the golden module was filled with keys without the "m." prefix
    sd = {"m." + k: T(det_array(k, s, 13)).float() for k, s in
          {"in_proj_weight": (3 * d, d), "in_proj_bias": (3 * d,), "out_proj.weight": (d, d),
           "out_proj.bias": (d,)}.items()}
    for tag, (Lq, Lk) in {"tt": (256, 256), "ss": (1024, 1024), "ts": (256, 1024), "st": (1024, 256)}.items():
        a = lambda n, L: T(det_input(n + tag, (L, B, d), 13, -1, 1)).transpose(0, 1)
        xq, pq = a("xq", Lq), a("pq", Lq)
        if tag in ("tt", "ss"):
            xk, pk = xq, pq
        else:
            xk, pk = a("xk", Lk), a("pk", Lk)
        o = O.mha(xq, pq, xk, pk, sd, "m", nh, T(g["kpm_" + tag])).transpose(0, 1).numpy()
        np.testing.assert_allclose(o[::8], g["out_" + tag], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(np.linalg.norm(o, axis=-1), g["rownorm_" + tag], rtol=1e-4)


def test_layer_backward(golden):
    g = golden("g4_layer_bwd")
    d, nh, ff, B, Lt, Ls = [int(v) for v in g["dims"]]
    shapes = {k[len("encoder.layers.0."):]: v for k, v in fusion_shapes(d, ff, 1).items()
              if k.startswith("encoder.layers.0.")}
    sd = {"L." + k: T(det_array(k, s, 14)).float().requires_grad_() for k, s in shapes.items()}
    a = lambda n, L: T(det_input(n, (L, B, d), 14, -1, 1)).transpose(0, 1).contiguous()
    s1, s2 = a("s1", Lt).requires_grad_(), a("s2", Ls).requires_grad_()
    o1, o2 = O.fusion_layer(s1, s2, a("p1", Lt), a("p2", Ls), None, None, sd, "L", nh)
    ((o1 * a("g1", Lt)).sum() + (o2 * a("g2", Ls)).sum()).backward()
    np.testing.assert_allclose(o1.detach().transpose(0, 1).numpy(), g["out1"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(o2.detach().transpose(0, 1).numpy(), g["out2"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(s1.grad.transpose(0, 1).numpy(), g["ds1"], rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(s2.grad.transpose(0, 1).numpy(), g["ds2"], rtol=1e-3, atol=1e-4)
    for k, v in sd.items():
        np.testing.assert_allclose(v.grad.numpy(), g["grad." + k[2:]], rtol=2e-3, atol=2e-4, err_msg=k)


def test_decode_and_filter(golden):
    g = golden("g6_decode")
    dec = O.decode_teacher(T(g["logits"]), T(g["boxes"]), T(g["ious"]), 0.40, 32, 256, 256)
    np.testing.assert_allclose(dec["best_bbox"].numpy(), g["best_bbox"], rtol=1e-6, atol=1e-5)
    np.testing.assert_allclose(dec["best_score"].numpy(), g["best_score"], rtol=1e-6)
    conf = T(g["best_conf"])          # includes the hand-placed tau-eps / tau / tau+eps samples
    res = O.middleware_filter(T(g["best_bbox"]), conf, T(g["mid_unlabeled_box_extract"]),
                              T(g["mid_unlabeled_resize_factors"]), T(g["mid_aug_box_extract"]),
                              T(g["mid_aug_resize_factors"]), 256, 0.85)
    assert np.array_equal(res["aug_success"].numpy(), g["aug_success"])
    assert list(g["aug_success"][3:6]) == [False, True, True]
    np.testing.assert_allclose(res["aug_anno"].numpy(), g["aug_anno"], rtol=1e-6, atol=1e-4)


def test_compaction(golden):
    g = golden("g7_compaction")
    names = [k[5:] for k in g.files if k.startswith("mask_")]
    assert len(names) >= 6
    for n in names:
        idx = O.compact_indices(g["mask_" + n], len(g["mask_" + n]))
        assert np.array_equal(idx, g["idx_" + n]), n
    assert O.compact_indices(np.zeros(24, bool), 24) is None


def test_boxes(golden):
    g = golden("g8_boxes")
    b1, b2 = T(g["b1"]), T(g["b2"])
    x1, x2 = O.box_cxcywh_to_xyxy(b1), O.box_cxcywh_to_xyxy(b2)
    giou, iou = O.generalized_box_iou_pair_xyxy(x1, x2)
    iou2, union = O.box_iou_pair_xyxy(x1, x2)
    np.testing.assert_allclose(x1.numpy(), g["xyxy1"], rtol=0, atol=0)
    np.testing.assert_allclose(giou.numpy(), g["giou"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(iou.numpy(), g["iou"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(union.numpy(), g["union"], rtol=1e-6)
    np.testing.assert_allclose(O.box_xyxy_to_xywh(x1).numpy(), g["xywh1"], rtol=0, atol=0)
    np.testing.assert_allclose(O.box_xyxy_to_cxcywh(x1).numpy(), g["cxcywh1"], rtol=0, atol=0)


def test_matcher_and_criterion(golden):
    g = golden("g9_matcher_criterion")
    gt = g["gt"]
    B, L = 8, 1024
    logits = det_input("logits", (B, L, 2), 51, -2, 2)
    boxes = det_input("boxes", (B, L, 4), 51, 0.05, 0.6)
    ious = det_input("ious", (B, L, 1), 51, 0.0, 1.0)
    wd = {'loss_ce_sup': 8.334, 'loss_bbox_sup': 5, 'loss_giou_sup': 2, 'loss_iou_pred_sup': 1.0,
          'loss_ce_unsup': 8.334, 'loss_bbox_unsup': 5, 'loss_giou_unsup': 2}
    for tag in ("sup", "unsup"):
        pos, neg = O.matcher_masks(gt, L, with_neg=(tag == "sup"))
        assert np.array_equal(pos, g["pos_mask_" + tag])
        assert np.array_equal(neg, g["neg_count_" + tag])
        lg, bx = T(logits).requires_grad_(), T(boxes).requires_grad_()
        io = T(ious).requires_grad_() if tag == "sup" else None
        losses = O.criterion(lg, bx, io, T(gt), 0.0625, tag)
        for k, v in losses.items():
            np.testing.assert_allclose(v.detach().numpy(), g["%s_%s" % (k, tag)], rtol=2e-5, err_msg=k)
        total = sum(losses[k] * wd[k + "_" + tag] for k in losses if k + "_" + tag in wd)
        total.backward()
        np.testing.assert_allclose(lg.grad.numpy(), g["grad_logits_" + tag], rtol=1e-4, atol=1e-8)
        np.testing.assert_allclose(bx.grad.numpy(), g["grad_boxes_" + tag], rtol=1e-4, atol=1e-7)
        if io is not None:
            np.testing.assert_allclose(io.grad.numpy(), g["grad_ious_" + tag], rtol=1e-4, atol=1e-8)


def test_tracker_post(golden):
    g = golden("g11_tracker_post")
    bb = g["init_bbox"]
    center = np.array([bb[0] + bb[2] / 2, bb[1] + bb[3] / 2])
    size = np.array([bb[2], bb[3]])
    h = np.hanning(32)
    window = np.outer(h, h).flatten()
    for i in range(g["logits"].shape[0]):
        bbox, pscore, center, size, _ = O.tracker_post(g["logits"][i], g["boxes"][i], window, 0.49,
                                                       center, size, tuple(g["image_hw"]))
        np.testing.assert_allclose(bbox, g["target_bbox"][i], rtol=1e-6, atol=1e-4)
        np.testing.assert_allclose(pscore.max(), g["best"][i], rtol=1e-6)
        np.testing.assert_allclose(center, g["center"][i], rtol=1e-6, atol=1e-4)


def test_xcorr(golden):
    g = golden("g12_xcorr")
    for tag in ("ocean255", "cfg5"):
        B, C, hx, hz = [int(v) for v in g["dims_" + tag]]
        shp_x = [(hx - 2, hx - 2), (hx - 4, hx - 2), (hx - 2, hx - 4)]
        shp_z = [(hz - 2, hz - 2), (hz - 4, hz - 2), (hz - 2, hz - 4)]
        xs = [det_input("x%d%s" % (i, tag), (B, C) + s, 71, -1, 1) for i, s in enumerate(shp_x)]
        zs = [det_input("z%d%s" % (i, tag), (B, C) + s, 71, -1, 1) for i, s in enumerate(shp_z)]
        np.testing.assert_allclose(O.xcorr_depthwise(xs[0], zs[0]), g["re11_" + tag], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(O.groupdw(xs, zs, g["weight"]), g["groupdw_" + tag], rtol=1e-4, atol=1e-5)


def test_ema(golden):
    g = golden("g13_ema")

    def mk(seed):
        m = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3), torch.nn.BatchNorm2d(8), torch.nn.Linear(8, 4))
        sd = {k: T(det_array(k, v.shape, seed)).to(v.dtype) for k, v in m.state_dict().items()}
        sd["1.num_batches_tracked"] = torch.tensor(7 if seed == 81 else 100)
        return sd

    stu, tea = mk(81), mk(82)
    new = O.ema_update(tea, stu, 0.998)
    for k, v in new.items():
        np.testing.assert_allclose(v.numpy(), g["after." + k], rtol=1e-6, atol=1e-7, err_msg=k)
    cp = O.ema_update(new, stu, 0)
    for k, v in cp.items():
        np.testing.assert_allclose(v.numpy(), g["copied." + k], rtol=1e-6, atol=1e-7, err_msg=k)


def test_preprocess_and_posenc(golden):
    g = golden("g14_pre_pos")
    img = det_input("img", (2, 3, 16, 16), 91, 0, 256, np.uint8)
    avg = det_input("avg", (2, 3), 91, 0, 255)
    o, a = O.preprocess(T(img), T(avg), [0.485, 0.456, 0.406], [0.229, 0.224, 0.225])
    np.testing.assert_allclose(o.numpy(), g["img_out"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(a.numpy(), g["avg_out"], rtol=1e-6, atol=1e-6)
    for hw in (16, 32):
        p = O.position_embedding_sine(torch.zeros(1, hw, hw, dtype=torch.bool))
        np.testing.assert_allclose(p.numpy(), g["pos%d" % hw], rtol=1e-5, atol=1e-6)
    p = O.position_embedding_sine(T(g["mask8"]))
    np.testing.assert_allclose(p.numpy(), g["pos8_masked"], rtol=1e-5, atol=1e-6)


def _transform_inputs(g):
    seed = int(g["seed"])
    B, C, H, W = 8, 3, 32, 48
    return (T(det_input("img", (B, C, H, W), seed, -2, 2)), T(det_input("tmpl", (B, C, 16, 24), seed + 1, -2, 2)),
            T(det_input("avg", (B, 3), seed, -1, 1)), T(det_input("avg_t", (B, 3), seed + 1, -1, 1)))


def test_g16_in_network_transforms(golden):
    """GrayScale and EllipseRotationAffine restated for given draws == the reference run with those draws."""
    g = golden("g16_transforms")
    img, tmpl, avg, avg_t = _transform_inputs(g)
    im, bb = O.ellipse_rotation_affine(img, T(g["bbox"]).clone(), avg, T(g["mask"]), T(g["angle"]), T(g["lr_flip"]))
    assert torch.equal(im, T(g["era_img"]))
    np.testing.assert_allclose(bb.numpy(), g["era_bbox"], rtol=0, atol=1e-5)
    assert bool(T(g["lr_flip"]).any()) and bool((T(g["angle"]).abs() > 90).any())        # flips are exercised
    imgs, avgs = O.gray_scale([tmpl, img], [avg_t, avg], T(g["gray_tag"]))
    assert torch.equal(imgs[0], T(g["gray_tmpl"])) and torch.equal(imgs[1], T(g["gray_img"]))
    np.testing.assert_allclose(avgs[0].numpy(), g["gray_avg_t"], rtol=0, atol=1e-7)
    np.testing.assert_allclose(avgs[1].numpy(), g["gray_avg"], rtol=0, atol=1e-7)
    assert bool(T(g["gray_tag"]).any()) and not bool(T(g["gray_tag"]).all())

