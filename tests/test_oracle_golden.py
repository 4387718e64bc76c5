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


def test_matcher_and_criterion_for_boxes_off_the_map(golden):
    """Boxes partly and wholly beyond the 32x32 map (the reference's numpy slices wrap for negative bounds), empty cell ranges, the whole map."""
    g = golden("g22_matcher_offmap")
    gt = g["gt"]
    B, L = gt.shape[0], 1024
    logits = det_input("logits", (B, L, 2), 53, -2, 2)
    boxes = det_input("boxes", (B, L, 4), 53, 0.05, 0.6)
    ious = det_input("ious", (B, L, 1), 53, 0.0, 1.0)
    # the cases are really in there: a wrapped slice (box 0 lies left of the map, its positives are columns 0..27),
    # an empty set (box 2 lies right of it)
    assert g["pos_mask_sup"][0].reshape(32, 32)[16, :28].all() and not g["pos_mask_sup"][2].any()
    for tag in ("sup", "unsup"):
        pos, neg = O.matcher_masks(gt, L, with_neg=(tag == "sup"))
        assert np.array_equal(pos, g["pos_mask_" + tag])
        assert np.array_equal(neg, g["neg_count_" + tag])
        lg, bx = T(logits).requires_grad_(), T(boxes).requires_grad_()
        io = T(ious).requires_grad_() if tag == "sup" else None
        losses = O.criterion(lg, bx, io, T(gt), 0.0625, tag)
        for k, v in losses.items():
            np.testing.assert_allclose(v.detach().numpy(), g["%s_%s" % (k, tag)], rtol=2e-5, err_msg=k)


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



def _ocean_state(g):
    """Seeded parameters of the Siamese head's neck and tower, as the golden run filled them."""
    from spot_amd import ocean
    from spot_amd.detinit import det_fill_positive_norm_
    B, Cb, Ch, S = (int(v) for v in g["dims"])
    seed = int(g["seed"])
    neck = det_fill_positive_norm_(ocean.AdjustLayer(Cb, Ch), seed)
    tower = det_fill_positive_norm_(ocean.box_tower(inchannels=Ch, outchannels=Ch, towernum=1, iou_prediction=True), seed + 1)
    xf, zf = T(det_input("xf", (B, Cb, 31, 31), seed, -1, 1)), T(det_input("zf", (B, Cb, 15, 15), seed, -1, 1))
    return neck, tower, xf, zf


def test_g17_ocean_head_forward_and_gradients(golden):
    """Neck + box_tower restated == the reference's modules, train mode (with gradients) and eval mode."""
    g = golden("g17_ocean")
    neck, tower, xf, zf = _ocean_state(g)
    seed = int(g["seed"])
    nsd = {k: v.clone() for k, v in neck.state_dict().items()}
    tsd = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in tower.state_dict().items()}
    for k in list(tsd):
        if "running" in k:
            tsd[k] = tsd[k].detach()
    nsd["downsample.0.weight"].requires_grad_()
    xf.requires_grad_(), zf.requires_grad_()
    _, z = O.ocean_adjust(zf, _prefixed(nsd), "n", True, crop=True)
    x = O.ocean_adjust(xf, _prefixed(nsd), "n", True)
    bbox, cls, cls_feat, reg_feat, ious = O.ocean_box_tower(x, z, tsd, True, batch_iou=2)
    for name, got in (("train_bbox", bbox), ("train_cls", cls), ("train_cls_feat", cls_feat), ("train_reg_feat", reg_feat),
                      ("train_ious", ious)):
        np.testing.assert_allclose(got.detach().numpy(), g[name], rtol=1e-4, atol=1e-5, err_msg=name)
    ups = {k: T(det_input("up_" + k, tuple(v.shape), seed, -1, 1))
           for k, v in (("bbox", bbox), ("cls", cls), ("iou", ious), ("reg_feat", reg_feat))}
    ((bbox * ups["bbox"]).sum() + (cls * ups["cls"]).sum() + (ious * ups["iou"]).sum() + (reg_feat * ups["reg_feat"]).sum()).backward()
    for key in (k for k in g.files if k.startswith("dtower.") and not k.endswith("tower.0.bias")):
        want = g[key]
        np.testing.assert_allclose(tsd[key[len("dtower."):]].grad.numpy(), want, rtol=2e-3, atol=2e-4 * np.abs(want).max(),
                                   err_msg=key)
    for got, key in ((xf.grad, "d_xf"), (zf.grad, "d_zf"), (nsd["downsample.0.weight"].grad, "dneck.downsample.0.weight")):
        np.testing.assert_allclose(got.numpy(), g[key], rtol=2e-3, atol=2e-4 * np.abs(g[key]).max(), err_msg=key)
    # the train-mode passes (template, then search) updated the running statistics
    np.testing.assert_allclose(nsd["downsample.1.running_mean"].numpy(), g["neck_running_mean"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(tsd["bbox_tower.1.running_var"].numpy(), g["tower_running_var"], rtol=1e-5, atol=1e-6)
    with torch.no_grad():
        _, z = O.ocean_adjust(zf, _prefixed(nsd), "n", False, crop=True)
        x = O.ocean_adjust(xf, _prefixed(nsd), "n", False)
        bbox_e, cls_e, _, reg_e, none = O.ocean_box_tower(x, z, tsd, False)
    assert none is None
    for name, got in (("eval_bbox", bbox_e), ("eval_cls", cls_e), ("eval_reg_feat", reg_e)):
        np.testing.assert_allclose(got.numpy(), g[name], rtol=1e-4, atol=1e-5, err_msg=name)
    np.testing.assert_allclose(O.ocean_pred_to_image(T(g["eval_bbox"]), 255).numpy(), g["pred_to_image"], rtol=0, atol=1e-5)
    B, S = int(g["dims"][0]), int(g["dims"][3])
    offs = O.ocean_align_offsets(T(g["pred_to_image"]).permute(0, 2, 3, 1).reshape(B, -1, 4), (S, S), 255)
    np.testing.assert_allclose(offs.numpy(), g["offsets"], rtol=1e-6, atol=1e-5)


def _prefixed(sd, p="n"):
    """View of a state dict under a name prefix (tensors shared, so in-place running-statistics updates land in sd)."""
    return {p + "." + k: v for k, v in sd.items()}


def test_g17_ocean_labels_and_losses(golden):
    """Label generator and criterion restated == the reference's, bit for bit on the labels."""
    g = golden("g17_ocean")
    S, seed = int(g["dims"][3]), int(g["seed"])
    anno = T(g["anno"])
    n = anno.shape[0]
    cls, reg, reg_w, iou_w = O.ocean_labels(anno, S, 8, 255)
    assert reg.dtype == torch.float64 and g["reg_label"].dtype == np.float64
    for got, key in ((cls, "cls_label"), (reg, "reg_label"), (reg_w, "reg_weight"), (iou_w, "iou_weight")):
        assert np.array_equal(got.numpy(), g[key]), key
    assert 0 < g["reg_weight"].sum() < g["iou_weight"].sum() and g["cls_label"].sum() > 0
    pred = T(det_input("pred_ltrb", (n, 4, S, S), seed, 1, 90)).requires_grad_()
    align = O.ocean_align_labels(pred, reg, reg_w)
    np.testing.assert_allclose(align.detach().numpy(), g["align_label"], rtol=1e-12, atol=0)
    logits = T(det_input("logits", (n, 1, S, S), seed, -3, 3)).requires_grad_()
    align_logits = T(det_input("align_logits", (n, 1, S, S), seed, -3, 3)).requires_grad_()
    pred_ious = T(det_input("pred_ious", (n, 1, S, S), seed, 0, 1)).requires_grad_()
    losses = O.ocean_losses(logits, cls, align_logits, align, pred, reg, reg_w, pred_ious, iou_w)
    for k, v in losses.items():
        np.testing.assert_allclose(v.detach().numpy(), g["loss." + k], rtol=1e-6, atol=1e-7, err_msg=k)
    (losses["loss_ce"] + 1.2 * losses["loss_align"] + 1.5 * losses["loss_bbox"] + 0.7 * losses["loss_iou_pred"]).backward()
    for got, key in ((pred.grad, "d_pred"), (logits.grad, "d_logits"), (align_logits.grad, "d_align_logits"),
                     (pred_ious.grad, "d_pred_ious")):
        np.testing.assert_allclose(got.numpy(), g[key], rtol=1e-5, atol=1e-9, err_msg=key)


def test_g18_crop_windows(golden):
    """The crop window / padding restated == the reference's get_subwindow and sample_target (calls without resize)."""
    g = golden("g18_crop")
    im = det_input("frame", (120, 160, 3), int(g["seed"]), 0, 256, np.uint8)
    avg = g["avg"]
    for i, (pos, sz) in enumerate(zip(g["sub_pos"], g["sub_sz"])):
        patch, _, _ = O.crop_window(im, pos, int(sz), avg)
        assert np.array_equal(patch.transpose(2, 0, 1)[None], g["sub%d" % i]), i
        assert np.array_equal(O.resize_linear_u8(patch, int(sz)), patch)
    for i, (bb, sz) in enumerate(zip(g["st_boxes"], g["st_sz"])):
        pos = [bb[0] + 0.5 * bb[2], bb[1] + 0.5 * bb[3]]
        patch, _, _ = O.crop_window(im, pos, int(sz), avg)
        assert np.array_equal(patch, g["st%d" % i]), i


def test_resize_rule_properties():
    """The fixed-point bilinear restatement: constant images stay constant, a 2:1 reduction is the box average, an
    integer up-scaling reproduces the source at the sample centres' pixels, outputs stay within the input range."""
    rng = np.random.default_rng(3)
    for n, S in ((37, 128), (300, 256), (64, 32), (129, 128), (5, 7)):
        const = np.full((n, n, 3), 173, np.uint8)
        assert np.array_equal(O.resize_linear_u8(const, S), np.full((S, S, 3), 173, np.uint8))
        p = rng.integers(0, 256, (n, n, 3), dtype=np.uint8)
        r = O.resize_linear_u8(p, S)
        assert r.shape == (S, S, 3) and r.min() >= p.min() and r.max() <= p.max()
    p = rng.integers(0, 256, (64, 64, 3), dtype=np.uint8)
    q = p.astype(np.int64)
    assert np.array_equal(O.resize_linear_u8(p, 32), ((q[0::2, 0::2] + q[0::2, 1::2] + q[1::2, 0::2] + q[1::2, 1::2] + 2) >> 2))
    ramp = np.tile(np.arange(0, 200, 2, dtype=np.uint8)[None, :, None], (100, 1, 3))          # linear in x
    up = O.resize_linear_u8(ramp, 300).astype(np.int64)
    want = np.clip(((np.arange(300) + 0.5) / 3 - 0.5), 0, 99) * 2
    assert np.abs(up[150, :, 0] - want).max() <= 1


def test_g19_ocean_actor_decode(golden):
    """OceanSparseActor's pseudo-label decoding restated == the reference's (ocean_sparse_actor.py:40-94)."""
    g = golden("g19_ocean_model")
    seed, B, S = int(g["seed"]), 6, 25
    lg, al, io = (T(det_input(n, (B, 1, S, S), seed, lo, hi)) for n, lo, hi in (("lg", -4, 4), ("al", -4, 4), ("io", 0, 1)))
    boxes = T(np.concatenate([det_input("b0", (B, 2, S, S), seed, -20, 120), det_input("b1", (B, 2, S, S), seed, 100, 260)], 1))
    boxes[0, 2:] = boxes[0, :2] + 0.25
    bbox, score, conf = O.ocean_decode_teacher(lg, al, io, boxes, 0.6, 0.30)
    np.testing.assert_allclose(bbox.numpy(), g["best_bbox"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(score.numpy(), g["best_score"], rtol=1e-6)
    np.testing.assert_allclose(conf.numpy(), g["best_conf"], rtol=0, atol=0)
    assert (g["best_bbox"][0, 2:] == 1.0).all()


def _ocean_tracker_case(g, i):
    S, seed = int(g["score_size"]), int(g["seed"])
    maps = (det_input("cls%d" % i, (1, 1, S, S), seed, -4, 4), det_input("ltrb%d" % i, (1, 4, S, S), seed, 5, 90),
            det_input("al%d" % i, (1, 1, S, S), seed, -4, 4))
    return maps, np.array([100.0 + 7 * i, 80.0 - 3 * i]), np.array([40.0 + 5 * i, 60.0 - 4 * i]), 0.8 + 0.15 * i


def test_g20_ocean_tracker_update(golden):
    """Ocean_Tracker.update restated == the reference's, and its crop windows (origin by Python's round)."""
    g = golden("g20_ocean_tracker")
    S = int(g["score_size"])
    window = np.outer(np.hanning(S), np.hanning(S))
    for i in range(4):
        (cls, ltrb, al), tpos, tsz, scale_z = _ocean_tracker_case(g, i)
        pos, sz, sc = O.ocean_tracker_update(cls[0, 0], ltrb[0], al[0, 0], tpos, tsz * scale_z, window, scale_z)
        np.testing.assert_allclose(pos, g["pos%d" % i], rtol=1e-6)
        np.testing.assert_allclose(sz, g["sz%d" % i], rtol=1e-6)
        np.testing.assert_allclose(sc, g["score%d" % i], rtol=1e-6)
    im = det_input("frame", (120, 160, 3), int(g["seed"]), 0, 256, np.uint8)
    for i, (pos, sz) in enumerate(zip(g["sub_pos"], g["sub_sz"])):
        sz = int(sz)
        c = (sz + 1) / 2
        x0, y0 = round(pos[0] - c), round(pos[1] - c)
        # the same window through crop_window: its origin floor(p - c + 0.5) differs from round() on exact halves only
        pad = np.zeros((sz, sz, 3), np.uint8)
        pad[:] = g["avg"].astype(np.uint8)
        ys, xs = np.arange(y0, y0 + sz), np.arange(x0, x0 + sz)
        iy, ix = (ys >= 0) & (ys < 120), (xs >= 0) & (xs < 160)
        pad[np.ix_(iy, ix)] = im[np.ix_(ys[iy], xs[ix])]
        assert np.array_equal(pad.transpose(2, 0, 1), g["sub%d" % i]), i


def _conv3x3_of_translated(x, w, dy, dx):
    """3x3 / pad 1 convolution that reads x[h + dy, w + dx] where it would read x[h, w], the map continued by zeros
    (the translation is applied to the zero-extended plane, so rows that move in from the border are kept)."""
    import torch.nn.functional as F
    P = max(abs(dy), abs(dx)) + 1
    H, W = x.shape[-2:]
    xp = F.pad(x, (P, P, P, P))
    moved = torch.zeros_like(xp)
    Hp, Wp = xp.shape[-2:]
    hs, he = max(0, -dy), min(Hp, Hp - dy)
    ws, we = max(0, -dx), min(Wp, Wp - dx)
    moved[..., hs:he, ws:we] = xp[..., hs + dy:he + dy, ws + dx:we + dx]
    return F.conv2d(moved, w)[..., P - 1:P - 1 + H, P - 1:P - 1 + W]


def test_deform_conv_restatement_reduces_to_ordinary_convolution():
    """The deformable-convolution restatement has no fixture of the reference behind it, so it is anchored to the
    operator's defining properties (ltr/models/neck/dcn/src/deform_conv_cuda_kernel.cu:220-227: channel 2k of the
    offsets moves tap k along rows, 2k+1 along columns; a sample is zero unless -1 < h < H and -1 < w < W):
    zero offsets give aten's convolution; whole-pixel offsets of one tap give that tap's weight on the translated,
    zero-extended map; a half-pixel offset gives the mean of the two neighbouring translations; deformable groups move
    only their own channels."""
    import torch.nn.functional as F
    B, C, Co, H, W = 2, 6, 5, 9, 11
    x = T(det_input("x", (B, C, H, W), 31, -1, 1)).double()
    w = T(det_input("w", (Co, C, 3, 3), 32, -0.5, 0.5)).double()
    zero = torch.zeros(B, 18, H, W, dtype=torch.float64)
    ref = F.conv2d(x, w, None, 1, 1)
    torch.testing.assert_close(O.deform_conv(x, zero, w, 1, 1, 1), ref, rtol=1e-12, atol=1e-12)
    # strided, dilated, unpadded forms
    for stride, pad, dil in ((2, 1, 1), (1, 2, 2), (1, 0, 1)):
        r = F.conv2d(x, w, None, stride, pad, dil)
        z = torch.zeros(B, 18, r.shape[2], r.shape[3], dtype=torch.float64)
        torch.testing.assert_close(O.deform_conv(x, z, w, stride, pad, dil), r, rtol=1e-12, atol=1e-12)
    # one tap moved by whole pixels: rows on channel 2k, columns on channel 2k + 1
    for tap, (dy, dx) in ((0, (2, 0)), (4, (0, -3)), (7, (-1, 2))):
        off = zero.clone()
        off[:, 2 * tap], off[:, 2 * tap + 1] = dy, dx
        ky, kx = divmod(tap, 3)
        w_rest = w.clone()
        w_rest[:, :, ky, kx] = 0
        w_tap = torch.zeros_like(w)
        w_tap[:, :, ky, kx] = w[:, :, ky, kx]
        want = F.conv2d(x, w_rest, None, 1, 1) + _conv3x3_of_translated(x, w_tap, dy, dx)
        torch.testing.assert_close(O.deform_conv(x, off, w, 1, 1, 1), want, rtol=1e-12, atol=1e-12)
    # half a pixel down on every tap: the mean of the map and its translation by one row
    off = zero.clone()
    off[:, 0::2] = 0.5
    want = 0.5 * (ref + _conv3x3_of_translated(x, w, 1, 0))
    torch.testing.assert_close(O.deform_conv(x, off, w, 1, 1, 1), want, rtol=1e-12, atol=1e-12)
    # two deformable groups: the second group's offsets leave the first half of the channels alone
    off2 = torch.zeros(B, 36, H, W, dtype=torch.float64)
    off2[:, 18 + 1::2] = 1.0                                        # second group, one column to the right
    w_lo, w_hi = w.clone(), w.clone()
    w_lo[:, C // 2:] = 0
    w_hi[:, :C // 2] = 0
    want = F.conv2d(x, w_lo, None, 1, 1) + _conv3x3_of_translated(x, w_hi, 0, 1)
    torch.testing.assert_close(O.deform_conv(x, off2, w, 1, 1, 1, 2), want, rtol=1e-12, atol=1e-12)
