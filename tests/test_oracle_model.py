"""Pins the oracle's full-model path (backbone + fusion + heads + decode + filter, and the student
loss/gradients) against goldens produced by the reference.  CPU only."""
import numpy as np
import torch

from oracle import spot_oracle as O
from spot_amd.detinit import det_state_dict, det_input
from spot_amd.synthetic import Settings


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def model_sd(seed, requires_grad=False):
    """State dict with the reference's key names/shapes, taken from this package's module definition."""
    from spot_amd.transt import transt_resnet50
    net = transt_resnet50(Settings(device="cpu"))
    sd = {k: v.float() if v.dtype.is_floating_point else v for k, v in det_state_dict(net, seed).items()}
    if requires_grad:
        names = {k for k, _ in net.named_parameters()}
        sd = {k: (v.requires_grad_() if k in names else v) for k, v in sd.items()}
    return sd


def test_teacher_pipeline_golden(golden):
    g = golden("g5_teacher")
    B, seed = int(g["B"]), int(g["seed"])
    sd = model_sd(seed)
    sd["iou_embed.layers.2.bias"] = sd["iou_embed.layers.2.bias"] + float(g["iou_bias_shift"])
    key = T(det_input("key_images", (B, 3, 128, 128), seed, 0, 256, np.uint8))
    unl = T(det_input("unlabeled_images", (B, 3, 256, 256), seed, 0, 256, np.uint8))
    with torch.no_grad():
        out = O.transt_forward(key, unl, sd)
    np.testing.assert_allclose(out["pred_logits"].numpy(), g["pred_logits"], rtol=1e-3, atol=2e-4)
    np.testing.assert_allclose(out["pred_boxes"].numpy(), g["pred_boxes"], rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(out["pred_ious"].numpy(), g["pred_ious"], rtol=1e-3, atol=1e-4)
    dec = O.decode_teacher(out["pred_logits"], out["pred_boxes"], out["pred_ious"], 0.40, 32, 256, 256)
    np.testing.assert_allclose(dec["best_bbox"].numpy(), g["best_bbox"], atol=0.5)
    np.testing.assert_allclose(dec["best_conf"].numpy(), g["best_conf"], rtol=1e-3)
    # middleware on the golden decode (inputs as the golden script generated them)
    xy, wh = det_input("unl_xy", (B, 1, 2), seed, 0, 400), det_input("unl_wh", (B, 1, 2), seed, 20, 200)
    ue = np.concatenate([xy, wh], -1).astype(np.float32)
    xy, wh = det_input("aug_xy", (B, 1, 2), seed, 0, 400), det_input("aug_wh", (B, 1, 2), seed, 20, 200)
    ae = np.concatenate([xy, wh], -1).astype(np.float32)
    res = O.middleware_filter(dec["best_bbox"], dec["best_conf"], T(ue), T(det_input("unl_rf", (B,), seed, 0.5, 2.0)),
                              T(ae), T(det_input("aug_rf", (B,), seed, 0.5, 2.0)), 256, 0.85)
    assert np.array_equal(res["aug_success"].numpy(), g["aug_success"])
    np.testing.assert_allclose(res["aug_anno"].numpy(), g["aug_anno"], atol=0.5)


def student_batches(seed, bs, bu):
    def mk(pt, ps, n, sd):
        d = {}
        d[pt + "_images"] = T(det_input(pt + "_img", (n, 3, 128, 128), sd, 0, 256, np.uint8))
        d[ps + "_images"] = T(det_input(ps + "_img", (n, 3, 256, 256), sd, 0, 256, np.uint8))
        d[pt + "_avg"] = T(det_input(pt + "_avg", (n, 3), sd, 0, 255))
        d[ps + "_avg"] = T(det_input(ps + "_avg", (n, 3), sd, 0, 255))
        d[pt + "_anno"] = T(det_input(pt + "_anno", (n, 4), sd, 20, 60))
        xy, wh = det_input(ps + "_xy", (n, 2), sd, 30, 120), det_input(ps + "_wh", (n, 2), sd, 30, 100)
        d[ps + "_anno"] = T(np.concatenate([xy, wh], -1).astype(np.float32))
        return d
    return mk("template", "search", bs, seed), mk("relay", "aug", bu, seed + 100)


def test_student_step_golden(golden):
    g = golden("g15_student")
    seed, bs, bu = int(g["seed"]), int(g["bs"]), int(g["bu"])
    sd = model_sd(seed, requires_grad=True)
    sup, sp = student_batches(seed, bs, bu)
    np.testing.assert_array_equal(sup["search_anno"].numpy(), g["search_anno"])
    template = torch.cat([sup["template_images"], sp["relay_images"]])
    search = torch.cat([sup["search_images"], sp["aug_images"]])
    out = O.transt_forward(template, search, sd, b_iou=bs, training=True)
    o_sup = {"logits": out["pred_logits"][:bs], "boxes": out["pred_boxes"][:bs], "ious": out["pred_ious"]}
    o_un = {"logits": out["pred_logits"][bs:], "boxes": out["pred_boxes"][bs:]}
    wd = Settings().weight_dict
    loss, ls, lu = O.sp_training_loss(o_sup, o_un, sup["search_anno"], sp["aug_anno"], wd)
    np.testing.assert_allclose(loss.item(), float(g["loss"]), rtol=1e-3)
    loss.backward()
    gn = dict(zip([str(k) for k in g["gn_keys"]], g["gn_vals"]))
    worst = 0.0
    for k, v in sd.items():
        if v.requires_grad and k in gn and gn[k] > 1e-6:
            worst = max(worst, abs(float(v.grad.norm()) - gn[k]) / gn[k])
    assert worst < 2e-2, worst
    for k in [n[len("gslice."):] for n in g.files if n.startswith("gslice.")]:
        ref = g["gslice." + k]
        np.testing.assert_allclose(sd[k].grad.reshape(-1)[:64].numpy(), ref, rtol=2e-2,
                                   atol=2e-3 * max(1e-6, float(np.abs(ref).max())), err_msg=k)


def test_student_step_gradient_samples_golden(golden):
    """Every parameter of the fusion network, input_proj and the heads: the evenly strided gradient samples the
    reference produced, compared direction and size."""
    g = golden("g15_student")
    seed, bs, bu = int(g["seed"]), int(g["bs"]), int(g["bu"])
    sd = model_sd(seed, requires_grad=True)
    sup, sp = student_batches(seed, bs, bu)
    template = torch.cat([sup["template_images"], sp["relay_images"]])
    search = torch.cat([sup["search_images"], sp["aug_images"]])
    out = O.transt_forward(template, search, sd, b_iou=bs, training=True)
    o_sup = {"logits": out["pred_logits"][:bs], "boxes": out["pred_boxes"][:bs], "ious": out["pred_ious"]}
    o_un = {"logits": out["pred_logits"][bs:], "boxes": out["pred_boxes"][bs:]}
    loss, _, _ = O.sp_training_loss(o_sup, o_un, sup["search_anno"], sp["aug_anno"], Settings().weight_dict)
    loss.backward()
    check_gradient_samples(g, {k: v.grad for k, v in sd.items() if v.requires_grad and v.grad is not None})


def check_gradient_samples(g, grads, cos_min=0.9999, rel_max=2e-3):
    """grads: name -> gradient tensor.  Against the golden's `gsub.<name>` samples (<= 2048 evenly strided elements
    of each gradient): cosine >= cos_min per tensor and over all samples together, and the largest deviation
    <= rel_max of the tensor's largest sample."""
    names = [n[len("gsub."):] for n in g.files if n.startswith("gsub.")]
    assert len(names) > 100
    all_a, all_b = [], []
    for k in names:
        ref = g["gsub." + k].astype(np.float64)
        got = grads[k].detach().reshape(-1).cpu().double().numpy()
        got = got[::max(1, got.size // 2048)]
        assert got.shape == ref.shape, k
        scale = np.abs(ref).max()
        if scale < 1e-12:
            assert np.abs(got).max() < 1e-9, k
            continue
        cos = float(got @ ref / (np.linalg.norm(got) * np.linalg.norm(ref)))
        assert cos >= cos_min, (k, cos)
        assert np.abs(got - ref).max() <= rel_max * scale, (k, np.abs(got - ref).max() / scale)
        all_a.append(got / scale)
        all_b.append(ref / scale)
    a, b = np.concatenate(all_a), np.concatenate(all_b)
    assert float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b))) >= 0.99999


def teacher_b32_inputs(g):
    B, seed = int(g["B"]), int(g["seed"])
    key = det_input("key_images", (B, 3, 128, 128), seed, 0, 256, np.uint8)
    unl = det_input("unlabeled_images", (B, 3, 256, 256), seed, 0, 256, np.uint8)
    xy, wh = det_input("unl_xy", (B, 1, 2), seed, 0, 400), det_input("unl_wh", (B, 1, 2), seed, 20, 200)
    ue = np.concatenate([xy, wh], -1).astype(np.float32)
    xy, wh = det_input("aug_xy", (B, 1, 2), seed, 0, 400), det_input("aug_wh", (B, 1, 2), seed, 20, 200)
    ae = np.concatenate([xy, wh], -1).astype(np.float32)
    mid = {"unlabeled_box_extract": T(ue), "unlabeled_resize_factors": T(det_input("unl_rf", (B,), seed, 0.5, 2.0)),
           "aug_box_extract": T(ae), "aug_resize_factors": T(det_input("aug_rf", (B,), seed, 0.5, 2.0))}
    return T(key), T(unl), mid


def test_teacher_pipeline_golden_at_benchmark_batch(golden):
    """The 32-clip pseudo-labelling pass of the benchmark configuration against the reference's outputs."""
    g = golden("g5_teacher_b32")
    B, seed, st = int(g["B"]), int(g["seed"]), int(g["slice_step"])
    assert B == 32
    sd = model_sd(seed)
    sd["iou_embed.layers.2.bias"] = sd["iou_embed.layers.2.bias"] + float(g["iou_bias_shift"])
    key, unl, mid = teacher_b32_inputs(g)
    with torch.no_grad():
        out = O.transt_forward(key, unl, sd)
    np.testing.assert_allclose(out["pred_logits"].numpy()[:, ::st], g["pred_logits"], rtol=1e-3, atol=2e-4)
    np.testing.assert_allclose(out["pred_boxes"].numpy()[:, ::st], g["pred_boxes"], rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(out["pred_ious"].numpy()[:, ::st], g["pred_ious"], rtol=1e-3, atol=1e-4)
    dec = O.decode_teacher(out["pred_logits"], out["pred_boxes"], out["pred_ious"], 0.40, 32, 256, 256)
    np.testing.assert_allclose(dec["best_bbox"].numpy(), g["best_bbox"], atol=0.5)
    np.testing.assert_allclose(dec["best_score"].numpy(), g["best_score"], rtol=1e-3)
    np.testing.assert_allclose(dec["best_conf"].numpy(), g["best_conf"], rtol=1e-3)
    res = O.middleware_filter(dec["best_bbox"], dec["best_conf"], mid["unlabeled_box_extract"],
                              mid["unlabeled_resize_factors"], mid["aug_box_extract"], mid["aug_resize_factors"], 256, 0.85)
    assert np.array_equal(res["aug_success"].numpy(), g["aug_success"])
    assert 0 < int(g["aug_success"].sum()) < B
    np.testing.assert_allclose(res["aug_anno"].numpy(), g["aug_anno"], atol=0.5)


def test_teacher_pipeline_train_mode_golden_at_benchmark_batch(golden):
    """The same 32-clip pass with the teacher in train() mode (the mode the reference's trainer leaves it in, and the
    one the benchmark times): batch-statistics BatchNorm in both backbone passes, dropout 0; outputs, keep mask and
    the BatchNorm buffers after the pass."""
    g = golden("g23_teacher_train_b32")
    B, seed, st = int(g["B"]), int(g["seed"]), int(g["slice_step"])
    sd = model_sd(seed)
    sd["iou_embed.layers.2.bias"] = sd["iou_embed.layers.2.bias"] + float(g["iou_bias_shift"])
    key, unl, mid = teacher_b32_inputs(g)
    stats = {}
    with torch.no_grad():
        out = O.transt_forward(key, unl, sd, training=True, stats=stats)
    np.testing.assert_allclose(out["pred_logits"].numpy()[:, ::st], g["pred_logits"], rtol=1e-3, atol=2e-4)
    np.testing.assert_allclose(out["pred_boxes"].numpy()[:, ::st], g["pred_boxes"], rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(out["pred_ious"].numpy()[:, ::st], g["pred_ious"], rtol=1e-3, atol=1e-4)
    dec = O.decode_teacher(out["pred_logits"], out["pred_boxes"], out["pred_ious"], 0.40, 32, 256, 256)
    np.testing.assert_allclose(dec["best_bbox"].numpy(), g["best_bbox"], atol=0.5)
    np.testing.assert_allclose(dec["best_score"].numpy(), g["best_score"], rtol=1e-3)
    np.testing.assert_allclose(dec["best_conf"].numpy(), g["best_conf"], rtol=1e-3)
    res = O.middleware_filter(dec["best_bbox"], dec["best_conf"], mid["unlabeled_box_extract"],
                              mid["unlabeled_resize_factors"], mid["aug_box_extract"], mid["aug_resize_factors"], 256, 0.85)
    assert np.array_equal(res["aug_success"].numpy(), g["aug_success"])
    assert 0 < int(g["aug_success"].sum()) < B
    np.testing.assert_allclose(res["aug_anno"].numpy(), g["aug_anno"], atol=0.5)
    check_batchnorm_buffers(g, stats)


def check_batchnorm_buffers(g, buffers):
    """The golden's `bn.<layer>.<buffer>` entries against `buffers` (name -> tensor): two passes (search, template)
    have moved every layer's running statistics once each."""
    names = [n[len("bn."):] for n in g.files if n.startswith("bn.")]
    assert len(names) == 12
    for k in names:
        got = buffers[k].detach().cpu().numpy()
        if k.endswith("num_batches_tracked"):
            assert int(got) == int(g["bn." + k]) == 2, k
        else:
            np.testing.assert_allclose(got, g["bn." + k], rtol=1e-4, atol=1e-6, err_msg=k)


def burnin_batch(seed, bs):
    d = {"template_images": T(det_input("template_img", (bs, 3, 128, 128), seed, 0, 256, np.uint8)),
         "search_images": T(det_input("search_img", (bs, 3, 256, 256), seed, 0, 256, np.uint8)),
         "template_avg": T(det_input("template_avg", (bs, 3), seed, 0, 255)),
         "search_avg": T(det_input("search_avg", (bs, 3), seed, 0, 255)),
         "template_anno": T(det_input("template_anno", (bs, 4), seed, 20, 60))}
    xy, wh = det_input("search_xy", (bs, 2), seed, 30, 120), det_input("search_wh", (bs, 2), seed, 30, 100)
    d["search_anno"] = T(np.concatenate([xy, wh], -1).astype(np.float32))
    return d


BURNIN_STATS = {"Loss/ce": "loss_ce", "Loss/bbox": "loss_bbox", "Loss/giou": "loss_giou", "iou": "iou",
                "Loss/iou_pred": "loss_iou_pred"}


def test_burnin_step_golden(golden):
    """actor(mode="burnin") of the reference: loss, every statistic, every gradient norm and the gradient samples."""
    g = golden("g24_burnin")
    seed, bs = int(g["seed"]), int(g["bs"])
    sd = model_sd(seed, requires_grad=True)
    d = burnin_batch(seed, bs)
    np.testing.assert_array_equal(d["search_anno"].numpy(), g["search_anno"])
    out = O.transt_forward(d["template_images"], d["search_images"], sd, b_iou=bs, training=True)
    loss, ld = O.burnin_loss({"logits": out["pred_logits"], "boxes": out["pred_boxes"], "ious": out["pred_ious"]},
                             d["search_anno"], Settings().weight_dict)
    np.testing.assert_allclose(loss.item(), float(g["loss"]), rtol=1e-3)
    ref_stats = dict(zip([str(k) for k in g["stat_keys"]], g["stat_vals"]))
    assert set(ref_stats) == set(BURNIN_STATS) | {"Loss/total"}
    for name, key in BURNIN_STATS.items():
        np.testing.assert_allclose(ld[key].item(), ref_stats[name], rtol=2e-3, atol=1e-5, err_msg=name)
    loss.backward()
    gn = dict(zip([str(k) for k in g["gn_keys"]], g["gn_vals"]))
    worst = max(abs(float(v.grad.norm()) - gn[k]) / gn[k] for k, v in sd.items()
                if v.requires_grad and v.grad is not None and gn.get(k, 0) > 1e-6)
    assert worst < 2e-2, worst
    check_gradient_samples(g, {k: v.grad for k, v in sd.items() if v.requires_grad and v.grad is not None})


def test_online_template_track_golden(golden):
    """TransT.template / TransT.track at the network boundary and the tracker's state update over five frames."""
    g = golden("g21_track")
    seed, n = int(g["seed"]), int(g["n_frames"])
    sd = model_sd(seed)
    z = T(det_input("z_crop", (1, 3, 128, 128), seed, 0, 256, np.uint8))
    xs = T(det_input("x_crops", (n, 3, 256, 256), seed, 0, 256, np.uint8))
    bb = g["init_bbox"]
    center, size = np.array([bb[0] + bb[2] / 2, bb[1] + bb[3] / 2]), np.array([bb[2], bb[3]])
    h = np.hanning(32)
    window = np.outer(h, h).flatten()
    for i in range(n):
        with torch.no_grad():
            out = O.transt_forward(z, xs[i:i + 1], sd)
        np.testing.assert_allclose(out["pred_logits"].numpy()[0], g["pred_logits"][i], rtol=1e-3, atol=2e-4)
        np.testing.assert_allclose(out["pred_boxes"].numpy()[0], g["pred_boxes"][i], rtol=1e-3, atol=1e-4)
        bbox, pscore, center, size, _ = O.tracker_post(out["pred_logits"].numpy()[0], out["pred_boxes"].numpy()[0], window,
                                                       0.49, center, size, tuple(g["image_hw"]))
        np.testing.assert_allclose(bbox, g["target_bbox"][i], atol=0.5)
        np.testing.assert_allclose(pscore.max(), g["best"][i], rtol=1e-3)
