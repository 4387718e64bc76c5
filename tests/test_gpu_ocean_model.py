"""The Siamese variant's backbone, model wrapper and actor on the GPU: backbone against the reference's own module
(golden g19), pseudo-label decoding against the reference's actor (g19), and the assembled model end to end
(teacher inference, sparse-supervised student step, EMA, online template / track).  Needs a real MI355X."""
import types

import numpy as np
import pytest
import torch

from oracle import spot_oracle as O
from spot_amd.detinit import det_fill_, det_fill_positive_norm_, det_input
from spot_amd.tensordict import TensorDict

pytestmark = pytest.mark.gpu
DEV = "cuda"


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def G(a):
    return T(a).to(DEV)


def _close(got, want, rtol, name=""):
    want = np.asarray(want)
    np.testing.assert_allclose(got.detach().cpu().numpy(), want, rtol=0, atol=rtol * max(float(np.abs(want).max()), 1e-6),
                               err_msg=name)


def _settings(**kw):
    s = types.SimpleNamespace(
        device=DEV, used_layers=[3], before_neck_dim=1024, head_dim=256, head_tower_num=1, iou_prediction_tag=True,
        strong_augmentation_tag=False, aug_settings={"gray": 0.0, "jitter": 0.0, "angle": 10, "lr_f_prob": 0.35, "ud_f_prob": 0.1},
        normalize_mean=[0.485, 0.456, 0.406], normalize_std=[0.229, 0.224, 0.225], search_sz=255, temp_sz=127, output_sz=25,
        batch_size=8, stride=8, search_feature_sz=25,
        curriculum_settings={"confidence_threshold": 0.82, "window_penalty": 0.30, "cls_ratio": 0.94},
        weight_dict={"loss_ce_sup": 1, "loss_bbox_sup": 1, "loss_align_sup": 1, "loss_iou_pred_sup": 0.5,
                     "loss_ce_unsup": 1, "loss_bbox_unsup": 1, "loss_align_unsup": 1})
    for k, v in kw.items():
        setattr(s, k, v)
    return s


def test_ocean_backbone_golden():
    from spot_amd import lib
    from spot_amd.ocean_model import ResNet50_Ocean
    lib.load()
    g = np.load(__import__("os").path.join(__import__("os").path.dirname(__file__), "golden", "g19_ocean_model.npz"))
    seed = int(g["seed"])
    net = det_fill_(ResNet50_Ocean(used_layers=[3]), seed)
    assert list(net.state_dict().keys()) == list(g["keys"])
    net = net.to(DEV)
    x = G(det_input("crop", (2, 3, 127, 127), seed, -2, 2)).requires_grad_()
    net.eval()
    with torch.no_grad():
        stages, p3 = net(x.detach())
    assert p3.shape == (2, 1024, 15, 15) and stages[2].shape == (2, 512, 15, 15) and stages[0].shape == (2, 64, 61, 61)
    _close(p3[:, ::32], g["eval_p3"], 1e-4, "eval p3")
    _close(stages[2][:, ::64], g["eval_p2"], 1e-4, "eval p2")
    _close(stages[0][:, ::8, ::4, ::4], g["eval_stem"], 1e-5, "eval stem")
    net.train()
    _, p3 = net(x)
    _close(p3[:, ::32], g["train_p3"], 2e-4, "train p3")
    (p3 * G(det_input("up", tuple(p3.shape), seed, -1, 1))).sum().backward()
    _close(x.grad[:, :, ::2, ::2], g["d_crop"], 2e-2, "d crop")        # 16 blocks of batch-statistics BatchNorm
    params = dict(net.named_parameters())
    # parameter gradients behind up to 16 blocks of batch-statistics BatchNorm on a batch of two: a ReLU that flips
    # under fp32 rounding moves single elements, so the bar is on the relative L2 error of each tensor
    for key in (k for k in g.files if k.startswith("d.")):
        got = params[key[2:]].grad
        got = got.reshape(-1)[::max(1, got.numel() // 4096)].cpu().double().numpy()
        err = np.linalg.norm(got - g[key]) / np.linalg.norm(g[key])
        assert err < 2e-2, (key, err)
    _close(net.features.layer3[0].bn2.running_mean, g["running_mean_l3"], 1e-4, "running mean")


def test_ocean_actor_decode_golden():
    from spot_amd.ocean_model import OceanSparseActor
    g = np.load(__import__("os").path.join(__import__("os").path.dirname(__file__), "golden", "g19_ocean_model.npz"))
    seed, B, S = int(g["seed"]), 6, 25
    boxes = np.concatenate([det_input("b0", (B, 2, S, S), seed, -20, 120), det_input("b1", (B, 2, S, S), seed, 100, 260)], 1)
    boxes[0, 2:] = boxes[0, :2] + 0.25
    raw = TensorDict({"pred_logits": G(det_input("lg", (B, 1, S, S), seed, -4, 4)),
                      "align_logits": G(det_input("al", (B, 1, S, S), seed, -4, 4)),
                      "pred_ious": G(det_input("io", (B, 1, S, S), seed, 0, 1)), "bbox_pred_to_img": G(boxes)})
    actor = OceanSparseActor.__new__(OceanSparseActor)
    actor.settings = types.SimpleNamespace(curriculum_settings={"cls_ratio": 0.6, "window_penalty": 0.30}, output_sz=S)
    actor.teacher_net, actor._window = (lambda teacher_data=None: raw), None
    res = actor._inference_with_teacher(teacher_data=None)
    np.testing.assert_allclose(res["best_bbox"].cpu().numpy(), g["best_bbox"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(res["best_score"].cpu().numpy(), g["best_score"], rtol=1e-6)
    assert np.array_equal(res["best_conf"].cpu().numpy(), g["best_conf"])


def _pair(prefix_t, prefix_s, n, seed):
    d = TensorDict()
    d[prefix_t + "_images"] = G(det_input(prefix_t + "_img", (n, 3, 127, 127), seed, 0, 256, np.uint8))
    d[prefix_s + "_images"] = G(det_input(prefix_s + "_img", (n, 3, 255, 255), seed, 0, 256, np.uint8))
    d[prefix_t + "_avg"] = G(det_input(prefix_t + "_avg", (n, 3), seed, 0, 255))
    d[prefix_s + "_avg"] = G(det_input(prefix_s + "_avg", (n, 3), seed, 0, 255))
    d[prefix_t + "_anno"] = G(det_input(prefix_t + "_anno", (n, 4), seed, 20, 60))
    xy, wh = det_input(prefix_s + "_xy", (n, 2), seed, 60, 110), det_input(prefix_s + "_wh", (n, 2), seed, 40, 90)
    d[prefix_s + "_anno"] = G(np.concatenate([xy, wh], -1).astype(np.float32))
    return d


def test_ocean_model_end_to_end():
    """Teacher inference -> pseudo labels; a sparse-supervised student step whose loss is finite and whose gradients
    reach backbone, neck, towers and the aligned head; the head's outputs equal the oracle's composition on the
    model's own neck features; EMA; template() / track()."""
    from spot_amd import ocean
    from spot_amd.ocean_model import OceanSparseActor, ocean_resnet50
    S = _settings()
    student = det_fill_positive_norm_(ocean_resnet50(_settings(device="cpu"), "student"), 5).to(DEV)
    teacher = det_fill_positive_norm_(ocean_resnet50(_settings(device="cpu"), "teacher"), 6).to(DEV)
    crit = ocean.ocean_loss(S)
    actor = OceanSparseActor(S, student, teacher, crit)
    teach = _pair("key", "unlabeled", 3, 1)
    pseudo = actor(mode="inference", teacher_data=teach)
    assert pseudo["best_bbox"].shape == (3, 4) and torch.isfinite(pseudo["best_bbox"]).all()
    assert ((pseudo["best_conf"] > 0) & (pseudo["best_conf"] < 1)).all() and (pseudo["best_bbox"][:, 2:] >= 1).all()
    sup, sp = _pair("template", "search", 2, 2), _pair("relay", "aug", 3, 3)
    sup["masks"] = torch.zeros(2, dtype=torch.bool, device=DEV)
    sp["masks"] = torch.ones(3, dtype=torch.bool, device=DEV)
    loss, stats = actor(mode="spsup", burnin_data=sup, sp_data=sp)
    assert torch.isfinite(loss) and set(stats) >= {"Loss/total", "Loss/ce", "Loss/bbox", "Loss/align", "Loss/iou_pred", "iou"}
    loss.backward()
    for name in ("features.features.conv1.weight", "features.features.layer3.5.conv3.weight", "neck.downsample.0.weight",
                 "connect_model.cls_encode.matrix12_k.0.weight", "connect_model.bbox_pred.weight",
                 "connect_model.iou_pred.weight", "align_head.rpn_conv.conv.weight", "align_head.rpn_cls.bias"):
        gr = dict(student.named_parameters())[name].grad
        assert gr is not None and torch.isfinite(gr).all() and float(gr.abs().max()) > 0, name
    # the head on the model's own neck features against the oracle (eval mode, running statistics)
    student.eval()
    with torch.no_grad():
        z = student.necessary_pre(sup, tag="template")[0] if sup["template_images"].dtype == torch.uint8 else sup["template_images"]
        x = student.necessary_pre(sup, tag="search")[0] if sup["search_images"].dtype == torch.uint8 else sup["search_images"]
        _, zf = student.feature_extractor(z)
        _, xf = student.feature_extractor(x)
        _, zf = student.neck(zf, crop=True)
        xf = student.neck(xf)
        assert xf.shape[-2:] == (31, 31) and zf.shape[-2:] == (7, 7)
        out = student._fusion_pipeline(xf, zf, pred_to_img_tag=True, b_iou=2)
        sd = {k: v.detach().cpu() for k, v in student.connect_model.state_dict().items()}
        bbox, cls, _, reg_feat, ious = O.ocean_box_tower(xf.cpu(), zf.cpu(), sd, False, towernum=1, batch_iou=2)
        _close(out["pred_logits"], cls.numpy(), 2e-4, "cls")
        _close(out["pred_boxes"], bbox.numpy(), 2e-4, "ltrb")
        _close(out["pred_ious"], ious.numpy(), 2e-4, "ious")
        student.template(z[:1])
        cls_t, bbox_t, align_t = student.track(x[:1])
        _close(cls_t, out["pred_logits"][:1].cpu().numpy(), 1e-5, "track cls")
        _close(align_t, out["align_logits"][:1].cpu().numpy(), 1e-4, "track align")
    before = next(teacher.parameters()).clone()
    actor(mode="update", keep_rate=0.5)
    want = 0.5 * next(student.parameters()) + 0.5 * before
    _close(next(teacher.parameters()), want.detach().cpu().numpy(), 1e-6, "ema")


def test_ocean_tracker_update_and_crop_golden():
    """Ocean_Tracker: the decode kernel + host update against the reference's update() (g20); crop windows against
    its get_subwindow_tracking; python2round; then a short sequence through the real network runs and stays finite."""
    import os
    from spot_amd.ocean_model import ocean_resnet50
    from spot_amd.ocean_tracker import Ocean_Tracker, OceanConfig
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "g20_ocean_tracker.npz"))
    S, seed = int(g["score_size"]), int(g["seed"])
    p = OceanConfig()
    assert p.score_size == S
    trk = Ocean_Tracker("t", None)
    window = np.outer(np.hanning(S), np.hanning(S))
    trk._window_dev = torch.from_numpy(window).to(DEV)
    for i in range(4):
        maps = (G(det_input("cls%d" % i, (1, 1, S, S), seed, -4, 4)), G(det_input("ltrb%d" % i, (1, 4, S, S), seed, 5, 90)),
                G(det_input("al%d" % i, (1, 1, S, S), seed, -4, 4)))
        net = types.SimpleNamespace(track=lambda x, m=maps: m)
        tpos, tsz, scale_z = np.array([100.0 + 7 * i, 80.0 - 3 * i]), np.array([40.0 + 5 * i, 60.0 - 4 * i]), 0.8 + 0.15 * i
        pos, sz, sc = trk.update(net, None, tpos, tsz * scale_z, window, scale_z, p)
        np.testing.assert_allclose(pos, g["pos%d" % i], rtol=1e-6)
        np.testing.assert_allclose(sz, g["sz%d" % i], rtol=1e-6)
        np.testing.assert_allclose(sc, g["score%d" % i], rtol=2e-6)
    im = det_input("frame", (120, 160, 3), seed, 0, 256, np.uint8)
    for i, (pos, sz) in enumerate(zip(g["sub_pos"], g["sub_sz"])):
        got = trk.get_subwindow_tracking(im, np.array(pos), int(sz), int(sz), g["avg"])
        assert np.array_equal(got[0].cpu().numpy(), g["sub%d" % i]), i
    assert [trk.python2round(v) for v in (0.5, 1.5, 2.5, 2.4, 254.5, 255.5, 100.49)] == list(g["p2round"])
    net = det_fill_positive_norm_(ocean_resnet50(_settings(device="cpu"), "teacher"), 9).to(DEV).eval()
    rng = np.random.default_rng(0)
    frames = rng.integers(0, 255, size=(8, 240, 320, 3), dtype=np.uint8)
    runs = []
    for graph in (False, True):                      # eager launches, then the captured graph from the third frame on
        trk = Ocean_Tracker("t", net, use_graph=graph)
        trk.initialize(frames[0], {"init_bbox": [120.0, 90.0, 60.0, 50.0]})
        outs = [trk.track(f) for f in frames[1:]]
        assert (trk._graph is not None) == graph
        for out in outs:
            assert np.isfinite(out["target_bbox"]).all() and out["target_bbox"][2] >= 10 and 0 <= out["best_score"] <= 1
        runs.append(np.array([o["target_bbox"] + [float(o["best_score"])] for o in outs]))
    np.testing.assert_array_equal(runs[0], runs[1])


def test_ocean_step_with_flat_gradient_buffer_matches_plain_autograd():
    """The student in flat parameter storage with gradients attached to one flat buffer (kernels add weight
    gradients straight into it): the same sparse-supervised step gives the same loss and the same gradients as
    plain autograd, and one fused AdamW step moves every trainable parameter."""
    from spot_amd import ocean
    from spot_amd.arena import ParamArena
    from spot_amd.ocean_model import OceanSparseActor, ocean_resnet50
    from spot_amd.optim import FlatAdamW
    S = _settings()
    crit = ocean.ocean_loss(S)
    sup, sp = _pair("template", "search", 2, 12), _pair("relay", "aug", 2, 13)
    sup["masks"] = torch.zeros(2, dtype=torch.bool, device=DEV)
    sp["masks"] = torch.ones(2, dtype=torch.bool, device=DEV)
    results = []
    for flat in (False, True):
        student = det_fill_positive_norm_(ocean_resnet50(_settings(device="cpu"), "student"), 5).to(DEV).train()
        teacher = det_fill_positive_norm_(ocean_resnet50(_settings(device="cpu"), "teacher"), 6).to(DEV)
        arena = None
        if flat:
            arena = ParamArena(student)
            student._spot_arena = arena
            arena.attach_grads()
            arena.zero_grad()
        actor = OceanSparseActor(S, student, teacher, crit)
        loss, _ = actor(mode="spsup", burnin_data=TensorDict(dict(sup)), sp_data=TensorDict(dict(sp)))
        loss.backward()
        grads = {n: p.grad.detach().clone() for n, p in student.named_parameters() if p.grad is not None}
        results.append((float(loss.detach()), grads))
        if flat:
            opt = FlatAdamW([{"params": [p for p in student.parameters() if p.requires_grad]}], arena, lr=1e-3, weight_decay=1e-4)
            before = arena.flat.clone()
            opt.step()
            moved = (arena.flat != before).float().mean().item()
            assert moved > 0.9, moved
    (l0, g0), (l1, g1) = results
    assert abs(l0 - l1) <= 1e-6 * abs(l0)
    assert set(g0) == set(g1)
    gmax = max(float(g.abs().max()) for g in g0.values())
    for n in g0:                 # gradients that are zero up to rounding (a bias in front of batch statistics) get the global floor
        scale = float(g0[n].abs().max())
        assert float((g0[n] - g1[n]).abs().max()) <= 2e-5 * scale + 1e-7 * gmax, n


def test_ocean_trainer_iteration_end_to_end():
    """LTRSparseTrainer with the Siamese variant: teacher pseudo labels -> middleware filter -> compaction -> student
    step with gradients in the flat buffer -> fused AdamW, then an EMA update; nothing but the kept count and the
    loss's finiteness is read by the host."""
    from spot_amd import ocean
    from spot_amd.middleware import SparseMiddleware
    from spot_amd.ocean_model import OceanSparseActor, ocean_resnet50
    from spot_amd.optim import FlatAdamW
    from spot_amd.synthetic import sup_batch, unsup_batch
    from spot_amd.trainer import LTRSparseTrainer
    S = _settings(batch_size_sup=2, batch_size_unsup=4, teacher_update_frequency=2, keep_rate=0.9)
    S.curriculum_settings["confidence_threshold"] = 0.0
    S.batch_size = 6
    student = det_fill_positive_norm_(ocean_resnet50(_settings(device="cpu"), "student"), 5).to(DEV)
    teacher = det_fill_positive_norm_(ocean_resnet50(_settings(device="cpu"), "teacher"), 5).to(DEV)
    actor = OceanSparseActor(S, student, teacher, ocean.ocean_loss(S))
    trainer = LTRSparseTrainer(actor, None, None, S, SparseMiddleware(S))
    trainer.arena.attach_grads()
    trainer.optimizer = FlatAdamW([{"params": [p for p in student.parameters() if p.requires_grad]}], trainer.arena,
                                  lr=1e-4, weight_decay=1e-4)
    w0 = student.connect_model.bbox_pred.weight.detach().clone()
    t0 = next(teacher.parameters()).detach().clone()
    for i in range(2):
        data_unsup, mid = unsup_batch(4, 3 + i, temp_sz=127, search_sz=255, device=DEV)
        stats = trainer.train_iteration(i, sup_batch(2, 3 + i, temp_sz=127, search_sz=255, device=DEV), data_unsup, mid)
        assert stats is not None and torch.isfinite(stats["Loss/total"])
    assert not torch.equal(w0, student.connect_model.bbox_pred.weight.detach())
    assert not torch.equal(t0, next(teacher.parameters()).detach())          # iteration 1 began with an EMA update


def test_ocean_two_stream_features_match_sequential():
    """Template pass on a second stream beside the search pass (the search pass's running-statistics updates applied
    after the join): same outputs, same BatchNorm running buffers and same gradients as one stream."""
    from spot_amd import fusion, ocean
    from spot_amd.ocean_model import OceanSparseActor, ocean_resnet50
    S = _settings()
    crit = ocean.ocean_loss(S)
    sup, sp = _pair("template", "search", 2, 21), _pair("relay", "aug", 3, 22)
    sup["masks"] = torch.zeros(2, dtype=torch.bool, device=DEV)
    sp["masks"] = torch.ones(3, dtype=torch.bool, device=DEV)
    saved, runs = fusion.TWO_STREAMS, []
    try:
        for two in (False, True):
            fusion.TWO_STREAMS = two
            student = det_fill_positive_norm_(ocean_resnet50(_settings(device="cpu"), "student"), 5).to(DEV).train()
            actor = OceanSparseActor(S, student, student, crit)
            loss, _ = actor(mode="spsup", burnin_data=TensorDict(dict(sup)), sp_data=TensorDict(dict(sp)))
            loss.backward()
            torch.cuda.synchronize()
            runs.append((loss.detach().clone(), {n: p.grad.clone() for n, p in student.named_parameters() if p.grad is not None},
                         {n: b.clone() for n, b in student.named_buffers() if "running" in n or "tracked" in n}))
    finally:
        fusion.TWO_STREAMS = saved
    (l0, g0, b0), (l1, g1, b1) = runs
    assert torch.equal(l0, l1)
    for n in b0:
        assert torch.equal(b0[n], b1[n]), n
    gmax = max(float(g.abs().max()) for g in g0.values())
    for n in g0:          # weight gradients are summed by float atomics (split-K): the order of the addends varies
        assert float((g0[n] - g1[n]).abs().max()) <= 5e-5 * float(g0[n].abs().max()) + 1e-6 * gmax, n


def test_frozen_weight_tables_are_forgotten_per_network():
    """Tables derived from a network's weights go when that network says its weights changed, also after its
    parameters moved into flat storage; another network's tables stay (a captured graph may still read them)."""
    from spot_amd import ops
    from spot_amd.arena import ParamArena
    a = torch.nn.Conv2d(32, 32, 3, padding=1, bias=False).to(DEV)
    b = torch.nn.Conv2d(32, 32, 3, padding=1, bias=False).to(DEV)
    bn = torch.nn.BatchNorm2d(32).to(DEV).eval()
    x = torch.randn(2, 32, 15, 15, device=DEV)
    ops.frozen_weights.clear()
    with torch.no_grad(), ops.frozen_weights():
        ya = ops.conv_bn_act(x, a, bn, relu=True)
        ops.conv_bn_act(x, b, bn, relu=True)
    assert len(ops.frozen_weights._tables) == 2
    ParamArena(a)                                    # the parameter's storage moves
    ops.frozen_weights.forget(a)
    assert len(ops.frozen_weights._tables) == 1
    with torch.no_grad():
        a.weight.mul_(2.0)
    with torch.no_grad(), ops.frozen_weights():
        ya2 = ops.conv_bn_act(x, a, bn, relu=False)
        ref = ops.batch_norm_act(ops.conv_general(x, a), bn)
    np.testing.assert_allclose(ya2.cpu().numpy(), ref.cpu().numpy(), rtol=0, atol=1e-5 * float(ref.abs().max()))
    ops.frozen_weights.forget(b)
    ops.frozen_weights.forget(a)
    assert len(ops.frozen_weights._tables) == 0 and ya.shape == ya2.shape
