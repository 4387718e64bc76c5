#!/usr/bin/env python3
"""Times the Siamese (SP-Ocean) head on one MI355X: neck features -> box_tower -> box decoding -> aligned head, forward
and forward + backward, at the 255-px (25x25 response) and 384-px (41x41 response) search sizes.  Beside it the same
tower built from the framework's stock modules (the convolution library, grouped convolution for the correlation),
which is what the reference executes; the aligned head has no such counterpart (its extension does not exist here).

    python tools/ocean_bench.py [--batch 128] [--search 384] [--steps 5]
prints one JSON line per configuration."""
import argparse
import json
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spot_amd import lib, ocean  # noqa: E402
from spot_amd.detinit import det_fill_positive_norm_  # noqa: E402


def stock_tower(t, search, kernel, batch_iou):
    """box_tower.forward with the modules called as they are (ocean_fusion.py:87-114)."""
    def enc(m, z, x):
        return ([getattr(m, n + "_k")(z) for n in ("matrix11", "matrix12", "matrix21")],
                [getattr(m, n + "_s")(x) for n in ("matrix11", "matrix12", "matrix21")])

    def xc(x, k):
        b, c = k.shape[:2]
        o = F.conv2d(x.reshape(1, b * c, *x.shape[2:]), k.reshape(b * c, 1, *k.shape[2:]), groups=b * c)
        return o.reshape(b, c, *o.shape[2:])

    def dw(w, zs, xs):
        sw = F.softmax(w, 0)
        return sum(sw[i] * xc(xs[i], zs[i]) for i in range(3))
    cz, cx = enc(t.cls_encode, kernel, search)
    rz, rx = enc(t.reg_encode, kernel, search)
    cls_dw, reg_dw = dw(t.cls_dw.weight, cz, cx), dw(t.reg_dw.weight, rz, rx)
    x_reg = t.bbox_tower(reg_dw)
    x = torch.exp(torch.clamp_max(t.adjust * t.bbox_pred(x_reg) + t.bias, 4.7))
    cls = 0.1 * t.cls_pred(t.cls_tower(cls_dw))
    ious = t.iou_pred(t.iou_pred_tower(reg_dw.detach()[:batch_iou])).sigmoid()
    return x, cls, x_reg, ious


def timed(fn, steps, warmup=2):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps


def variant(steps):
    """The assembled SP-Ocean variant with the reference's settings (spot_ocean_003.py: 4-layer towers, 255 / 127 crops):
    online tracker rate, and one sparse-supervised iteration (teacher pseudo-labelling of 24 clips, student step on
    8 + 24 pairs with in-network augmentation, backward)."""
    import time
    import types
    import numpy as np
    from spot_amd.ocean_model import OceanSparseActor, ocean_resnet50
    from spot_amd.ocean_tracker import Ocean_Tracker
    from spot_amd.tensordict import TensorDict
    S = types.SimpleNamespace(
        device="cuda", used_layers=[3], before_neck_dim=1024, head_dim=256, head_tower_num=4, iou_prediction_tag=True,
        strong_augmentation_tag=True, aug_settings={"gray": 0.05, "jitter": 0.2, "angle": 10, "lr_f_prob": 0.35, "ud_f_prob": 0.1},
        normalize_mean=[0.485, 0.456, 0.406], normalize_std=[0.229, 0.224, 0.225], search_sz=255, temp_sz=127, output_sz=25,
        batch_size=32, stride=8, search_feature_sz=25,
        curriculum_settings={"confidence_threshold": 0.82, "window_penalty": 0.30, "cls_ratio": 0.94},
        weight_dict={"loss_ce_sup": 1, "loss_bbox_sup": 1, "loss_align_sup": 1, "loss_iou_pred_sup": 0.5,
                     "loss_ce_unsup": 1, "loss_bbox_unsup": 1, "loss_align_unsup": 1})
    cpu = types.SimpleNamespace(**{**vars(S), "device": "cpu"})
    student = det_fill_positive_norm_(ocean_resnet50(cpu, "student"), 1).cuda()
    teacher = det_fill_positive_norm_(ocean_resnet50(cpu, "teacher"), 2).cuda()
    rng = np.random.default_rng(0)
    # ---- tracker
    trk = Ocean_Tracker("t", teacher.eval())
    frames = rng.integers(0, 255, size=(50, 256, 256, 3), dtype=np.uint8)
    for rep in range(2):
        trk.initialize(frames[0], {"init_bbox": [96.0, 96.0, 64.0, 64.0]})
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for f in frames[1:]:
            trk.track(f)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
    print(json.dumps({"config": "SP-Ocean tracker API, B=1, 49 track() calls on 256x256 frames (frame upload, device crop, "
                                "network, decode kernel, one D2H)", "value": round(49 / el, 2),
                      "unit": "frames/s", "ms_per_frame": round(el / 49 * 1e3, 3), "graph_replay": trk._graph is not None}))
    # ---- one sparse-supervised iteration
    teacher.train(), student.train()
    actor = OceanSparseActor(S, student, teacher, ocean.ocean_loss(S))

    def pair(t, s, n):
        d = TensorDict()
        d[t + "_images"] = torch.from_numpy(rng.integers(0, 255, (n, 3, 127, 127), dtype=np.uint8)).cuda()
        d[s + "_images"] = torch.from_numpy(rng.integers(0, 255, (n, 3, 255, 255), dtype=np.uint8)).cuda()
        d[t + "_avg"], d[s + "_avg"] = torch.rand(n, 3).cuda() * 255, torch.rand(n, 3).cuda() * 255
        d[t + "_anno"] = torch.tensor([[40.0, 40.0, 50.0, 45.0]]).repeat(n, 1).cuda()
        d[s + "_anno"] = torch.cat([torch.rand(n, 2) * 60 + 60, torch.rand(n, 2) * 50 + 40], 1).cuda()
        return d
    teach, sup, sp = pair("key", "unlabeled", 24), pair("template", "search", 8), pair("relay", "aug", 24)
    sup["masks"] = torch.zeros(8, dtype=torch.bool, device="cuda")
    sp["masks"] = torch.ones(24, dtype=torch.bool, device="cuda")
    params = list(student.parameters())

    def iteration():
        pseudo = actor(mode="inference", teacher_data=TensorDict(dict(teach)))
        spd = TensorDict(dict(sp))
        spd["aug_anno"] = pseudo["best_bbox"]
        for p_ in params:
            p_.grad = None
        loss, _ = actor(mode="spsup", burnin_data=TensorDict(dict(sup)), sp_data=spd)
        loss.backward()
    ms = timed(iteration, steps)
    print(json.dumps({"config": "SP-Ocean iteration: teacher pseudo-labels on 24 clips (train-mode teacher), student "
                                "step on 8 + 24 pairs with in-network augmentation, forward + backward, fp32, eager",
                      "ms_per_iteration": ms, "value": round(32 / ms * 1e3, 1), "unit": "clips/s",
                      "peak_mem_GB": torch.cuda.max_memory_allocated() / 2 ** 30}))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "variant":
        lib.load()
        return variant(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--search", type=int, default=384, choices=(255, 384))
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--channels", type=int, default=256)
    ap.add_argument("--skip-stock", action="store_true")
    a = ap.parse_args()
    lib.load()
    dev, Cn, B = "cuda", a.channels, a.batch
    hx = 31 if a.search == 255 else 47                  # neck map of the search patch; response = hx - 6
    S = hx - 6
    tower = det_fill_positive_norm_(ocean.box_tower(Cn, Cn, towernum=1, iou_prediction=True), 1).to(dev).train()
    head = det_fill_positive_norm_(ocean.AlignHead(Cn, Cn), 2).to(dev).train()
    g = torch.Generator().manual_seed(0)
    xf = (torch.rand(B, Cn, hx, hx, generator=g) * 2 - 1).to(dev)
    zf = (torch.rand(B, Cn, 7, 7, generator=g) * 2 - 1).to(dev)
    gx, gy = (torch.tensor(v, dtype=torch.float32, device=dev) for v in ocean.response_grid(S, a.search))
    params = [p for p in list(tower.parameters()) + list(head.parameters())]

    def ours(backward):
        x, z = xf.clone().requires_grad_(backward), zf.clone().requires_grad_(backward)
        with torch.set_grad_enabled(backward):
            out = ocean.fusion_pipeline(tower, head, x, z, gx, gy, a.search, iou_prediction=True, b_iou=B)
            if backward:
                for p in params:
                    p.grad = None
                (out["pred_logits"].sum() + out["pred_boxes"].log().sum() + out["align_logits"].sum()
                 + out["pred_ious"].sum()).backward()

    def stock(backward):
        x, z = xf.clone().requires_grad_(backward), zf.clone().requires_grad_(backward)
        with torch.set_grad_enabled(backward):
            bbox, cls, _, ious = stock_tower(tower, x, z, B)
            if backward:
                for p in params:
                    p.grad = None
                (cls.sum() + bbox.log().sum() + ious.sum()).backward()

    def ours_tower_only(backward):
        x, z = xf.clone().requires_grad_(backward), zf.clone().requires_grad_(backward)
        with torch.set_grad_enabled(backward):
            bbox, cls, _, _, extra = tower(x, z, iou_prediction=True, batch_iou=B)
            if backward:
                for p in params:
                    p.grad = None
                (cls.sum() + bbox.log().sum() + extra["pred_ious"].sum()).backward()

    res = {"workload": "SP-Ocean head, search %d (response %dx%d), B=%d, C=%d, fp32, train-mode BatchNorm" % (a.search, S, S, B, Cn),
           "ms_fwd": timed(lambda: ours(False), a.steps), "ms_fwd_bwd": timed(lambda: ours(True), a.steps),
           "ms_tower_fwd": timed(lambda: ours_tower_only(False), a.steps),
           "ms_tower_fwd_bwd": timed(lambda: ours_tower_only(True), a.steps)}
    if not a.skip_stock:
        res["ms_tower_fwd_stock_modules"] = timed(lambda: stock(False), a.steps)
        res["ms_tower_fwd_bwd_stock_modules"] = timed(lambda: stock(True), a.steps)
    res["clips_per_s_fwd_bwd"] = B / res["ms_fwd_bwd"] * 1e3
    res["peak_mem_GB"] = torch.cuda.max_memory_allocated() / 2 ** 30
    print(json.dumps(res))


if __name__ == "__main__":
    main()
