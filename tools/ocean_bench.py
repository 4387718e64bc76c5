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


def main():
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
