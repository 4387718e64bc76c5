#!/usr/bin/env python3
"""Throughput of the configurations that are parity-test cases rather than the bench line, on one MI355X.

    python tools/measure_configs.py [cfg1] [cfg3] [cfg5]

cfg1: online tracker API, B=1, 50 synthetic 256x256 frames (template cached once, track() per frame).
cfg3: teacher pseudo-labelling + decode + IoU-confidence filter only, one rank's 128-clip shard of the 1024-clip
      job (ranks own contiguous blocks and never communicate, so the 8-GPU figure is 8 shards side by side).
cfg5: GroupDW depthwise-correlation head, 384^2 search (45x45 / 43x45 / 45x43 branches), B=128, C=256,
      forward and forward+backward.
Each prints one JSON line.
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from spot_amd import lib, ops  # noqa: E402
from spot_amd.detinit import det_fill_  # noqa: E402
from spot_amd.synthetic import Settings, unsup_batch  # noqa: E402
from spot_amd.transt import transt_resnet50  # noqa: E402


def cfg1():
    from spot_amd.tracker import NetWithBackbone, TransT_Tracker
    net = NetWithBackbone(net=det_fill_(transt_resnet50(Settings(device="cpu")), 1).cuda(), use_gpu=True)
    trk = TransT_Tracker("t", net, 0.49, 128, 256)
    rng = np.random.default_rng(0)
    frames = rng.integers(0, 255, size=(50, 256, 256, 3), dtype=np.uint8)
    for y in range(50):                                   # a drifting bright square
        frames[y, 96 + y // 4:160 + y // 4, 96 + y // 5:160 + y // 5] = 230
    for rep in range(2):                                  # first pass warms the convolution library
        trk.initialize(frames[0], {"init_bbox": [96.0, 96.0, 64.0, 64.0]})
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for f in frames[1:]:
            trk.track(f)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
    print(json.dumps({"config": "cfg1: tracker API, B=1, 49 track() calls on 256x256 frames incl. " +
                                "frame upload, device crop + resize" +
                                ", network, device decode, D2H", "value": round(49 / el, 2), "unit": "frames/s",
                      "ms_per_frame": round(el / 49 * 1e3, 3), "graph_replay": trk._graph is not None}))


def cfg3():
    from spot_amd.actor import TranstSparseActor
    from spot_amd.middleware import SparseMiddleware
    from spot_amd.tensordict import TensorDict
    S = Settings(device="cuda")
    net = det_fill_(transt_resnet50(S), 0).cuda().eval()
    actor, mw = TranstSparseActor(S, net, net, None), SparseMiddleware(S)
    n = 128
    data, mid = unsup_batch(n, 2, device="cuda")
    mid = {k: v.cuda() for k, v in mid.items()}

    def run():
        with torch.no_grad():
            d = TensorDict({k: v for k, v in data.items() if k.startswith(("key_", "unlabeled_"))})
            pseudo = actor(mode="inference", teacher_data=d)
            return mw.record_relay_results(mid, pseudo)

    # random weights give arbitrary IoU-head confidences: shift the head's last bias once so that about three of
    # four clips pass the threshold, as with a trained teacher (otherwise this times the all-rejected case)
    with torch.no_grad():
        d = TensorDict({k: v for k, v in data.items() if k.startswith(("key_", "unlabeled_"))})
        conf = actor(mode="inference", teacher_data=d)["best_conf"].clamp(1e-6, 1 - 1e-6)
        logit = torch.log(conf / (1 - conf))
        net.iou_embed.layers[2].bias += float(torch.log(torch.tensor(0.85 / 0.15))) - torch.quantile(logit, 0.25)
    for _ in range(2):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        out = run()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(json.dumps({"config": "cfg3: teacher pseudo-label + decode + filter, one rank's 128-clip shard (of 1024 over 8 "
                                "GPUs), fp32, eval mode", "value": round(reps * n / el, 1), "unit": "clips/s per GPU",
                      "ms_per_shard": round(el / reps * 1e3, 2), "kept": int(out["aug_success"].sum())}))


def cfg5():
    B, Cn, hx, hz = 128, 256, 45, 5
    r = lambda *s: torch.randn(*s, device="cuda")
    xs = [r(B, Cn, hx, hx), r(B, Cn, hx - 2, hx), r(B, Cn, hx, hx - 2)]
    zs = [r(B, Cn, hz, hz), r(B, Cn, hz - 2, hz), r(B, Cn, hz, hz - 2)]
    w = r(3)
    ho = hx - hz + 1
    fwd_bytes = 4.0 * (sum(x.numel() for x in xs) + sum(z.numel() for z in zs) + B * Cn * ho * ho)

    from tools.microbench import timeit          # clocks brought up first, median of five timed runs

    tf = timeit(lambda: ops.groupdw(zs, xs, w))
    for t in xs + zs + [w]:
        t.requires_grad_()
    up = r(B, Cn, ho, ho)

    def fb():
        for t in xs + zs + [w]:
            t.grad = None
        (ops.groupdw(zs, xs, w) * up).sum().backward()

    tfb = timeit(fb, 10)
    print(json.dumps({"config": "cfg5: GroupDW head, 384^2 search, B=128, C=256, fp32",
                      "fwd_us": round(tf * 1e6, 1), "fwd_GBps": round(fwd_bytes / tf / 1e9, 1),
                      "fwd_frac_of_8TBps": round(fwd_bytes / tf / 8e12, 3),
                      "fwd_clips_per_s": round(B / tf, 0), "fwd_bwd_us": round(tfb * 1e6, 1),
                      "fwd_bwd_clips_per_s": round(B / tfb, 0), "algorithmic_fwd_bytes": fwd_bytes}))


if __name__ == "__main__":
    lib.load()
    which = sys.argv[1:] or ["cfg1", "cfg3", "cfg5"]
    for name in which:
        {"cfg1": cfg1, "cfg3": cfg3, "cfg5": cfg5}[name]()
