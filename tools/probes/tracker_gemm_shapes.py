"""Per-shape GEMM / attention launch times of one tracker frame (eager, HIP events around every launch)."""
import os
import sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from spot_amd import ops
from spot_amd.detinit import det_fill_
from spot_amd.synthetic import Settings
from spot_amd.tracker import NetWithBackbone, TransT_Tracker
from spot_amd.transt import transt_resnet50

net = NetWithBackbone(net=det_fill_(transt_resnet50(Settings(device="cpu")), 1).cuda(), use_gpu=True)
trk = TransT_Tracker("t", net, 0.49, 128, 256, use_graph=False)
rng = np.random.default_rng(0)
frames = rng.integers(0, 255, size=(12, 256, 256, 3), dtype=np.uint8)
trk.initialize(frames[0], {"init_bbox": [96.0, 96.0, 64.0, 64.0]})
for f in frames[1:4]:
    trk.track(f)
torch.cuda.synchronize()
timer = ops.TIMER = ops.KernelTimer(shapes=True)
n = 0
for f in frames[4:]:
    trk.track(f)
    n += 1
torch.cuda.synchronize()
ops.TIMER = None
rows = sorted(((v["ms"] / n * 1e3, k, v["launches"] / n, v["work"] / (v["ms"] * 1e-3) / 1e12) for k, v in timer.summary().items()),
              reverse=True)
print("us/frame  launches/frame  TF   key   (total %.1f us/frame)" % sum(r[0] for r in rows))
for us, k, ln, tf in rows:
    print("%8.1f  %5.1f  %6.1f  %s" % (us, ln, tf, k))
