#!/usr/bin/env python3
"""Times the convolution library on the ResNet-50 3x3 / 7x7 shapes of one student pass (B=40), forward,
input gradient and weight gradient separately, to see what a hand-written kernel would have to beat."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from torch.nn.grad import conv2d_input, conv2d_weight

from tools.microbench import timeit
from spot_amd import lib, ops

B = int(os.environ.get("SPOT_CB_BATCH", "40"))
shapes = []   # name, Cin, Cout, H(in), k, stride, pad, dil, count per stream pass
for tag, S in (("search", 256), ("template", 128)):
    shapes += [(tag + " stem7x7", 3, 64, S, 7, 2, 3, 1, 1),
               (tag + " l1 3x3", 64, 64, S // 4, 3, 1, 1, 1, 3),
               (tag + " l2.0 3x3 s2", 128, 128, S // 4, 3, 2, 1, 1, 1),
               (tag + " l2 3x3", 128, 128, S // 8, 3, 1, 1, 1, 3),
               (tag + " l3.0 3x3 d2", 256, 256, S // 8, 3, 1, 2, 2, 1),
               (tag + " l3 3x3", 256, 256, S // 8, 3, 1, 1, 1, 5)]
tot = [0.0, 0.0, 0.0]
totw = [0.0, 0.0, 0.0]
for name, ci, co, h, k, s, p, d, cnt in shapes:
    x = torch.randn(B, ci, h, h, device="cuda")
    w = torch.randn(co, ci, k, k, device="cuda")
    y = F.conv2d(x, w, None, s, p, d)
    dy = torch.randn_like(y)
    fl = 2.0 * y.numel() * ci * k * k
    t0 = timeit(lambda: F.conv2d(x, w, None, s, p, d), 10)
    t1 = timeit(lambda: conv2d_input(x.shape, w, dy, s, p, d), 10) if ci > 3 else 0.0
    t2 = timeit(lambda: conv2d_weight(x, w.shape, dy, s, p, d), 10)
    tw = tg = 0.0
    if k == 3 and s == 1:
        tw = timeit(lambda: ops._winograd(x, w, co, False, d), 10)
        tg = timeit(lambda: ops._winograd(dy, w, ci, True, d), 10)
        dwb = torch.empty_like(w)
        wsb = torch.empty((lib.load().spot_conv3x3_winograd_wgrad_ws_floats(B, ci, co, h, h),), device="cuda")
        tww = timeit(lambda: lib.call("spot_conv3x3_winograd_wgrad_f32", ops._p(x), None, ops._p(dy), ops._p(dwb), ops._p(wsb),
                                      B, ci, co, h, h, d, 0, ops._st()), 10)
        totw[0] += tw * cnt
        totw[1] += tg * cnt
        totw[2] += tww * cnt
    else:
        tww = 0.0
        totw[2] += t2 * cnt
        totw[0] += t0 * cnt
        totw[1] += t1 * cnt
    for i, t in enumerate((t0, t1, t2)):
        tot[i] += t * cnt
    print("%-24s x%d  fwd %7.1f us %6.1f TF | dgrad %7.1f us %6.1f TF | wgrad %7.1f us %6.1f TF" % (
        name, cnt, t0 * 1e6, fl / t0 / 1e12, t1 * 1e6, fl / t1 / 1e12 if t1 else 0, t2 * 1e6, fl / t2 / 1e12) +
          (" | winograd fwd %7.1f us, dgrad %7.1f us, wgrad %7.1f us" % (tw * 1e6, tg * 1e6, tww * 1e6) if tw else ""))
print("per pass over both streams: fwd %.2f ms, dgrad %.2f ms, wgrad %.2f ms" % tuple(t * 1e3 for t in tot))
print("with the Winograd path where it applies: fwd %.2f ms, dgrad %.2f ms, wgrad %.2f ms" % tuple(t * 1e3 for t in totw))
