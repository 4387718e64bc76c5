#!/usr/bin/env python3
"""Per-shape table of the GEMM / attention launches of one bench step, each timed with the chip to itself
(second stream off, no look-ahead teacher pass).

    python tools/gemm_shapes.py [--steps 2]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--clips", type=int, default=32)
    ap.add_argument("--sup", type=int, default=8)
    ap.add_argument("--dropout", type=float, default=0.1)
    ap.add_argument("--attention", default="f32")
    ap.add_argument("--products", default="f32")
    args = ap.parse_args()
    from spot_amd import fusion, lib, ops
    from spot_amd.tensordict import TensorDict
    lib.load()
    torch.cuda.set_device(0)
    fusion.TWO_STREAMS = False
    S, trainer, data_sup, data_unsup, mid = bench.build_job(args, 0)
    for i in range(3):
        trainer.train_iteration(i, TensorDict(data_sup), TensorDict(data_unsup), mid)
    torch.cuda.synchronize()
    timer = ops.TIMER = ops.KernelTimer(shapes=True)
    for i in range(args.steps):
        trainer.train_iteration(3 + i, TensorDict(data_sup), TensorDict(data_unsup), mid)
    agg = timer.summary()
    ops.TIMER = None
    total = sum(a["ms"] for a in agg.values()) / args.steps
    print("timed launches: %.2f ms per step" % total)
    print("%-58s %6s %9s %9s %7s" % ("kernel / shape", "n/step", "us each", "ms/step", "TF"))
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1]["ms"]):
        n = a["launches"] / args.steps
        print("%-58s %6.1f %9.1f %9.3f %7.1f" % (k, n, a["ms"] * 1e3 / a["launches"], a["ms"] / args.steps,
                                                  a["work"] / (a["ms"] * 1e-3) / 1e12))


if __name__ == "__main__":
    main()
