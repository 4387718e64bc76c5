#!/usr/bin/env python3
"""Host-side cost of a bench step: cProfile over a few steady steps (the device is not waited for inside the
profiled region except where the step itself reads from it).

    python tools/host_profile.py [--steps 4]
"""
import argparse
import cProfile
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--clips", type=int, default=32)
    ap.add_argument("--sup", type=int, default=8)
    ap.add_argument("--dropout", type=float, default=0.1)
    ap.add_argument("--attention", default="f32")
    ap.add_argument("--products", default="f32")
    ap.add_argument("--sort", default="tottime")
    args = ap.parse_args()
    from spot_amd import lib
    from spot_amd.tensordict import TensorDict
    lib.load()
    torch.cuda.set_device(0)
    S, trainer, data_sup, data_unsup, mid = bench.build_job(args, 0)
    for i in range(3):
        trainer.train_iteration(i, TensorDict(data_sup), TensorDict(data_unsup), mid, next_unsup=TensorDict(data_unsup))
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for i in range(args.steps):
        trainer.train_iteration(10 + i, TensorDict(data_sup), TensorDict(data_unsup), mid, next_unsup=TensorDict(data_unsup))
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats(args.sort).print_stats(45)


if __name__ == "__main__":
    main()
