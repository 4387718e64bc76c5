"""Soak run of the bench step: a few hundred iterations; prints device memory in use and the loss at intervals.
A leak (buffers kept from step to step) or a drifting step time shows up here, not in a 10-step bench."""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from spot_amd.tensordict import TensorDict  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--every", type=int, default=50)
    a = ap.parse_args()
    args = argparse.Namespace(dropout=0.1, sup=8, clips=32, attention="f32", products="f32")
    S, trainer, data_sup, data_unsup, mid = bench.build_job(args, 0)
    unsup_view = TensorDict(data_unsup)
    t0 = time.perf_counter()
    last = None
    for i in range(a.steps):
        stats = trainer.train_iteration(i, TensorDict(data_sup), unsup_view, mid, next_unsup=unsup_view)
        last = stats if stats is not None else last
        if (i + 1) % a.every == 0:
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            t0 = time.perf_counter()
            loss = None if last is None else float(last.get("Loss/total", float("nan")))
            print("step %4d  %.2f ms/step  allocated %.1f MB  reserved %.1f MB  peak %.1f MB  skipped %d  loss %s" % (
                i + 1, el / a.every * 1e3, torch.cuda.memory_allocated() / 2 ** 20, torch.cuda.memory_reserved() / 2 ** 20,
                torch.cuda.max_memory_allocated() / 2 ** 20, trainer.skipped, loss), flush=True)


if __name__ == "__main__":
    main()
