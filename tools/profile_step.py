#!/usr/bin/env python3
"""Host-side view of one bench step: where the CPU spends its enqueue time and where it waits.

    python tools/profile_step.py [--steps 5] [--cprofile]
Prints wall-clock per phase (enqueue only, then the wait for the GPU) and optionally a cProfile table.
"""
import argparse
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--clips", type=int, default=32)
    ap.add_argument("--sup", type=int, default=8)
    ap.add_argument("--dropout", type=float, default=0.1)
    ap.add_argument("--cprofile", action="store_true")
    ap.add_argument("--attention", default="f32")
    args = ap.parse_args()
    from spot_amd import lib
    from spot_amd.tensordict import TensorDict
    lib.load()
    torch.cuda.set_device(0)
    S, trainer, data_sup, data_unsup, mid = bench.build_job(args, 0)
    for i in range(3):
        trainer.train_iteration(i, TensorDict(data_sup), TensorDict(data_unsup), mid)
    torch.cuda.synchronize()
    sync = torch.cuda.synchronize
    now = time.perf_counter
    rows = []
    for i in range(args.steps):
        du, ds = TensorDict(data_unsup), TensorDict(data_sup)
        sync()
        t0 = now()
        pseudo = trainer._pseudo_label_generation(du)
        aug = trainer._middleware_ops(mid, pseudo)
        t1 = now()
        sync()
        t2 = now()
        loss, stats, n_keep = trainer._sparse_training(ds, du, aug, True)
        t3 = now()
        sync()
        t4 = now()
        trainer.arena.zero_grad()
        trainer.reducer.arm()
        loss.backward()
        trainer.reducer.finish()
        t5 = now()
        sync()
        t6 = now()
        trainer.optimizer.step()
        t7 = now()
        sync()
        t8 = now()
        rows.append([t1 - t0, t2 - t0, t3 - t2, t4 - t2, t5 - t4, t6 - t4, t7 - t6, t8 - t6])
    names = ["teacher enq", "teacher total", "student fwd enq", "student fwd total", "backward enq", "backward total",
             "adamw enq", "adamw total"]
    print("phase (ms, mean over %d steps)" % args.steps)
    for j, n in enumerate(names):
        print("  %-20s %8.2f" % (n, 1e3 * sum(r[j] for r in rows) / len(rows)))
    print("  sum of totals        %8.2f" % (1e3 * sum(r[1] + r[3] + r[5] + r[7] for r in rows) / len(rows)))
    if args.cprofile:
        pr = cProfile.Profile()
        pr.enable()
        for i in range(args.steps):
            trainer.train_iteration(3 + i, TensorDict(data_sup), TensorDict(data_unsup), mid)
        torch.cuda.synchronize()
        pr.disable()
        st = pstats.Stats(pr, stream=sys.stdout)
        st.sort_stats("tottime").print_stats(45)


if __name__ == "__main__":
    main()
