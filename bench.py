#!/usr/bin/env python3
"""Benchmark of the SPOT tracker hot path on MI355X.

One step = one iteration of the sparsely-supervised loop on a synthetic batch that is already
resident in HBM: (teacher EMA every 5th step) -> teacher pseudo-labelling of 32 clips (backbone,
fusion, heads) -> decode -> IoU-confidence filter -> keep/refill compaction -> student forward on
8 labelled + 32 pseudo-labelled pairs -> losses -> backward -> (RCCL gradient all-reduce) -> AdamW.
value = clips pseudo-labelled AND trained on per second, whole job (all ranks).

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
    python bench.py --gpus N ...        (no launcher: starts the N ranks itself, one fresh process per GPU)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X dense fp32 matrix peak
PEAK_HBM_GBS = 8000.0


def build_world(args):
    """This process's place in the job.  A rank launched for --gpus N whose device does not exist says how many GPUs
    the job needs; a world size that is not --gpus N is a launcher mistake and says so."""
    from spot_amd import dist as sdist
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if world_env != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s) (WORLD_SIZE); start it as `python "
                         "bench.py --gpus %d` or under torchrun with --nproc-per-node %d"
                         % (args.gpus, world_env, args.gpus, args.gpus))
    have = torch.cuda.device_count()
    need = 1 if args.rehearse_on_one_gpu else args.gpus
    if have < need:
        raise SystemExit("bench.py: --gpus %d needs %d GPUs on this node, found %d (rank %s)"
                         % (args.gpus, need, have, os.environ.get("RANK", "0")))
    if args.rehearse_on_one_gpu:
        os.environ["LOCAL_RANK"] = "0"          # every rank on GPU 0; RCCL refuses that, so the group runs over gloo
    rank, world, local = sdist.init_from_env("gloo" if args.rehearse_on_one_gpu else "nccl")
    torch.cuda.set_device(local)
    return rank, world, local


def build_job(args, rank):
    from spot_amd.actor import TranstSparseActor
    from spot_amd.arena import ParamArena
    from spot_amd.criterion import transt_loss
    from spot_amd.detinit import det_fill_
    from spot_amd.middleware import SparseMiddleware
    from spot_amd.synthetic import Settings, sup_batch, unsup_batch
    from spot_amd.trainer import LTRSparseTrainer
    from spot_amd.transt import transt_resnet50

    # the reference's training configuration: dropout 0.1; in-network augmentation with brightness jitter 0.2,
    # grey-scale 0.05, rotation +-10 degrees, left/right flip 0.35, up/down flip 0.10 (spot_transt_003.py:45,80,85)
    S = Settings(device="cuda", dropout=args.dropout, batch_size_sup=args.sup, batch_size_unsup=args.clips)
    S.aug_settings.update(gray=0.05, jitter=0.2, angle=10, lr_f_prob=0.35, ud_f_prob=0.10)
    S.strong_augmentation_tag = True
    S.attention_dtype = args.attention
    if args.products != "f32":
        from spot_amd import ops as _ops
        _ops.MATMUL_PRECISION = args.products
    ops_seed = 1000 + rank
    from spot_amd import ops as _ops
    _ops.manual_seed(ops_seed)
    torch.manual_seed(ops_seed)
    stu = det_fill_(transt_resnet50(S), 0).cuda()
    tea = det_fill_(transt_resnet50(S, network="teacher"), 0).cuda()
    stu._spot_arena = ParamArena(stu)
    tea._spot_arena = ParamArena(tea)
    actor = TranstSparseActor(S, stu, tea, transt_loss(S))
    lr, wd, br = (S.optimizer_parameters[k] for k in ("learning_rate", "weight_decay", "backbone_ratio"))
    groups = [{"params": [p for n, p in stu.named_parameters() if "backbone" not in n and p.requires_grad]},
              {"params": [p for n, p in stu.named_parameters() if "backbone" in n and p.requires_grad], "lr": br * lr}]
    from spot_amd.optim import FlatAdamW
    opt = FlatAdamW(groups, stu._spot_arena, lr=lr, weight_decay=wd)
    trainer = LTRSparseTrainer(actor, None, opt, S, SparseMiddleware(S))
    data_unsup, mid = unsup_batch(args.clips, 1 + rank, device="cuda")
    data_sup = sup_batch(args.sup, 1 + rank, device="cuda")
    mid = {k: v.cuda() for k, v in mid.items()}
    # Random weights give arbitrary IoU-head confidences; shift the teacher's final IoU bias once so that
    # about 3 of 4 clips pass tau = 0.85 (a trained teacher's regime) instead of none or all.
    with torch.no_grad():
        pseudo = trainer._pseudo_label_generation(data_unsup)
        conf = pseudo["best_conf"].clamp(1e-6, 1 - 1e-6)
        logit = torch.log(conf / (1 - conf))
        target = torch.log(torch.tensor(0.85 / 0.15, device="cuda"))
        shift = target - torch.quantile(logit, 0.25)
        tea.iou_embed.layers[2].bias += shift
        stu.iou_embed.layers[2].bias += shift
    return S, trainer, data_sup, data_unsup, mid


class Cfg3Job:
    """BASELINE.json configs[2]: teacher pseudo-labelling + IoU-confidence filtering over `total` synthetic clips
    (ltr_sparse_trainer.py:288-313 followed by ltr/data/middleware.py:49-152), sharded over the ranks in contiguous
    blocks (spot_amd.dist.shard_range).  The teacher runs in eval() here, so clips do not interact: a rank needs no
    collective to label its shard, and the concatenation of the ranks' outputs equals the single-GPU result bit for
    bit.  One step = every rank labels its whole shard once, `clips_per_pass` clips per network pass."""

    KEYS = ("best_bbox", "best_score", "best_conf", "aug_anno", "aug_success")

    def __init__(self, total, clips_per_pass, rank, world, seed=2):
        from spot_amd.actor import TranstSparseActor
        from spot_amd.detinit import det_fill_
        from spot_amd.dist import shard_range
        from spot_amd.middleware import SparseMiddleware
        from spot_amd.synthetic import Settings, unsup_clips
        from spot_amd.tensordict import TensorDict
        from spot_amd.transt import transt_resnet50
        self.TensorDict = TensorDict
        self.total, self.per_pass = total, clips_per_pass
        self.lo, self.hi = shard_range(total, rank, world)
        S = Settings(device="cuda")
        net = det_fill_(transt_resnet50(S, network="teacher"), 0).cuda().eval()
        self.actor, self.mw = TranstSparseActor(S, net, net, None), SparseMiddleware(S)
        # Random weights give arbitrary IoU-head confidences; the head's last bias is shifted once so that about three
        # of four clips pass tau = 0.85 (a trained teacher's regime).  Every rank derives the shift from the same
        # clips (the job's first 32), so all ranks hold the same teacher.
        with torch.no_grad():
            d0, _ = unsup_clips(0, min(32, total), seed, device="cuda")
            conf = self._label(d0, None)["best_conf"].clamp(1e-6, 1 - 1e-6)
            logit = torch.log(conf / (1 - conf))
            net.iou_embed.layers[2].bias += torch.log(torch.tensor(0.85 / 0.15, device="cuda")) - torch.quantile(logit, 0.25)
        self.data, self.mid = unsup_clips(self.lo, self.hi, seed, device="cuda")

    def _label(self, data, mid):
        d = self.TensorDict({k: v for k, v in data.items() if k.startswith(("key_", "unlabeled_"))})
        pseudo = self.actor(mode="inference", teacher_data=d)
        if mid is None:
            return pseudo
        res = self.mw.record_relay_results(mid, pseudo)
        return {**{k: pseudo[k] for k in ("best_bbox", "best_score", "best_conf")},
                "aug_anno": res["aug_anno"], "aug_success": res["aug_success"]}

    def step(self):
        """Labels of this rank's shard, in clip order."""
        outs = []
        with torch.no_grad():
            for a in range(0, self.hi - self.lo, self.per_pass):
                e = min(a + self.per_pass, self.hi - self.lo)
                outs.append(self._label({k: v[a:e] for k, v in self.data.items()}, {k: v[a:e] for k, v in self.mid.items()}))
        if not outs:
            return None
        return {k: torch.cat([o[k] for o in outs]) for k in self.KEYS}

    def gather(self, mine, collective, on_host):
        """All ranks' labels in clip order on every rank (blocks are contiguous and ordered by rank): 57 bytes a clip."""
        if not collective:
            return mine
        from spot_amd.dist import shard_range
        world = dist.get_world_size()
        sizes = [b - a for a, b in (shard_range(self.total, r, world) for r in range(world))]
        out = {}
        for k in self.KEYS:
            ref = mine[k] if mine is not None else None
            shape = tuple(ref.shape[1:]) if ref is not None else {"best_bbox": (4,), "aug_anno": (4,)}.get(k, ())
            dtype = ref.dtype if ref is not None else (torch.bool if k == "aug_success" else torch.float32)
            dev = "cpu" if on_host else "cuda"
            pad = torch.zeros((max(sizes),) + shape, dtype=torch.uint8 if dtype == torch.bool else dtype, device=dev)
            if ref is not None and ref.shape[0]:
                pad[:ref.shape[0]] = ref.to(pad.dtype).to(dev)
            parts = [torch.empty_like(pad) for _ in range(world)]
            dist.all_gather(parts, pad)
            out[k] = torch.cat([p[:n] for p, n in zip(parts, sizes)]).to(dtype)
        return out


# timer key -> kernel symbol prefix in the profiles (template arguments: A layout, B layout, aligned operands, then
# the workgroup tile and the row-sum flag, which the prefix leaves open)
# (whole tiles of aligned operands run on gemm_ring_kernel<A layout, B layout, tile, row sums>, the rest on gemm_kernel)
KERNEL_SYMBOLS = {"gemm_nt": ("gemm_ring_kernel<1, 1", "gemm_kernel<1, 1, true"),
                  "gemm_nn": ("gemm_ring_kernel<1, 0", "gemm_kernel<1, 0, true"),
                  "gemm_tn": ("gemm_ring_kernel<0, 0", "gemm_kernel<0, 0, true"),
                  "gemm_tt": ("gemm_ring_kernel<0, 1", "gemm_kernel<0, 1, true"),
                  "mha_fwd": ("mha_fwd_pp_kernel", "mha_fwd_kernel<false>"), "mha_bwd": ("mha_bwd_kernel<false",)}


TRAFFIC_FILE = "r3_traffic.json"


def measured_traffic(key):
    """HBM bytes per launch of the kernel class (launch-weighted over its instantiations), from the committed
    counter passes over this same command (profiles/<TRAFFIC_FILE>, written by tools/collect_traffic.sh); None if
    no such measurement exists."""
    symbols = KERNEL_SYMBOLS.get(key.split(" ")[0])
    label = None if symbols is None else " | ".join(x + ", ...>" for x in symbols)
    path = os.path.join(ROOT, "profiles", TRAFFIC_FILE)
    if symbols is None or not os.path.exists(path):
        return None, label
    rows = [v for k, v in json.load(open(path))["kernels"].items() if k.startswith(symbols)]
    n = sum(r["launches"] for r in rows)
    if n == 0:
        return None, label
    return round(sum(r["hbm_bytes_per_launch"] * r["launches"] for r in rows) / n), label


def cpu_baseline(args):
    """The oracle (a port of the reference's PyTorch-CPU path) on a bounded sample of the same workload."""
    import numpy as np
    from oracle import spot_oracle as O
    from spot_amd.detinit import det_state_dict, det_input
    from spot_amd.synthetic import Settings
    from spot_amd.transt import transt_resnet50
    threads = torch.get_num_threads()
    S = Settings(device="cpu")
    net = transt_resnet50(Settings(device="cpu", dropout=0.0))
    names = {k for k, _ in net.named_parameters()}
    sd = {k: (v.float().requires_grad_() if k in names else v) for k, v in det_state_dict(net, 0).items()}
    n_unsup, n_sup = 4, 1
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    key = T(det_input("key_images", (n_unsup, 3, 128, 128), 1, 0, 256, np.uint8))
    unl = T(det_input("unlabeled_images", (n_unsup, 3, 256, 256), 1, 0, 256, np.uint8))
    tmpl = T(det_input("template_images", (n_sup + n_unsup, 3, 128, 128), 1, 0, 256, np.uint8))
    srch = T(det_input("search_images", (n_sup + n_unsup, 3, 256, 256), 1, 0, 256, np.uint8))
    anno = T(np.concatenate([det_input("xy", (n_sup + n_unsup, 2), 1, 40, 110),
                             det_input("wh", (n_sup + n_unsup, 2), 1, 40, 100)], -1).astype(np.float32))
    ext = T(np.concatenate([det_input("exy", (n_unsup, 2), 1, 0, 400), det_input("ewh", (n_unsup, 2), 1, 20, 200)], -1)
            .astype(np.float32))
    rf = T(det_input("rf", (n_unsup,), 1, 0.5, 2.0))
    t0 = time.time()
    reps = 0
    while reps < 1 or time.time() - t0 < 10.0:
        with torch.no_grad():
            out = O.transt_forward(key, unl, sd)
            dec = O.decode_teacher(out["pred_logits"], out["pred_boxes"], out["pred_ious"], 0.40, 32, 256, 256)
            flt = O.middleware_filter(dec["best_bbox"], dec["best_conf"], ext, rf, ext, rf, 256, 0.85)
            O.compact_indices(flt["aug_success"].numpy() | True, n_unsup)
        o = O.transt_forward(tmpl, srch, sd, b_iou=n_sup, training=True)
        loss, _, _ = O.sp_training_loss({"logits": o["pred_logits"][:n_sup], "boxes": o["pred_boxes"][:n_sup],
                                         "ious": o["pred_ious"]},
                                        {"logits": o["pred_logits"][n_sup:], "boxes": o["pred_boxes"][n_sup:]},
                                        anno[:n_sup], anno[n_sup:], S.weight_dict)
        loss.backward()
        for v in sd.values():
            if v.requires_grad:
                v.grad = None
        reps += 1
    el = time.time() - t0
    return {"value": round(reps * n_unsup / el, 4), "unit": "clips/s", "cores": threads, "kind": "port",
            "sample": "%d step(s) of %d clips + %d labelled pair(s): oracle teacher fwd + decode + filter + student "
                      "fwd/bwd (no optimizer), torch-CPU fp32, %d threads" % (reps, n_unsup, n_sup, threads)}


def main_cfg3(args, rank, world, collective):
    from spot_amd import ops
    job = Cfg3Job(args.total_clips, args.clips_per_pass, rank, world)

    def barrier():
        if collective:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        job.step()
    timer = None
    if not args.no_kernel_timing:
        only = None if args.time_kernels == "all" else args.time_kernels.split(",")
        timer = ops.TIMER = ops.KernelTimer(only=only)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        mine = job.step()
    barrier()
    el = time.perf_counter() - t0
    ops.TIMER = None
    per_rank_ms, comm_ranks = [el / args.steps * 1e3], 1
    if collective:
        comm_ranks = dist.get_world_size()
        t = torch.tensor([el], dtype=torch.float64, device="cpu" if args.rehearse_on_one_gpu else "cuda")
        every = [torch.zeros_like(t) for _ in range(comm_ranks)]
        dist.all_gather(every, t)
        per_rank_ms = [float(v.item()) / args.steps * 1e3 for v in every]
        el = max(float(v.item()) for v in every)
    labels = job.gather(mine, collective, args.rehearse_on_one_gpu)        # outside the timed region: 57 bytes a clip
    if rank == 0:
        assert labels["aug_success"].shape[0] == args.total_clips
        roofline = None
        if timer is not None:
            agg = timer.summary()
            dom = max(agg.items(), key=lambda kv: kv[1]["ms"])
            a = dom[1]
            ach = a["work"] / (a["ms"] * 1e-3) / 1e12
            roofline = {"kernel": dom[0], "symbol": measured_traffic(dom[0])[1] or dom[0], "bound": "mfma",
                        "achieved": round(ach, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(ach / PEAK_F32_MFMA_TFLOPS, 4), "traffic": None, "launches": a["launches"],
                        "avg_us": round(a["ms"] * 1e3 / a["launches"], 2), "flops_per_launch": a["work"] / a["launches"]}
        import hashlib
        digest = hashlib.sha256(b"".join(labels[k].cpu().contiguous().numpy().tobytes() for k in Cfg3Job.KEYS)).hexdigest()
        print(json.dumps({
            "metric": "tracked frames/sec (teacher pseudo-labelling + IoU-confidence filter)",
            "value": round(args.total_clips * args.steps / el, 3), "unit": "clips/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(el / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None,
            **({"rehearsal": "all %d ranks shared GPU 0 over gloo: functional check of the multi-rank path, not a "
                             "measurement" % world} if args.rehearse_on_one_gpu else {}),
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "cfg3: teacher pseudo-labelling + decode + IoU-confidence filter + re-projection over "
                                   "%d synthetic clips (key 128^2 / unlabelled 256^2 crops), sharded over the ranks in "
                                   "contiguous blocks, %d clips per network pass, teacher in eval(); no collective on "
                                   "the data path (the ranks' labels are all-gathered after the timed region)"
                                   % (args.total_clips, args.clips_per_pass),
                       "total_clips": args.total_clips, "clips_per_rank": [job.hi - job.lo] if world == 1 else None,
                       "parallelism": "dp%d" % world, "communicator_ranks": comm_ranks,
                       "ms_per_step_per_rank": [round(v, 3) for v in per_rank_ms],
                       "kept": int(labels["aug_success"].sum()), "labels_sha256": digest},
            "roofline": roofline, "cpu_baseline": None}))
    if collective:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", choices=("cfg2", "cfg3"), default="cfg2",
                    help="cfg2 (default, the headline): the whole sparsely-supervised iteration on 32 clips per GPU, weak "
                         "scaling.  cfg3: teacher pseudo-labelling + filter over --total-clips clips sharded over the "
                         "ranks in contiguous blocks, no collective on the data path, strong scaling")
    ap.add_argument("--total-clips", type=int, default=1024, help="cfg3: clips of the whole job")
    ap.add_argument("--clips-per-pass", type=int, default=128, help="cfg3: clips per network pass on a rank")
    ap.add_argument("--clips", type=int, default=32, help="unlabelled clips per GPU per step")
    ap.add_argument("--sup", type=int, default=8, help="labelled pairs per GPU per step")
    ap.add_argument("--dropout", type=float, default=0.1, help="student dropout (reference config: 0.1)")
    ap.add_argument("--attention", choices=("f32", "bf16"), default="f32",
                    help="attention-core matrix products: exact fp32 (default, the headline config) or bf16")
    ap.add_argument("--products", choices=("f32", "bf16x3-linear-fwd", "bf16x3"), default="f32",
                    help="GEMM products: exact fp32 (default, the headline config) or the bf16x3 experiment (operands cut "
                         "in three bf16 pieces, six piece products on the bf16 matrix instruction) for the forward "
                         "nn.Linear products only or for every GEMM launched from Python; a separately labelled line")
    ap.add_argument("--exclusive-steps", type=int, default=2,
                    help="extra untimed-for-throughput steps with the second stream off, for per-kernel durations")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="functional rehearsal of --gpus N on a machine with one GPU: all ranks share GPU 0 and the "
                         "process group runs over gloo (RCCL refuses two ranks on one device); the line it prints says "
                         "so and is not a measurement")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-lookahead", action="store_true",
                    help="do not hand the trainer the next step's unlabelled batch (no teacher pass beside the backward)")
    ap.add_argument("--timer-shapes", action="store_true", help="key the timed GEMM launches by problem shape as well")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--time-kernels", default="gemm_nt",
                    help="kernel classes bracketed with HIP events in the timed steps, comma separated, or 'all' "
                         "(gemm_nt, gemm_nn, gemm_tn, gemm_tt, mha_fwd, mha_bwd).  Default: the class with the "
                         "largest total time in the rocprofv3 summary of this command (the forward-layout GEMM); "
                         "bracketing all six costs about 2 %% of the step")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Started without a launcher: this process becomes the launcher.  It has made no GPU call (importing torch
        # makes none) and makes none: the N ranks are fresh child processes, rank r on GPU r, over RCCL.
        from spot_amd.dist import launch_local
        sys.exit(launch_local(args.gpus, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]))

    from spot_amd import lib, ops
    from spot_amd.tensordict import TensorDict
    lib.load()
    rank, world, local = build_world(args)
    from spot_amd import dist as sdist
    collective = sdist.active()          # more than one rank (or a forced one-rank rehearsal of that path)
    if args.config == "cfg3":
        return main_cfg3(args, rank, world, collective)
    S, trainer, data_sup, data_unsup, mid = build_job(args, rank)

    def barrier():
        if collective:
            dist.barrier()
        torch.cuda.synchronize()

    # every step hands the trainer the next step's unlabelled batch too (here the same resident batch), as a
    # data loader with one batch of look-ahead would
    unsup_view = TensorDict(data_unsup)
    prefetch = not args.no_lookahead

    def step(i, lookahead=True):
        return trainer.train_iteration(i, TensorDict(data_sup), unsup_view, mid,
                                       next_unsup=unsup_view if (prefetch and lookahead) else None)

    for i in range(args.warmup):
        step(i)
    timer = None
    if not args.no_kernel_timing:
        only = None if args.time_kernels == "all" else args.time_kernels.split(",")
        timer = ops.TIMER = ops.KernelTimer(shapes=args.timer_shapes, only=only)
    skipped0 = trainer.skipped
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    barrier()
    el = time.perf_counter() - t0
    ops.TIMER = None
    # The fusion layers and the backbone run the template stream on a second HIP stream and the next step's teacher
    # pass runs beside the backward, so in the timed region a launch's duration includes the kernels that share
    # the chip with it.  A short extra leg with the second
    # stream switched off gives each kernel's duration when it has the chip to itself (reported beside the
    # timed region's figure, never instead of it).
    exclusive = None
    if timer is not None and args.exclusive_steps > 0:
        from spot_amd import fusion
        two = fusion.TWO_STREAMS
        fusion.TWO_STREAMS = False
        step(args.warmup + args.steps, lookahead=False)          # settle into the serial schedule
        exclusive = ops.TIMER = ops.KernelTimer(only=only)
        torch.cuda.synchronize()
        tx = time.perf_counter()
        for i in range(args.exclusive_steps):
            step(args.warmup + args.steps + 1 + i, lookahead=False)
        torch.cuda.synchronize()
        exclusive_ms = (time.perf_counter() - tx) / args.exclusive_steps * 1e3
        ops.TIMER = None
        fusion.TWO_STREAMS = two
    per_rank_ms = [el / args.steps * 1e3]
    comm_ranks = 1
    if collective:
        comm_ranks = dist.get_world_size()
        mine = torch.tensor([el], dtype=torch.float64, device="cpu" if args.rehearse_on_one_gpu else "cuda")
        every = [torch.zeros_like(mine) for _ in range(comm_ranks)]
        dist.all_gather(every, mine)
        per_rank_ms = [float(t.item()) / args.steps * 1e3 for t in every]
        el = max(float(t.item()) for t in every)              # the job is as slow as its slowest rank
    done = args.steps - (trainer.skipped - skipped0)

    if rank == 0:
        roofline = None
        kernels = {}
        if timer is not None:
            agg = timer.summary()
            for k, a in agg.items():
                kernels[k] = {"launches": a["launches"], "ms_total": round(a["ms"], 3),
                              "tflops": round(a["work"] / (a["ms"] * 1e-3) / 1e12, 2) if a["ms"] > 0 else None}
            dom = max(agg.items(), key=lambda kv: kv[1]["ms"])
            a = dom[1]
            ach = a["work"] / (a["ms"] * 1e-3) / 1e12
            traffic, symbol = measured_traffic(dom[0])
            roofline = {"kernel": dom[0], "symbol": symbol, "bound": "mfma", "achieved": round(ach, 2),
                        "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_F32_MFMA_TFLOPS, 4),
                        "traffic": traffic,
                        "traffic_source": (None if traffic is None else
                                           "profiles/%s (committed rocprofv3 counter pass over this same command, "
                                           "launch-weighted over the class's instantiations; not measured in this run)"
                                           % TRAFFIC_FILE),
                        "launches": a["launches"], "avg_us": round(a["ms"] * 1e3 / a["launches"], 2),
                        "flops_per_launch": a["work"] / a["launches"]}
            if exclusive is not None:
                x = exclusive.summary().get(dom[0])
                if x and x["ms"] > 0:
                    xa = x["work"] / (x["ms"] * 1e-3) / 1e12
                    roofline["second_stream_off"] = {
                        "achieved": round(xa, 2), "frac": round(xa / PEAK_F32_MFMA_TFLOPS, 4),
                        "avg_us": round(x["ms"] * 1e3 / x["launches"], 2), "launches": x["launches"],
                        "ms_per_step_serial_schedule": round(exclusive_ms, 3),
                        "kernel_ms_per_step": {k: round(v["ms"] / args.exclusive_steps, 3)
                                               for k, v in exclusive.summary().items()},
                        "note": "same kernels in %d extra steps with the template stream serialised behind the "
                                "search stream and no look-ahead teacher pass: a launch then has the chip to itself"
                                % args.exclusive_steps}
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(args)
        line = {
            "metric": "tracked frames/sec (fwd-bwd pseudo-labelling + train step)",
            "value": round(done * args.clips * world / el, 3), "unit": "clips/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(el / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            **({"rehearsal": "all %d ranks shared GPU 0 over gloo: functional check of the multi-rank path, not a "
                             "measurement" % world} if args.rehearse_on_one_gpu else {}),
            "dtype": ("f32" if args.attention == "f32" else "f32 (bf16 attention products)") +
                     ("" if args.products == "f32" else " via bf16x3 (EXPERIMENT: %s as six bf16 piece products, everything "
                      "else exact fp32; not the headline configuration)"
                      % ("forward nn.Linear products" if args.products == "bf16x3-linear-fwd" else
                         "linear-layer, 1x1- and patch-matrix-convolution GEMMs (forward and both gradients)")),
            "data": "synthetic",
            "config": {"workload": "cfg2: %d clips x 3 frames per GPU per step; teacher pseudo-label + decode + "
                                   "filter + compaction + student fwd/bwd/AdamW on %d labelled + %d pseudo-labelled "
                                   "pairs (TransT ResNet-50, 128^2 template / 256^2 search, fp32%s, student dropout %g, "
                                   "in-network augmentation as spot_transt_003.py: brightness jitter 0.2, grey-scale "
                                   "0.05, rotation 10 deg + flips 0.35/0.10 on the pseudo-labelled pairs; the teacher stays in the module-default train() mode as in the "
                                   "reference, whose actor only toggles the student: batch-statistics BatchNorm and "
                                   "dropout are active in the pseudo-labelling pass too)"
                                   % (args.clips, args.sup, args.clips, "" if args.attention == "f32" else
                                      " with bf16 attention products", args.dropout),
                       "clips_per_gpu": args.clips, "sup_pairs_per_gpu": args.sup, "skipped_steps": args.steps - done,
                       "parallelism": "dp%d" % world, "communicator_ranks": comm_ranks,
                       "ms_per_step_per_rank": [round(v, 3) for v in per_rank_ms]},
            "roofline": roofline, "cpu_baseline": cpu, "kernels": kernels,
        }
        print(json.dumps(line))
    if collective:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
