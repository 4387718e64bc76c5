"""Sparsely-supervised training loop (hot loop only) on the HIP kernels.

Host mirror of ltr/trainers/ltr_sparse_trainer.py:49-181,242-377: the burn-in stage (student alone on labelled pairs)
and, per iteration of the sparsely-supervised stage,
  teacher EMA every `teacher_update_frequency` iterations -> teacher pseudo-labels
  -> middleware filter -> keep/refill compaction -> student step.
Checkpointing, tensorboard and the DataLoader assembly of the reference are out
of scope; `loader_sup` / `loader_unsup` are any iterables yielding the same
TensorDicts (see synthetic.py).

What changed for the hardware:
  * everything between the two network passes stays on the GPU: decode, filter
    and compaction are kernels, the refill is a row gather by index
    (ltr_sparse_trainer.py:334-357 builds Python lists of per-sample tensors);
  * one host read per iteration decides "skip" (no valid pseudo label, or a
    non-finite loss), and the decision is made rank-consistent with a MIN
    all-reduce so data-parallel ranks never diverge.  The read happens after the
    student forward has been enqueued (the host never waits for the teacher); the
    reference returns before that forward when nothing was kept (:328-332), so an
    iteration skipped for that reason puts the student's BatchNorm running
    statistics back to what they were before it (the forward then ran on all-zero
    stand-in rows and its result is discarded);
  * the Siamese variant's gradient clipping (:126-131) is one norm + scale pass over
    the flat gradient buffer, and a non-finite norm skips the step on every rank;
  * gradients live in one flat buffer and are all-reduced in buckets that overlap
    backward (dist.py); the optimizer is created over the same parameters.
"""
import math
import time

import torch

from . import ops
from .arena import ParamArena
from .dist import GradBucketReducer, all_ranks_all, all_ranks_ok
from .tensordict import TensorDict


class LTRSparseTrainer:
    def __init__(self, actor, loader_assemble, optimizer, settings, middleware, lr_scheduler=None):
        self.actor = actor
        self.loader_assemble = loader_assemble
        self.optimizer = optimizer
        self.settings = settings
        self.middleware = middleware
        self.lr_scheduler = lr_scheduler
        self.device = torch.device(getattr(settings, "device", "cuda"))
        self.move_data_to_gpu = getattr(settings, "move_data_to_gpu", True)
        self.epoch = 0
        self.stats = {}
        stu = actor.student_net.module if hasattr(actor.student_net, "module") else actor.student_net
        self.arena = getattr(stu, "_spot_arena", None) or ParamArena(stu)
        stu._spot_arena = self.arena
        self.reducer = GradBucketReducer(self.arena)
        # torch.nn.utils.clip_grad_norm_(student.parameters(), 25) for the Siamese variant only, as the reference keys it
        # on the student's class name (base_sparse_trainer.py:32, ltr_sparse_trainer.py:126-131)
        self.arch = stu.__class__.__name__
        self.clip_grad_norm = getattr(settings, "clip_grad_norm", 25.0 if self.arch == "Ocean" else None)
        # the student's buffers (BatchNorm running statistics and counters), with a place to keep their values of
        # before the forward pass
        self._buffers = [t for _, t in self.arena.f32 if not isinstance(t, torch.nn.Parameter)]
        self._buffers += [t for _, t in self.arena.i64]
        # Snapshot = the two flat allocations whole (one copy each instead of one per buffer: ~130 launches less on the
        # critical stream).  Parameters do not change between the snapshot and the skip decision, so putting the whole
        # allocation back restores exactly the buffers.
        self._flat_before = torch.empty_like(self.arena.flat) if self._buffers else None
        self._i64_before = torch.empty_like(self.arena.flat_i64) if self._buffers else None
        self.num_frames = 0
        self.skipped = 0
        self._prefetched = None
        self._teacher_stream = None
        self._masks = {}

    # -- the two stages of an epoch ----------------------------------------------
    def train_epoch(self):
        """One epoch (ltr_sparse_trainer.py:142-181): burn-in on labelled pairs only up to `burn_in_epochs`; the first
        epoch after it copies the student into the teacher (keep rate 0); from then on the sparsely-supervised loop.
        `loader_assemble` provides the loaders as the reference's does (init_burnin_loader / loader_burnin,
        init_sparse_supervised_loader / loader_sup / loader_unsup)."""
        self.epoch += 1
        interval = getattr(self.settings, "train_epoch_interval", 1)
        if self.epoch <= self.settings.burn_in_epochs:
            if self.epoch % interval == 0:
                self.loader_assemble.init_burnin_loader(self.epoch)
                return self.cycle_dataset_burnin(self.loader_assemble.loader_burnin)
            return None
        if self.epoch == self.settings.burn_in_epochs + 1:
            self._teacher_model_update(keep_rate=0)
        if self.epoch % interval == 0:
            self.loader_assemble.init_sparse_supervised_loader(self.epoch)
            return self.cycle_dataset_sparse_supervised(self.loader_assemble.loader_sup, self.loader_assemble.loader_unsup)
        return None

    def cycle_dataset_burnin(self, loader, training=True):
        """Burn-in stage (ltr_sparse_trainer.py:49-88): the student alone on labelled pairs."""
        self.num_frames, self.start_time = 0, time.time()
        last = None
        for data in loader:
            stats = self.burnin_iteration(data, training)
            last = stats if stats is not None else last
        return last

    def burnin_iteration(self, data, training=True):
        """One burn-in step: forward with no strong augmentation (masks all False, :264-269), skip on a non-finite
        loss (:270-273, made rank-consistent), backward, gradient exchange, clipping for the Siamese variant, step."""
        if self.move_data_to_gpu:
            data = data.to(self.device, non_blocking=True)
        if training:
            self.arena.zero_grad()
        loss, stats = self._burnin_training(data, training)
        finite = bool(torch.isfinite(loss.detach()).item())
        if not all_ranks_ok(finite, self.device):
            self.skipped += 1
            return None
        if training:
            self.reducer.arm()
            loss.backward()
            self.reducer.finish()
            if self.clip_grad_norm:
                norm = ops.clip_grad_norm_(self.arena.grad_flat, float(self.clip_grad_norm))
                if not all_ranks_ok(math.isfinite(norm.item()), self.device):
                    self.skipped += 1
                    return None
            self.optimizer.step()
        self.num_frames += data["search_anno"].shape[0]
        return stats

    def _burnin_training(self, data, training=True):
        self.actor.train(training)
        with torch.set_grad_enabled(training):
            data["masks"] = self._mask(data["search_anno"].shape[0], False)
            return self.actor(mode="burnin", burnin_data=data)

    # -- one epoch over the two loaders ----------------------------------------
    def cycle_dataset_sparse_supervised(self, loader_sup, loader_unsup, training=True):
        self.num_frames, self.start_time = 0, time.time()
        last = None
        for i, (data_sup, (data_unsup, middle_info)) in enumerate(zip(loader_sup, loader_unsup)):
            last = self.train_iteration(i, data_sup, data_unsup, middle_info, training)
        return last

    def train_iteration(self, i, data_sup, data_unsup, middle_info, training=True, next_unsup=None):
        """One iteration of the hot loop.  ``next_unsup`` (optional): the unlabelled batch of iteration i+1.  When
        no teacher update falls between the two iterations, its pseudo labels are computed on a separate stream
        while this iteration's student backward runs: the teacher's weights are the same either way, so the
        labels are the ones iteration i+1 would compute itself."""
        if (i + 1) % self.settings.teacher_update_frequency == 0:
            self._teacher_model_update(keep_rate=self.settings.keep_rate)
        if self.move_data_to_gpu:
            data_unsup = data_unsup.to(self.device, non_blocking=True)
            data_sup = data_sup.to(self.device, non_blocking=True)
        pseudo_label = self._take_prefetched(i, data_unsup)
        if pseudo_label is None:
            pseudo_label = self._pseudo_label_generation(data_unsup)
        aug_label = self._middleware_ops(middle_info, pseudo_label)
        if training:
            self.arena.zero_grad()      # before the forward pass: also resets the kernels' gradient bookkeeping
            if self._buffers:
                self._flat_before.copy_(self.arena.flat)
                self._i64_before.copy_(self.arena.flat_i64)
        loss, stats, n_keep = self._sparse_training(data_sup, data_unsup, aug_label, training)
        if training:
            # The next iteration's teacher pass goes out before the host waits for this iteration's flags: the device
            # then has work queued across the read, and the pass runs beside the tail of the forward and the backward.
            # A skipped iteration leaves the teacher as it was, so the labels stay valid either way.
            self._prefetch_pseudo_labels(i + 1, next_unsup)
        # single host read: kept count and loss finiteness
        flags = torch.stack([n_keep.float().reshape(()), torch.isfinite(loss.detach()).float().reshape(())]).tolist()
        kept, finite = all_ranks_all([flags[0] > 0, flags[1] > 0], self.device)
        if not kept:
            # no valid pseudo label on some rank: the reference skips before the student forward, so nothing of
            # that forward may persist (a non-finite loss, below, is noticed after the forward there too)
            if training and self._buffers:
                self.arena.flat.copy_(self._flat_before)
                self.arena.flat_i64.copy_(self._i64_before)
            self.skipped += 1
            return None
        if not finite:
            self.skipped += 1
            return None
        if training:
            self.reducer.arm()
            loss.backward()
            self.reducer.finish()
            if self.clip_grad_norm:
                norm = ops.clip_grad_norm_(self.arena.grad_flat, float(self.clip_grad_norm))
                if not all_ranks_ok(math.isfinite(norm.item()), self.device):
                    self.skipped += 1
                    return None
            self.optimizer.step()
        self.num_frames += self.settings.batch_size
        return stats

    # -- teacher pass of the next iteration beside this iteration's backward ----------
    def _prefetch_pseudo_labels(self, i_next, next_unsup):
        self._prefetched = None
        if next_unsup is None or self.device.type != "cuda":
            return
        if (i_next + 1) % self.settings.teacher_update_frequency == 0:
            return                              # the teacher changes before iteration i_next: nothing to reuse
        main = torch.cuda.current_stream()
        if self._teacher_stream is None:
            self._teacher_stream = torch.cuda.Stream(device=self.device)
        ts = self._teacher_stream
        ts.wait_stream(main)
        with torch.cuda.stream(ts):
            if self.move_data_to_gpu:
                next_unsup = next_unsup.to(self.device, non_blocking=True)
            pseudo = self._pseudo_label_generation(next_unsup)
        self._prefetched = (i_next, [id(v) for v in next_unsup.values()], pseudo, ts)

    def _take_prefetched(self, i, data_unsup):
        pf, self._prefetched = self._prefetched, None
        if pf is None or pf[0] != i or pf[1] != [id(v) for v in data_unsup.values()]:
            return None
        _, _, pseudo, ts = pf
        main = torch.cuda.current_stream()
        main.wait_stream(ts)
        for v in pseudo.values():
            if torch.is_tensor(v):
                v.record_stream(main)
        return pseudo

    # -- pieces (names follow the reference) --------------------------------------
    def _teacher_model_update(self, keep_rate=0.998):
        self.actor.train(False)
        with torch.no_grad():
            self.actor(mode="update", keep_rate=keep_rate)

    def _pseudo_label_generation(self, data):
        self.actor.train(False)
        with torch.no_grad():
            data_load = TensorDict()
            for prefix in ("key", "unlabeled"):
                for suffix in ("images", "anno", "avg"):
                    data_load["{}_{}".format(prefix, suffix)] = data["{}_{}".format(prefix, suffix)]
            return self.actor(mode="inference", teacher_data=data_load)

    def _middleware_ops(self, middle_info, pseudo_label):
        return self.middleware.record_relay_results(middle_info, pseudo_label)

    def compact(self, data_unsup, aug_label):
        """Keep successful clips in order and refill cyclically to batch_size_unsup (:334-357)."""
        n_out = self.settings.batch_size_unsup
        keep = aug_label.get("aug_success_u8")
        if keep is None:
            keep = aug_label["aug_success"].to(torch.uint8)
        idx, n_keep = ops.compact_indices(keep.contiguous(), n_out)
        src = dict(data_unsup)
        src["aug_anno"] = aug_label["aug_anno"]
        out = TensorDict()
        for prefix in ("relay", "aug"):
            for suffix in ("images", "anno", "avg"):
                key = "{}_{}".format(prefix, suffix)
                out[key] = ops.gather_rows(src[key].contiguous(), idx)
        return out, n_keep, idx

    def _mask(self, n, value):
        key = (n, value)
        if key not in self._masks:
            self._masks[key] = torch.full((n,), value, dtype=torch.bool, device=self.device)
        return self._masks[key]

    def _sparse_training(self, data_sup, data_unsup, aug_label, training=True):
        self.actor.train(training)
        with torch.set_grad_enabled(training):
            data_load, n_keep, _ = self.compact(data_unsup, aug_label)
            # which instances are unsupervised pairs, i.e. take the strong augmentation (:366-368)
            data_sup["masks"] = self._mask(data_sup["search_anno"].shape[0], False)
            data_load["masks"] = self._mask(data_load["aug_anno"].shape[0], True)
            loss, stats = self.actor(mode="spsup", burnin_data=data_sup, sp_data=data_load)
        return loss, stats, n_keep
