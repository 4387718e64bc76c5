"""One process per GPU; gradients all-reduced over RCCL/xGMI in a few large buckets.

Replaces the reference's single-process nn.DataParallel (ltr/admin/multigpu.py:4-15),
which re-broadcasts all 23 M parameters every forward and reduces gradients onto
GPU 0.  Here weights are resident on every rank, pseudo-labelling needs no
collective at all, and the training step has exactly one exchange: a sum of the
flat gradient buffer (arena.py), cut into buckets that are launched as soon as
backward has produced every gradient inside them so the transfers overlap the
rest of backward.  xGMI is point-to-point (7 links per GPU), so buckets are kept
large (default 32 MB) to stay bandwidth- rather than latency-bound.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK/WORLD_SIZE/MASTER_* when launched by torchrun."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1 and not _FORCE:
        return 0, 1, 0
    rank = int(os.environ["RANK"])
    local = int(os.environ.get("LOCAL_RANK", rank))
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(local)
    if not dist.is_initialized():
        dist.init_process_group(backend=backend)
    return rank, world, local


# SPOT_FORCE_DIST=1: run the process group, barriers and bucket collectives even with a single rank (a one-GPU
# rehearsal of the multi-rank code path; every collective is then an identity)
_FORCE = os.environ.get("SPOT_FORCE_DIST", "0") == "1"


def active():
    """True when collectives are to be issued: more than one rank, or a forced single-rank rehearsal."""
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or _FORCE)


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def shard_range(n_items, rank, world):
    """Contiguous block partition of n_items clips over ranks (remainder to the first ranks)."""
    base, rem = divmod(n_items, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


class GradBucketReducer:
    """Bucketed, backward-overlapped gradient all-reduce over a ParamArena's flat grad buffer."""

    def __init__(self, arena, bucket_bytes=32 << 20, average=True, overlap=None):
        self.arena = arena
        self.world = world_size()
        self.average = average
        # overlap=True launches a bucket's collective from inside backward as soon as its gradients are
        # enqueued.  The whole exchange is 93 MB per step (under a millisecond over xGMI against a ~100 ms
        # step), so the default keeps it simple and launches every bucket right after backward, from the
        # caller's thread; SPOT_OVERLAP_ALLREDUCE=1 (or overlap=True) turns the early launches on.
        self.overlap = (os.environ.get("SPOT_OVERLAP_ALLREDUCE", "0") == "1") if overlap is None else bool(overlap)
        if arena.grad_flat is None:
            arena.attach_grads()
        params = [(k, t, o) for (k, t), o in zip(arena.f32, arena.offsets)
                  if isinstance(t, torch.nn.Parameter) and t.requires_grad]
        # backward reaches parameters roughly in reverse registration order
        params.sort(key=lambda e: -e[2])
        self.buckets = []          # [lo, hi, n_params]
        cur_lo = cur_hi = None
        count = 0
        limit = bucket_bytes // 4
        self.param_bucket = {}
        for k, t, o in params:
            lo, hi = o, o + (t.numel() + 63) // 64 * 64
            if cur_lo is None:
                cur_lo, cur_hi, count = lo, hi, 0
            if cur_hi - min(cur_lo, lo) > limit and count > 0:
                self.buckets.append([cur_lo, cur_hi, count])
                cur_lo, cur_hi, count = lo, hi, 0
            cur_lo, cur_hi = min(cur_lo, lo), max(cur_hi, hi)
            self.param_bucket[id(t)] = len(self.buckets)
            count += 1
        if count:
            self.buckets.append([cur_lo, cur_hi, count])
        self.pending = [0] * len(self.buckets)
        self.handles = []
        self._seen = set()
        self._armed = False
        self._armed_before = False
        self._overlap = True
        self.active = self.world > 1 or active()
        if self.active:
            for k, t, o in params:
                t.register_post_accumulate_grad_hook(self._hook)
            arena.add_ready_listener(self._hook)     # gradients written into .grad by the kernels themselves

    def arm(self):
        """Call before backward of each step.  The first armed backward of a process launches nothing early:
        convolution libraries pick their algorithms then, with device-wide synchronisation inside backward, and
        all buckets go out together in finish() instead."""
        self.pending = [b[2] for b in self.buckets]
        self.handles = []
        self._seen = set()
        self._armed = True
        self._overlap = self.overlap and self._armed_before
        self._armed_before = True

    def _hook(self, p):
        # A parameter can be announced twice (by the kernel that wrote its gradient into the buffer and by
        # autograd's own accumulation node); it counts once.
        if not self._armed or id(p) in self._seen:
            return
        self._seen.add(id(p))
        b = self.param_bucket[id(p)]
        self.pending[b] -= 1
        if self.pending[b] == 0 and self._overlap:
            self._launch(b)

    def _launch(self, b):
        lo, hi, _ = self.buckets[b]
        if self.arena.grad_flat.is_cuda:
            # The bucket's gradients were written by kernels on the main stream AND on the second stream the
            # fusion layers / backbone use for the template branch; the collective only orders itself behind
            # the stream it is issued from, so that stream first waits for every stream that produced them.
            from . import ops
            cur = torch.cuda.current_stream()
            for side, main in ops.compute_streams():
                if side.cuda_stream != cur.cuda_stream:
                    cur.wait_stream(side)
                if main.cuda_stream != cur.cuda_stream:
                    cur.wait_stream(main)
        self.handles.append(dist.all_reduce(self.arena.grad_flat[lo:hi], op=dist.ReduceOp.SUM, async_op=True))
        self.pending[b] = -1

    def finish(self):
        """Call after backward: launches buckets whose parameters got no gradient, waits, averages."""
        if not self.active:
            return
        for b, left in enumerate(self.pending):
            if left != -1:
                self._launch(b)
        for h in self.handles:
            h.wait()
        if self.average and self.world > 1:
            self.arena.grad_flat.mul_(1.0 / self.world)
        self._armed = False


def all_ranks_ok(flag_ok, device):
    """True only if every rank reports ok (keeps skip-iteration decisions rank-consistent)."""
    if not active():
        return bool(flag_ok)
    t = torch.tensor([1 if flag_ok else 0], dtype=torch.int32, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(t.item())
