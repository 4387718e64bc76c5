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
import datetime
import os
import socket
import subprocess
import sys

import torch
import torch.distributed as dist

# how long a collective may stay incomplete before the process group's watchdog (RCCL) or the wait itself (gloo)
# gives up; the default of ten minutes is longer than a benchmark run
COLLECTIVE_TIMEOUT_S = 300


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK/WORLD_SIZE/MASTER_* (set by torchrun or by launch_local).

    With the RCCL backend rank r drives GPU LOCAL_RANK: that device must exist (a clear error names how many GPUs the
    job needs and how many the machine has) and is bound to the communicator at creation (`device_id`), so barriers
    and the first collective never have to guess it."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1 and not _FORCE:
        return 0, 1, 0
    # the host driver supports dmabuf IPC only; read when the first GPU call of the process initialises the runtime
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    rank = int(os.environ["RANK"])
    local = int(os.environ.get("LOCAL_RANK", rank))
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    kwargs = {"timeout": datetime.timedelta(seconds=COLLECTIVE_TIMEOUT_S)}
    if backend == "nccl":
        have = torch.cuda.device_count()
        need = int(os.environ.get("LOCAL_WORLD_SIZE", world))
        if local >= have:
            raise RuntimeError("spot_amd.dist: this job needs %d GPUs on the node (one process per GPU), the machine "
                               "has %d; rank %d has no device %d" % (need, have, rank, local))
        torch.cuda.set_device(local)
        kwargs["device_id"] = torch.device("cuda", local)
    if not dist.is_initialized():
        dist.init_process_group(backend=backend, **kwargs)
    return rank, world, local


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_local(nproc, argv, env=None, timeout=None):
    """Start `nproc` fresh processes of one node running `argv` (a command list), one per GPU, with RANK / LOCAL_RANK /
    WORLD_SIZE / LOCAL_WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, and wait for them: what `torchrun --nnodes=1
    --nproc-per-node N` does, without re-executing anything.  The caller must not have touched the GPU (children are
    new processes, but a parent that holds a device context would share the card with rank 0).  Returns 0 when every
    rank exited 0; otherwise the first failing rank's exit code, after ending the ranks still running (exact PIDs)."""
    base = dict(os.environ if env is None else env)
    base.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), WORLD_SIZE=str(nproc),
                LOCAL_WORLD_SIZE=str(nproc))
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # the host driver supports dmabuf IPC only (RCCL needs it)
    procs = []
    for r in range(nproc):
        e = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen(list(argv), env=e))
    import time
    t0, rc = time.time(), 0
    live = set(range(nproc))
    while live:
        for r in sorted(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0 and rc == 0:
                rc = code
                sys.stderr.write("spot_amd.dist.launch_local: rank %d exited with code %d; ending the other ranks\n"
                                 % (r, code))
        if rc != 0 or (timeout is not None and time.time() - t0 > timeout):
            if rc == 0:
                rc = 124
                sys.stderr.write("spot_amd.dist.launch_local: time limit of %g s reached\n" % timeout)
            for r in sorted(live):
                procs[r].terminate()
            for r in sorted(live):
                try:
                    procs[r].wait(timeout=20)
                except subprocess.TimeoutExpired:
                    procs[r].kill()
                    procs[r].wait()
            live.clear()
        else:
            time.sleep(0.05)
    return rc


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
        self.issued = []           # bucket indices in the order their collectives were issued this step
        if self.active:
            self._check_layout()
            for k, t, o in params:
                t.register_post_accumulate_grad_hook(self._hook)
            arena.add_ready_listener(self._hook)     # gradients written into .grad by the kernels themselves

    def _check_layout(self):
        """Every rank must cut the same buckets (they are a pure function of the parameter layout): a rank with a
        different table would issue collectives of other sizes and the job would hang in the first wait."""
        if not (dist.is_available() and dist.is_initialized()):
            return
        import zlib
        sig = zlib.crc32(repr([(lo, hi, n) for lo, hi, n in self.buckets]).encode()) & 0x7fffffff
        dev = self.arena.grad_flat.device if dist.get_backend() == "nccl" else "cpu"
        t = torch.tensor([sig, -sig], dtype=torch.int64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        if int(t[0]) != sig or int(t[1]) != -sig:
            raise RuntimeError("spot_amd.dist: rank %d cut its gradient buffer into different buckets than its peers "
                               "(%d buckets, signature %d)" % (dist.get_rank(), len(self.buckets), sig))

    def arm(self):
        """Call before backward of each step.  The first armed backward of a process launches nothing early:
        convolution libraries pick their algorithms then, with device-wide synchronisation inside backward, and
        all buckets go out together in finish() instead."""
        self.pending = [b[2] for b in self.buckets]
        self.handles = []
        self.issued = []
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
            # RCCL's collective is a kernel ordered behind that stream.  Any other backend (gloo) moves device tensors
            # through the host from worker threads that synchronise private streams; issued while this process still
            # has compute queued, those waits were observed never to return (two ranks sharing one GPU; the same
            # collectives on an idle device complete), so the device is drained first.  Not the product path.
            if dist.is_initialized() and dist.get_backend() != "nccl":
                torch.cuda.synchronize()
        self.handles.append(dist.all_reduce(self.arena.grad_flat[lo:hi], op=dist.ReduceOp.SUM, async_op=True))
        self.issued.append(b)
        self.pending[b] = -1

    def finish(self):
        """Call after backward: launches buckets whose parameters got no gradient, waits, averages."""
        if not self.active:
            return
        for b, left in enumerate(self.pending):
            if left != -1:
                self._launch(b)
        # handles are waited for in issue order, which is the same on every rank: bucket order when everything goes
        # out here, the order in which backward completed the buckets (a property of the graph, not of timing) with
        # early launches
        for n, (b, h) in enumerate(zip(self.issued, self.handles)):
            try:
                h.wait(datetime.timedelta(seconds=COLLECTIVE_TIMEOUT_S))
            except Exception as e:
                lo, hi, _ = self.buckets[b]
                rank = dist.get_rank() if dist.is_initialized() else 0
                raise RuntimeError("spot_amd.dist: rank %d: all-reduce %d of %d this step (bucket %d, floats %d..%d, "
                                   "issue order %s) did not complete: %s"
                                   % (rank, n, len(self.handles), b, lo, hi, self.issued, e)) from e
        if self.average and self.world > 1:
            self.arena.grad_flat.mul_(1.0 / self.world)
        self._armed = False


def all_ranks_ok(flag_ok, device):
    """True only if every rank reports ok (keeps skip-iteration decisions rank-consistent)."""
    return all_ranks_all([flag_ok], device)[0]


def all_ranks_all(flags, device):
    """Element-wise AND of a few booleans over all ranks, in one small MIN all-reduce."""
    if not active():
        return [bool(f) for f in flags]
    t = torch.tensor([1 if f else 0 for f in flags], dtype=torch.int32, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return [bool(v) for v in t.tolist()]
