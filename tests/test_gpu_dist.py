"""Stream ordering of the bucketed gradient all-reduce on the GPU.

The training step's gradients are written by kernels on two compute streams (the template branch runs on a
second one) and partly straight into the flat gradient buffer by the kernels themselves.  A bucket's collective
must be ordered behind ALL of them.  Here the collective is replaced by a stand-in that snapshots the bucket on
the stream the collective would run on; every snapshot must equal the bucket's final contents."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


class _Done:
    def wait(self, timeout=None):
        return True


def test_bucket_collectives_see_complete_gradients(monkeypatch):
    import torch.distributed as dist
    from spot_amd import dist as sdist
    from spot_amd.actor import TranstSparseActor
    from spot_amd.arena import ParamArena
    from spot_amd.criterion import transt_loss
    from spot_amd.detinit import det_fill_
    from spot_amd.synthetic import Settings
    from spot_amd.tensordict import TensorDict
    from spot_amd.transt import transt_resnet50
    from tests.test_oracle_model import student_batches

    snaps = []

    def fake_all_reduce(t, op=None, async_op=False):
        snaps.append((t.data_ptr(), t.numel(), t.clone()))      # stream-ordered like the real collective's read
        return _Done()

    monkeypatch.setattr(sdist, "world_size", lambda: 2)
    monkeypatch.setattr(dist, "all_reduce", fake_all_reduce)

    S = Settings(dropout=0.0)
    net = det_fill_(transt_resnet50(Settings(device="cpu", dropout=0.0)), 17).cuda().train()
    arena = ParamArena(net, with_grads=True)
    reducer = sdist.GradBucketReducer(arena, bucket_bytes=4 << 20, average=False, overlap=True)
    assert len(reducer.buckets) > 5
    actor = TranstSparseActor(S, net, net, transt_loss(S))
    sup, sp = student_batches(17, 2, 6)
    sup = TensorDict({k: v.cuda() for k, v in sup.items()})
    sp = TensorDict({k: v.cuda() for k, v in sp.items()})
    for _ in range(2):                                       # second pass: warm allocator and kernels, tighter timing
        snaps.clear()
        arena.zero_grad()
        loss, _ = actor(mode="spsup", burnin_data=TensorDict(sup), sp_data=TensorDict(sp))
        reducer.arm()
        loss.backward()
        reducer.finish()
        torch.cuda.synchronize()
        assert len(snaps) == len(reducer.buckets)
        base = arena.grad_flat.data_ptr()
        covered = 0
        for ptr, n, snap in snaps:
            lo = (ptr - base) // 4
            final = arena.grad_flat[lo:lo + n]
            assert torch.equal(snap, final), "bucket at %d was reduced before its gradients were complete" % lo
            covered += n
        assert covered >= sum(p.numel() for p in net.parameters() if p.requires_grad)
        assert float(arena.grad_flat.abs().sum()) > 0


def test_rccl_single_rank_bucket_allreduce_is_identity(monkeypatch):
    """The real RCCL path on one rank: communicator set-up, asynchronous all-reduce of slices of the flat gradient
    buffer issued behind both compute streams, wait.  With one rank every collective is an identity, so the
    gradients must equal those of a step taken without the reducer."""
    import socket
    import torch.distributed as dist
    from spot_amd import dist as sdist
    from spot_amd.actor import TranstSparseActor
    from spot_amd.arena import ParamArena
    from spot_amd.criterion import transt_loss
    from spot_amd.detinit import det_fill_
    from spot_amd.synthetic import Settings
    from spot_amd.tensordict import TensorDict
    from spot_amd.transt import transt_resnet50
    from tests.test_oracle_model import student_batches

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1)
    try:
        monkeypatch.setattr(sdist, "_FORCE", True)
        assert sdist.active()
        S = Settings(dropout=0.0)
        net = det_fill_(transt_resnet50(Settings(device="cpu", dropout=0.0)), 19).cuda().train()
        arena = ParamArena(net, with_grads=True)
        actor = TranstSparseActor(S, net, net, transt_loss(S))
        sup, sp = student_batches(19, 2, 4)
        sup = TensorDict({k: v.cuda() for k, v in sup.items()})
        sp = TensorDict({k: v.cuda() for k, v in sp.items()})

        def grads(reducer):
            arena.zero_grad()
            loss, _ = actor(mode="spsup", burnin_data=TensorDict(sup), sp_data=TensorDict(sp))
            if reducer is not None:
                reducer.arm()
            loss.backward()
            if reducer is not None:
                reducer.finish()
            torch.cuda.synchronize()
            return arena.grad_flat.clone()

        base = grads(None)
        for overlap in (False, True):
            reducer = sdist.GradBucketReducer(arena, bucket_bytes=4 << 20, overlap=overlap)
            assert reducer.active and len(reducer.buckets) > 5
            for _ in range(2):                                  # early launches start with the second armed backward
                got = grads(reducer)
            nb = [k for k, _ in arena.f32 if "backbone" not in k]
            # same tolerance as the run-to-run comparison of two plain steps (atomic summation order differs)
            cos = torch.nn.functional.cosine_similarity(got, base, dim=0)
            assert float(cos) > 0.9999, float(cos)
            assert sdist.all_ranks_ok(True, "cuda") and not sdist.all_ranks_ok(False, "cuda")
            assert len(nb) > 0
    finally:
        dist.destroy_process_group()


def test_bench_starts_its_own_ranks_two_rank_rehearsal(tmp_path):
    """`python bench.py --gpus 2` without a launcher: the parent starts two fresh rank processes (spot_amd.dist.
    launch_local), they form a group, run the step with the gradient buckets all-reduced between them, agree on the
    slowest rank's time and rank 0 prints one line.  On a one-GPU box both ranks share GPU 0 over gloo."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--rehearse-on-one-gpu", "--steps", "2", "--warmup",
           "1", "--no-cpu-baseline", "--exclusive-steps", "0"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=110)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["communicator_ranks"] == 2
    assert len(line["config"]["ms_per_step_per_rank"]) == 2 and line["value"] > 0
    assert line["ms_per_step"] >= max(line["config"]["ms_per_step_per_rank"]) - 1e-6


def test_cfg3_sharded_pseudo_labelling_equals_single_batch():
    """bench.py's cfg3 job (teacher pseudo-labelling + filter over a clip set sharded in contiguous rank blocks): one
    rank labelling the whole set in passes of 32, three ranks labelling their blocks, and one batch holding every clip
    all give the same labels bit for bit (the teacher is in eval(): clips do not interact, and no kernel on the path
    chooses its arithmetic by batch size)."""
    import bench
    total = 96
    whole = bench.Cfg3Job(total, total, 0, 1).step()                        # one network pass over all clips
    passes = bench.Cfg3Job(total, 32, 0, 1).step()                          # N = 1 as bench.py runs it
    assert 0 < int(whole["aug_success"].sum()) < total
    for k in bench.Cfg3Job.KEYS:
        assert torch.equal(passes[k], whole[k]), k
    shards = [bench.Cfg3Job(total, 32, r, 3) for r in range(3)]
    assert [(j.lo, j.hi) for j in shards] == [(0, 32), (32, 64), (64, 96)]
    parts = [j.step() for j in shards]
    for k in bench.Cfg3Job.KEYS:
        assert torch.equal(torch.cat([p[k] for p in parts]), whole[k]), k


def test_bench_cfg3_two_rank_rehearsal():
    """`python bench.py --config cfg3 --gpus 2`: two self-launched ranks label their halves of the clip set, agree on
    the slowest rank's time, all-gather the labels; rank 0's line names cfg3 and carries the same label checksum as a
    one-rank run of the same job."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    lines = []
    for extra in (["--gpus", "2", "--rehearse-on-one-gpu"], ["--gpus", "1"]):
        cmd = [sys.executable, os.path.join(root, "bench.py"), "--config", "cfg3", "--total-clips", "64", "--clips-per-pass",
               "32", "--steps", "1", "--warmup", "1", "--no-kernel-timing"] + extra
        res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=110)
        assert res.returncode == 0, res.stderr[-2000:]
        out = [l for l in res.stdout.splitlines() if l.startswith("{")]
        assert len(out) == 1
        lines.append(json.loads(out[0]))
    two, one = lines
    assert two["n_gpus"] == 2 and two["config"]["communicator_ranks"] == 2 and two["scaling"] == "strong"
    assert two["config"]["workload"].startswith("cfg3") and len(two["config"]["ms_per_step_per_rank"]) == 2
    assert two["config"]["labels_sha256"] == one["config"]["labels_sha256"] and two["config"]["kept"] == one["config"]["kept"]
    assert 0 < one["config"]["kept"] < 64
