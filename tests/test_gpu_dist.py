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
    def wait(self):
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


