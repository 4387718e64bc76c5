"""World-size-2 checks of the data-parallel plumbing on CPU (gloo): clip sharding, bucketed gradient
all-reduce over the flat arena overlapped with backward, and the rank-consistent skip decision."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _model():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Linear(16, 64), torch.nn.ReLU(), torch.nn.BatchNorm1d(64),
                               torch.nn.Linear(64, 64), torch.nn.ReLU(), torch.nn.Linear(64, 4))


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from spot_amd.arena import ParamArena
    from spot_amd.dist import GradBucketReducer, all_ranks_ok, init_from_env, shard_range
    r, w, _ = init_from_env("gloo")
    assert (r, w) == (rank, world)
    net = _model()
    arena = ParamArena(net)
    red = GradBucketReducer(arena, bucket_bytes=4096)        # several buckets
    assert len(red.buckets) >= 2
    x = torch.from_numpy(np.random.default_rng(10 + rank).standard_normal((8, 16)).astype(np.float32))
    for _ in range(2):                                       # second pass checks re-arming and zeroing
        arena.zero_grad()
        red.arm()
        net(x).square().sum().backward()
        red.finish()
    grads = {k: p.grad.clone() for k, p in net.named_parameters()}
    # every .grad must still be a view into the flat buffer
    for (k, t), o in zip(arena.f32, arena.offsets):
        if isinstance(t, torch.nn.Parameter):
            assert t.grad.data_ptr() == arena.grad_flat[o:].data_ptr(), k
    lo, hi = shard_range(1024, rank, world)
    ok_all = all_ranks_ok(True, "cpu")
    ok_mixed = all_ranks_ok(rank == 0, "cpu")
    torch.save({"grads": grads, "range": (lo, hi), "ok_all": ok_all, "ok_mixed": ok_mixed}, out % rank)
    dist.destroy_process_group()


def test_bucketed_grad_allreduce_world2(tmp_path):
    world, port = 2, _free_port()
    out = str(tmp_path / "r%d.pt")
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    res = [torch.load(out % r, weights_only=False) for r in range(world)]
    # reference: average of the two ranks' local gradients
    ref = None
    for rank in range(world):
        net = _model()
        x = torch.from_numpy(np.random.default_rng(10 + rank).standard_normal((8, 16)).astype(np.float32))
        net(x).square().sum().backward()
        g = {k: p.grad.clone() for k, p in net.named_parameters()}
        ref = g if ref is None else {k: ref[k] + g[k] for k in g}
    ref = {k: v / world for k, v in ref.items()}
    for r in range(world):
        for k, v in ref.items():
            torch.testing.assert_close(res[r]["grads"][k], v, rtol=1e-5, atol=1e-6)
        assert res[r]["ok_all"] is True and res[r]["ok_mixed"] is False
    assert res[0]["range"] == (0, 512) and res[1]["range"] == (512, 1024)


def test_shard_range_covers_everything():
    from spot_amd.dist import shard_range
    for n, w in ((1024, 8), (10, 3), (5, 8), (0, 2)):
        spans = [shard_range(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1
