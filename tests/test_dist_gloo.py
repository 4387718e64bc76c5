"""World-size-2 checks of the data-parallel plumbing on CPU (gloo): clip sharding, bucketed gradient
all-reduce over the flat arena overlapped with backward, and the rank-consistent skip decision."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_two_ranks(tmp_path):
    """Two ranks through the launcher bench.py uses when it is started without torchrun (spot_amd.dist.launch_local)."""
    from spot_amd.dist import launch_local
    out = str(tmp_path / "r%d.pt")
    env = dict(os.environ, OMP_NUM_THREADS="2")
    rc = launch_local(2, [sys.executable, os.path.join(ROOT, "tests", "dist_gloo_worker.py"), out], env=env, timeout=300)
    assert rc == 0
    return [torch.load(out % r, weights_only=False) for r in range(2)]


def test_world2_bucketed_allreduce_and_trainer_iteration(tmp_path):
    from tests.dist_gloo_worker import model, rank_input
    world = 2
    res = _run_two_ranks(tmp_path)
    assert [r["rank"] for r in res] == [0, 1] and all(r["world"] == 2 for r in res)
    # ---- bucketed all-reduce: the average of the two ranks' local gradients
    ref = None
    for rank in range(world):
        net = model()
        net(rank_input(rank)).square().sum().backward()
        g = {k: p.grad.clone() for k, p in net.named_parameters()}
        ref = g if ref is None else {k: ref[k] + g[k] for k in g}
    ref = {k: v / world for k, v in ref.items()}
    for r in range(world):
        p1 = res[r]["part1"]
        for k, v in ref.items():
            torch.testing.assert_close(p1["grads"][k], v, rtol=1e-5, atol=1e-6)
        assert p1["ok_all"] is True and p1["ok_mixed"] is False
    assert res[0]["part1"]["range"] == (0, 512) and res[1]["part1"]["range"] == (512, 1024)
    # ---- the same exchange with early launches (overlap=True): same averaged gradients as the plain exchange, the
    # buckets go out from inside backward from the second armed backward on, in the same order on both ranks
    for r in range(world):
        p1 = res[r]["part1"]
        for k, v in p1["grads"].items():
            assert torch.equal(p1["grads_overlap"][k], v), k
        assert p1["early"][0] == [] and len(p1["early"][1]) == p1["n_buckets"] and p1["early"][2] == p1["early"][1]
        assert sorted(p1["issued_overlap"]) == list(range(p1["n_buckets"]))
    assert res[0]["part1"]["issued_overlap"] == res[1]["part1"]["issued_overlap"]
    # ---- train_iteration: iteration 1 is skipped on BOTH ranks although only rank 1 had nothing to train on; the
    # other iterations step, and the students stay bit-identical because they only ever see averaged gradients
    for r in range(world):
        p2 = res[r]["part2"]
        assert [e["stepped"] for e in p2["log"]] == [True, False, True, True]
        assert [e["changed"] for e in p2["log"]] == [True, False, True, True]
        assert p2["skipped"] == 1
    assert torch.equal(res[0]["part2"]["weights"], res[1]["part2"]["weights"])
    assert not torch.equal(res[0]["part2"]["running_mean"], res[1]["part2"]["running_mean"])   # BatchNorm stays per rank
    # ---- the same four iterations with the gradient buckets launched from inside backward: same skip decisions, the
    # same issue order on both ranks in every iteration, and weights equal to the plain exchange's bit for bit
    for r in range(world):
        po = res[r]["part2_overlap"]
        assert [e["stepped"] for e in po["log"]] == [True, False, True, True] and po["skipped"] == 1
        assert torch.equal(po["weights"], res[r]["part2"]["weights"])
    assert res[0]["part2_overlap"]["issued"] == res[1]["part2_overlap"]["issued"]
    assert all(len(i) > 1 for n, i in enumerate(res[0]["part2_overlap"]["issued"]) if n != 1)


def test_launcher_reports_a_failing_rank():
    from spot_amd.dist import launch_local
    code = "import os, sys, time\nif os.environ['RANK'] == '1':\n    sys.exit(7)\ntime.sleep(60)"
    assert launch_local(2, [sys.executable, "-c", code], timeout=120) == 7


def test_shard_range_covers_everything():
    from spot_amd.dist import shard_range
    for n, w in ((1024, 8), (10, 3), (5, 8), (0, 2)):
        spans = [shard_range(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1
