"""One rank of the world-size-2 CPU (gloo) checks; started by spot_amd.dist.launch_local from tests/test_dist_gloo.py.

Part 1: bucketed gradient all-reduce over the flat arena, sharding, the rank-consistent flag.
Part 2: LTRSparseTrainer.train_iteration itself with the device-bound pieces (teacher pass, filter kernel, compaction
kernel, student network) replaced by small CPU stand-ins, so that the iteration's control flow runs with two ranks: the
skip decision when ONE rank has no valid pseudo label, re-arming, gradient averaging, the optimizer step.
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def model():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Linear(16, 64), torch.nn.ReLU(), torch.nn.BatchNorm1d(64),
                               torch.nn.Linear(64, 64), torch.nn.ReLU(), torch.nn.Linear(64, 4))


def rank_input(rank):
    return torch.from_numpy(np.random.default_rng(10 + rank).standard_normal((8, 16)).astype(np.float32))


def part1(rank, world):
    from spot_amd.arena import ParamArena
    from spot_amd.dist import GradBucketReducer, all_ranks_ok, shard_range
    net = model()
    arena = ParamArena(net)
    red = GradBucketReducer(arena, bucket_bytes=4096)        # several buckets
    assert len(red.buckets) >= 2
    x = rank_input(rank)
    for _ in range(2):                                       # second pass checks re-arming and zeroing
        arena.zero_grad()
        red.arm()
        net(x).square().sum().backward()
        red.finish()
    assert red.issued == list(range(len(red.buckets)))
    grads = {k: p.grad.clone() for k, p in net.named_parameters()}
    for (k, t), o in zip(arena.f32, arena.offsets):          # every .grad must still be a view into the flat buffer
        if isinstance(t, torch.nn.Parameter):
            assert t.grad.data_ptr() == arena.grad_flat[o:].data_ptr(), k
    # the same exchange with buckets launched from inside backward as soon as they are complete (overlap=True): the
    # first armed backward of a process launches nothing early, the later ones do
    net2 = model()
    arena2 = ParamArena(net2)
    red2 = GradBucketReducer(arena2, bucket_bytes=4096, overlap=True)
    early = []
    for _ in range(3):
        arena2.zero_grad()
        red2.arm()
        net2(x).square().sum().backward()
        early.append(list(red2.issued))                      # what went out before finish()
        red2.finish()
    grads_overlap = {k: p.grad.clone() for k, p in net2.named_parameters()}
    return {"grads": grads, "range": shard_range(1024, rank, world), "ok_all": all_ranks_ok(True, "cpu"),
            "ok_mixed": all_ranks_ok(rank == 0, "cpu"), "grads_overlap": grads_overlap, "early": early,
            "issued_overlap": list(red2.issued), "n_buckets": len(red2.buckets)}


def part2(rank, world, overlap=False):
    from spot_amd.tensordict import TensorDict
    from spot_amd.trainer import LTRSparseTrainer
    from oracle import spot_oracle as O

    class Settings:
        device = "cpu"
        move_data_to_gpu = False
        teacher_update_frequency = 1000
        keep_rate = 0.998
        batch_size_sup, batch_size_unsup, batch_size = 2, 4, 6

    class Actor:                                  # the interface LTRSparseTrainer uses
        def __init__(self):
            self.student_net = model()

        def train(self, mode=True):
            self.student_net.train(mode)

        def __call__(self, mode, teacher_data=None, burnin_data=None, sp_data=None, keep_rate=None):
            if mode == "inference":
                return {"conf": teacher_data["unlabeled_anno"][:, 0]}
            assert mode == "spsup"
            x = torch.cat([burnin_data["search_anno"], sp_data["aug_anno"]])
            loss = self.student_net(x).square().mean()
            return loss, {"Loss/total": loss.detach()}

    class Trainer(LTRSparseTrainer):
        def _middleware_ops(self, middle_info, pseudo):
            return {"aug_success": pseudo["conf"] >= 0.5, "aug_anno": middle_info["anno"]}

        def compact(self, data_unsup, aug):                      # the keep / refill rule, from the CPU restatement
            keep = aug["aug_success"].numpy()
            idx = O.compact_indices(keep, self.settings.batch_size_unsup)
            n_keep = torch.tensor(int(keep.sum()))
            if idx is None:
                idx = np.zeros(self.settings.batch_size_unsup, np.int64)
            out = TensorDict({"aug_anno": aug["aug_anno"][torch.from_numpy(np.asarray(idx, np.int64))]})
            return out, n_keep, idx

    actor = Actor()
    opt = torch.optim.SGD(actor.student_net.parameters(), lr=0.1)
    trainer = Trainer(actor, None, opt, Settings(), None)
    if overlap:
        from spot_amd.dist import GradBucketReducer
        trainer.reducer = GradBucketReducer(trainer.arena, bucket_bytes=4096, overlap=True)
    issued = []
    rng = np.random.default_rng(100 + rank)
    log = []
    for i in range(4):
        sup = TensorDict({"search_anno": torch.from_numpy(rng.standard_normal((2, 16)).astype(np.float32))})
        conf = torch.full((4, 16), 0.9)
        if i == 1 and rank == 1:
            conf[:] = 0.1                                        # this rank's teacher produced nothing usable
        if i == 2:
            conf[1:, 0] = 0.1                                    # one clip kept on both ranks: refilled to four
        unsup = TensorDict({"key_images": conf, "key_anno": conf, "key_avg": conf, "unlabeled_images": conf,
                            "unlabeled_anno": conf, "unlabeled_avg": conf})
        mid = {"anno": torch.from_numpy(rng.standard_normal((4, 16)).astype(np.float32))}
        before = torch.cat([p.detach().flatten().clone() for p in actor.student_net.parameters()])
        stats = trainer.train_iteration(i, sup, unsup, mid)
        after = torch.cat([p.detach().flatten() for p in actor.student_net.parameters()])
        log.append({"stepped": stats is not None, "changed": not torch.equal(before, after)})
        issued.append(list(trainer.reducer.issued))
    weights = torch.cat([p.detach().flatten() for p in actor.student_net.parameters()])
    running = actor.student_net[2].running_mean.clone()
    return {"log": log, "weights": weights, "skipped": trainer.skipped, "running_mean": running, "issued": issued}


def main():
    out = sys.argv[1]
    from spot_amd.dist import init_from_env
    rank, world, _ = init_from_env("gloo")
    res = {"rank": rank, "world": world, "part1": part1(rank, world), "part2": part2(rank, world),
           "part2_overlap": part2(rank, world, overlap=True)}
    torch.save(res, out % rank)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
