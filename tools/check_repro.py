#!/usr/bin/env python3
"""Run-to-run spread of the student step's gradients (atomic accumulation order is the only intended source)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from spot_amd.actor import TranstSparseActor  # noqa: E402
from spot_amd.criterion import transt_loss  # noqa: E402
from spot_amd.detinit import det_fill_  # noqa: E402
from spot_amd.synthetic import Settings, sup_batch, unsup_batch  # noqa: E402
from spot_amd.tensordict import TensorDict  # noqa: E402
from spot_amd.transt import transt_resnet50  # noqa: E402


def main():
    S = Settings(dropout=0.0)
    bs, bu = int(sys.argv[1]) if len(sys.argv) > 1 else 1, int(sys.argv[2]) if len(sys.argv) > 2 else 2
    runs = []
    for rep in range(3):
        net = det_fill_(transt_resnet50(Settings(device="cpu", dropout=0.0)), 11).cuda().train()
        actor = TranstSparseActor(S, net, net, transt_loss(S))
        sup = sup_batch(bs, 5, device="cuda")
        un, _ = unsup_batch(bu, 5, device="cuda")
        sp = TensorDict({k: v for k, v in un.items() if k.startswith(("relay_", "aug_"))})
        loss, _ = actor(mode="spsup", burnin_data=TensorDict(sup), sp_data=sp)
        loss.backward()
        torch.cuda.synchronize()
        runs.append((loss.item(), {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None}))
    for a, b in ((0, 1), (0, 2)):
        worst = {"backbone": (0, ""), "other": (0, "")}
        for k, g in runs[a][1].items():
            d = float((g - runs[b][1][k]).abs().max()) / (float(g.abs().max()) + 1e-20)
            grp = "backbone" if "backbone" in k else "other"
            if d > worst[grp][0]:
                worst[grp] = (d, k)
        print("runs %d vs %d: loss %.7f / %.7f  worst rel diff backbone %.2e (%s)  other %.2e (%s)"
              % (a, b, runs[a][0], runs[b][0], worst["backbone"][0], worst["backbone"][1], worst["other"][0],
                 worst["other"][1]))


if __name__ == "__main__":
    main()
