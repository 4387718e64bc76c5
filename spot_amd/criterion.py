"""SP-TransT matcher + loss without host round-trips.

Mirror of ltr/models/loss/transt_matcher.py:10-113 and
ltr/models/tracking/transt_criterion.py:8-181.  The reference extracts the box
corners with four .item() calls per sample and builds index lists in numpy;
here the positive / negative token sets are dense [B, L] masks written by one
kernel (spot_match_masks) and every loss term is a masked reduction over all
tokens, so nothing synchronises with the host.  The paired GIoU/IoU (forward and
backward) is the box kernel.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import lib, ops


class TrackingMatcher(nn.Module):
    """Returns dense maps instead of index lists: (pos [B,L] bool, neg [B,L] uint8 multiplicity or None); a negative
    token listed k times by the reference (possible only for boxes off the map) has neg = k."""

    @torch.no_grad()
    def forward(self, outputs, targets, tag="sup"):
        logits = outputs["logits_" + tag]
        bs, num_queries = logits.shape[:2]
        n = int(math.isqrt(num_queries))
        gt = targets if torch.is_tensor(targets) else torch.cat([t["boxes"] for t in targets], dim=0)
        gt = gt.reshape(bs, 4).contiguous().float()
        pos = torch.empty((bs, num_queries), dtype=torch.uint8, device=logits.device)
        neg = torch.empty_like(pos) if tag == "sup" else None
        lib.call("spot_match_masks", ops._p(gt), bs, n, 3, ops._p(pos), ops._p(neg), ops._st())
        return pos.bool(), neg


def build_matcher():
    return TrackingMatcher()


class SetCriterionTransT(nn.Module):
    def __init__(self, settings, matcher):
        super().__init__()
        self.settings = settings
        self.matcher = matcher
        self.losses = ["labels", "boxes"]
        self.eos_coef = settings.eos_coef
        self.num_classes = settings.num_classes
        self.weight_dict = settings.weight_dict
        empty_weight = torch.ones(self.num_classes + 1)
        empty_weight[-1] = self.eos_coef
        self.register_buffer("empty_weight", empty_weight)

    def forward(self, outputs, targets, tag="sup"):
        """targets: [B,4] normalised cxcywh tensor (or the reference's list of dicts)."""
        logits = outputs["logits_" + tag]
        boxes = outputs["boxes_" + tag]
        B, L, _ = logits.shape
        gt = targets if torch.is_tensor(targets) else torch.cat([t["boxes"] for t in targets], dim=0)
        gt = gt.reshape(B, 4).float()
        pos, neg = self.matcher(outputs, gt, tag=tag)
        num_pos = pos.sum().clamp(min=1).float()

        # classification: positives are class 0, everything else the no-object class
        target_classes = (~pos).long()
        losses = {"loss_ce": F.cross_entropy(logits.transpose(1, 2), target_classes, self.empty_weight)}

        # box regression on the positive tokens, as masked dense sums
        tgt = gt[:, None, :].expand(B, L, 4)
        posf = pos.float()
        losses["loss_bbox"] = ((boxes - tgt).abs().sum(-1) * posf).sum() / num_pos
        giou, iou = ops.GIoUPairFn.apply(boxes.reshape(-1, 4), tgt.reshape(-1, 4))
        giou, iou = giou.view(B, L), iou.view(B, L)
        zero = torch.zeros((), device=logits.device)
        losses["loss_giou"] = torch.where(pos, 1 - giou, zero).sum() / num_pos
        losses["iou"] = torch.where(pos, iou, zero).sum().detach() / num_pos

        if neg is not None and ("ious_" + tag) in outputs:
            pred = outputs["ious_" + tag].squeeze(2)
            num_neg = neg.sum().clamp(min=1).float()
            err = (pred - iou.detach()).abs()
            losses["loss_iou_pred"] = torch.where(pos, err, zero).sum() / num_pos + \
                torch.where(neg > 0, err * neg, zero).sum() / num_neg
        return losses


def transt_loss(settings):
    matcher = build_matcher()
    criterion = SetCriterionTransT(settings=settings, matcher=matcher)
    return criterion.to(torch.device(settings.device))
