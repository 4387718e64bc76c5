"""SP-TransT matcher + loss without host round-trips.

Mirror of ltr/models/loss/transt_matcher.py:10-113 and
ltr/models/tracking/transt_criterion.py:8-181.  The reference extracts the box
corners with four .item() calls per sample and builds index lists in numpy;
here the positive / negative token sets are dense [B, L] masks written by one
kernel (spot_match_masks) and the loss terms are masked reductions over all
tokens inside one kernel pair (spot_transt_loss_f32 / _bwd_f32), so nothing
synchronises with the host.
"""
import math

import torch
import torch.nn as nn

from . import lib, ops


class TrackingMatcher(nn.Module):
    """Returns dense maps instead of index lists: (pos [B,L] uint8 0/1, neg [B,L] uint8 multiplicity or None); a negative
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
        return pos, neg


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
        pred = outputs.get("ious_" + tag) if neg is not None else None
        # class 0 (object) for the positive tokens, the no-object class for the rest
        ce, bbox, giou, iou, iou_pred = ops.TranstLossFn.apply(
            logits, boxes, pred.squeeze(2) if pred is not None else None, gt, pos, neg if pred is not None else None,
            1.0, float(self.eos_coef))
        losses = {"loss_ce": ce, "loss_bbox": bbox, "loss_giou": giou, "iou": iou}
        if pred is not None:
            losses["loss_iou_pred"] = iou_pred
        return losses


def transt_loss(settings):
    matcher = build_matcher()
    criterion = SetCriterionTransT(settings=settings, matcher=matcher)
    return criterion.to(torch.device(settings.device))
