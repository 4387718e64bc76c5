"""Box utilities (mirror of ltr/utils/box_ops.py:9-32,105-156) on the HIP kernels."""
import torch

from . import ops


def box_cxcywh_to_xyxy(x):
    x_c, y_c, w, h = x.unbind(-1)
    return torch.stack([(x_c - 0.5 * w), (y_c - 0.5 * h), (x_c + 0.5 * w), (y_c + 0.5 * h)], dim=-1)


def box_xyxy_to_cxcywh(x):
    x0, y0, x1, y1 = x.unbind(-1)
    return torch.stack([(x0 + x1) / 2, (y0 + y1) / 2, (x1 - x0), (y1 - y0)], dim=-1)


def box_xywh_to_xyxy(x):
    x0, y0, w, h = x.unbind(-1)
    return torch.stack([x0, y0, x0 + w, y0 + h], dim=-1)


def box_xyxy_to_xywh(x):
    x0, y0, x1, y1 = x.unbind(-1)
    return torch.stack([x0, y0, x1 - x0, y1 - y0], dim=-1)


def box_iou_pair_xyxy(boxes1, boxes2):
    """Paired IoU and union of [P,4] xyxy boxes (one kernel launch)."""
    _, iou, union = ops.box_iou_pair(boxes1.contiguous(), boxes2.contiguous(), cxcywh=False, want_giou=False)
    return iou, union


def generalized_box_iou_pair_xyxy(boxes1, boxes2):
    """Paired (GIoU, IoU) of [P,4] xyxy boxes."""
    giou, iou, _ = ops.box_iou_pair(boxes1.contiguous(), boxes2.contiguous(), cxcywh=False)
    return giou, iou


def generalized_box_iou_pair_cxcywh(src, tgt):
    """Differentiable (GIoU, IoU) of paired cxcywh boxes; gradient flows to ``src``."""
    return ops.GIoUPairFn.apply(src, tgt)
