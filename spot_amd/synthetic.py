"""Synthetic clip batches with the field names and shapes of the reference's loaders
(ltr/data/processing.py:105-219, ltr/data/loader.py:200-323): there are no datasets offline."""
import numpy as np
import torch

from .detinit import det_input
from .tensordict import TensorDict


def unsup_batch(B, seed, temp_sz=128, search_sz=256, device="cpu"):
    """(data_unsup, middle_info): key/unlabeled/relay/aug crops of B clips x 3 frames."""
    d = TensorDict()
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    for name, sz in (("key", temp_sz), ("unlabeled", search_sz), ("relay", temp_sz), ("aug", search_sz)):
        d[name + "_images"] = t(det_input(name + "_images", (B, 3, sz, sz), seed, 0, 256, np.uint8))
        d[name + "_avg"] = t(det_input(name + "_avg", (B, 3), seed, 0, 255))
        xy = det_input(name + "_xy", (B, 2), seed, sz * 0.15, sz * 0.45)
        wh = det_input(name + "_wh", (B, 2), seed, sz * 0.15, sz * 0.4)
        d[name + "_anno"] = t(np.concatenate([xy, wh], -1).astype(np.float32))
    m = {}
    for name in ("unlabeled", "aug"):
        xy = det_input(name + "_ext_xy", (B, 1, 2), seed, 0, 400)
        wh = det_input(name + "_ext_wh", (B, 1, 2), seed, 20, 200)
        m[name + "_box_extract"] = t(np.concatenate([xy, wh], -1).astype(np.float32))
        m[name + "_resize_factors"] = t(det_input(name + "_rf", (B,), seed, 0.5, 2.0))
    return d, m


def unsup_clips(lo, hi, seed, block=32, device="cpu", **kw):
    """Clips [lo, hi) of an unbounded synthetic clip set: the set is the concatenation of unsup_batch blocks of `block`
    clips (block b seeded with (seed, b)), so any contiguous range can be produced without its predecessors and every
    partition of a job over ranks sees the same clips."""
    parts_d, parts_m = [], []
    for b in range(lo // block, (hi + block - 1) // block if hi > lo else lo // block):
        d, m = unsup_batch(block, seed * 100003 + b, device="cpu", **kw)
        a, e = max(lo, b * block) - b * block, min(hi, (b + 1) * block) - b * block
        parts_d.append({k: v[a:e] for k, v in d.items()})
        parts_m.append({k: v[a:e] for k, v in m.items()})
    if not parts_d:
        d, m = unsup_batch(1, seed * 100003, device="cpu", **kw)
        parts_d, parts_m = [{k: v[:0] for k, v in d.items()}], [{k: v[:0] for k, v in m.items()}]
    data = TensorDict({k: torch.cat([p[k] for p in parts_d]).to(device) for k in parts_d[0]})
    mid = {k: torch.cat([p[k] for p in parts_m]).to(device) for k in parts_m[0]}
    return data, mid


def sup_batch(B, seed, temp_sz=128, search_sz=256, device="cpu"):
    d = TensorDict()
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    for name, sz in (("template", temp_sz), ("search", search_sz)):
        d[name + "_images"] = t(det_input(name + "_images", (B, 3, sz, sz), seed, 0, 256, np.uint8))
        d[name + "_avg"] = t(det_input(name + "_avg", (B, 3), seed, 0, 255))
        xy = det_input(name + "_xy", (B, 2), seed, sz * 0.15, sz * 0.45)
        wh = det_input(name + "_wh", (B, 2), seed, sz * 0.15, sz * 0.4)
        d[name + "_anno"] = t(np.concatenate([xy, wh], -1).astype(np.float32))
    return d


class Settings:
    """Plain namespace with the attribute names of ltr/train_settings/spot/spot_transt_003.py:21-111."""

    def __init__(self, **kw):
        self.device = "cuda"
        self.batch_size_sup, self.batch_size_unsup = 8, 24
        self.teacher_update_frequency, self.keep_rate = 5, 0.998
        self.burn_in_epochs, self.train_epoch_interval = 50, 1
        self.normalize_mean = [0.485, 0.456, 0.406]
        self.normalize_std = [0.229, 0.224, 0.225]
        self.search_feature_sz, self.template_feature_sz = 32, 16
        self.search_sz, self.temp_sz = 256, 128
        self.aug_settings = {"gray": 0.0, "jitter": 0.0, "angle": 10, "lr_f_prob": 0.35, "ud_f_prob": 0.10}
        self.position_embedding = "sine"
        self.hidden_dim, self.dropout, self.nheads = 256, 0.1, 8
        self.dim_feedforward, self.featurefusion_layers = 2048, 4
        self.num_classes, self.eos_coef = 1, 0.0625
        self.weight_dict = {"loss_ce_sup": 8.334, "loss_bbox_sup": 5, "loss_giou_sup": 2, "loss_iou_pred_sup": 1.0,
                            "loss_ce_unsup": 8.334, "loss_bbox_unsup": 5, "loss_giou_unsup": 2}
        self.curriculum_settings = {"confidence_threshold": 0.85, "max_gap": None, "propagating_step": 1,
                                    "window_penalty": 0.40}
        self.iou_prediction_tag = True
        self.strong_augmentation_tag = False
        self.optimizer_parameters = {"learning_rate": 1e-4, "weight_decay": 1e-4, "backbone_ratio": 0.1}
        for k, v in kw.items():
            setattr(self, k, v)
        self.batch_size = self.batch_size_sup + self.batch_size_unsup
