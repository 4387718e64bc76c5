"""Middleware between teacher inference and student training.

Host mirror of ltr/data/middleware.py:7-152 (record_relay_results): crop ->
image recovery of the pseudo boxes, the IoU-confidence filter
``conf >= confidence_threshold`` and the re-projection into the augmented crop.
The reference walks the batch in a Python list comprehension with .tolist() /
float() device->host syncs per clip; here it is one kernel launch for the batch.

The per-epoch JSON sampler-state bookkeeping (update_sampler_state, :154-223) is
host-side curriculum state outside this package's scope; ``epoch_state_info`` is
still filled (from one batched device->host copy) when ``record_state`` is set so
a caller that owns that bookkeeping can consume it.
"""
import torch

from . import ops


class SparseMiddleware:
    def __init__(self, settings, record_state=False):
        self.settings = settings
        self.search_sz = settings.search_sz
        self.confidence_threshold = settings.curriculum_settings["confidence_threshold"]
        self.device = settings.device
        self.record_state = record_state
        self.epoch_state_info = dict()

    def record_relay_results(self, middle_info, pseudo_label):
        dev = pseudo_label["best_bbox"].device
        f = lambda k, shape: middle_info[k].to(dev, non_blocking=True).float().reshape(shape).contiguous()
        B = pseudo_label["best_bbox"].shape[0]
        rec, aug, keep = ops.middleware_filter(
            pseudo_label["best_bbox"].contiguous(), pseudo_label["best_conf"].contiguous(),
            f("unlabeled_box_extract", (B, 4)), f("unlabeled_resize_factors", (B,)),
            f("aug_box_extract", (B, 4)), f("aug_resize_factors", (B,)),
            self.search_sz, self.confidence_threshold)
        out = {"aug_anno": aug, "aug_success": keep.bool(), "bbox_recovered": rec, "aug_success_u8": keep}
        if self.record_state and "unlabeled_record_keys" in middle_info:
            self._record(middle_info, rec, pseudo_label["best_conf"], keep)
        return out

    def _record(self, middle_info, rec, conf, keep):
        packed = torch.cat([rec, conf[:, None], keep[:, None].float()], dim=1).cpu().tolist()
        for i, row in enumerate(packed):
            if row[5] < 0.5:
                continue
            seq_key, path_id, frame_id = middle_info["unlabeled_record_keys"][i][0]
            e = self.epoch_state_info.setdefault(
                seq_key, {"path_ids": [], "relay_frame_ids": [], "bboxes": [], "confidence": []})
            e["path_ids"].append(path_id)
            e["relay_frame_ids"].append(frame_id)
            e["bboxes"].append(row[:4])
            e["confidence"].append(round(row[4], 4))
