"""Sparsely-supervised actor for SP-TransT on the HIP kernels.

Host mirror of ltr/actor/base_sparse_actor.py:4-53 and
ltr/actor/transt_sparse_actor.py:8-251: same constructor, same
``actor(mode, teacher_data=, burnin_data=, sp_data=, keep_rate=)`` dispatch with
modes update / inference / burnin / spsup, same returned keys.

Differences that matter for speed, not for results:
  * update: one EMA kernel over flat parameter storage (arena.py);
  * inference: the decode is one kernel (softmax, Hanning blend, arg-max, gather,
    box conversion, clamps) — no intermediate tensors, no host sync;
  * spsup/burnin: targets are a [B,4] tensor built with tensor ops instead of a
    per-sample Python loop; stats are returned as device tensors and only turned
    into floats when the caller asks (``stats_to_float``).
"""
import torch

from . import ops
from .arena import ParamArena
from .tensordict import TensorDict


class BaseSparseActor:
    def __init__(self, settings, student_net, teacher_net, objective):
        self.student_net = student_net
        self.teacher_net = teacher_net
        self.objective = objective
        self.settings = settings

    def __call__(self, data):
        raise NotImplementedError

    def to(self, device):
        self.student_net.to(device)
        self.teacher_net.to(device)

    def train(self, mode=True):
        # The trainer flips the mode around every network pass; walking the module tree costs half a millisecond of
        # host time each, at points where the device waits for the next launch.  The walk is skipped when the root
        # already has the mode (nothing in this package sets the mode of a sub-module alone).
        mode = bool(mode)
        if self.student_net.training != mode:
            self.student_net.train(mode)
        if self.objective is not None and self.objective.training != mode:
            self.objective.train(mode)

    def eval(self):
        self.train(False)


def _unwrap(net):
    return net.module if hasattr(net, "module") else net


class TranstSparseActor(BaseSparseActor):
    def __init__(self, settings, student_net, teacher_net, objective):
        super().__init__(settings, student_net, teacher_net, objective)
        self._arenas = None

    # -- update -------------------------------------------------------------
    def _get_arenas(self):
        if self._arenas is None:
            stu, tea = _unwrap(self.student_net), _unwrap(self.teacher_net)
            a_s = getattr(stu, "_spot_arena", None) or ParamArena(stu)
            a_t = getattr(tea, "_spot_arena", None) or ParamArena(tea)
            stu._spot_arena, tea._spot_arena = a_s, a_t
            if a_s.signature() != a_t.signature():
                missing = [k for k, _ in a_t.f32 if k not in dict(a_s.f32)]
                raise Exception("{} is not found in student model".format(missing[:1] or "entry"))
            self._arenas = (a_s, a_t)
        return self._arenas

    @torch.no_grad()
    def _update_teacher_model(self, keep_rate=0.998):
        a_s, a_t = self._get_arenas()
        ops.ema_lerp_(a_t.flat, a_s.flat, keep_rate)
        if a_t.i64:
            ops.ema_lerp_i64_(a_t.flat_i64, a_s.flat_i64, keep_rate)

    # -- inference ------------------------------------------------------------
    @torch.no_grad()
    def _inference_with_teacher(self, teacher_data):
        hw = teacher_data["unlabeled_images"].shape[-2:]
        outputs = self.teacher_net(teacher_data=teacher_data)
        return self._decode_teacher_result(outputs, hw)

    def _decode_teacher_result(self, outputs, search_hw):
        h, w = search_hw
        ious = outputs["pred_ious"]
        bbox, score, conf, idx = ops.decode_teacher(
            outputs["pred_logits"].contiguous(), outputs["pred_boxes"].contiguous(),
            ious.reshape(ious.shape[0], -1).contiguous(),
            self.settings.curriculum_settings["window_penalty"], self.settings.search_feature_sz, w, h)
        return TensorDict({"best_bbox": bbox, "best_score": score, "best_conf": conf, "best_idx": idx})

    # -- training ---------------------------------------------------------------
    @staticmethod
    def _build_targets(anno_xywh, hw):
        """xywh px -> normalised cxcywh (transt_sparse_actor.py:203-224), batched."""
        h, w = hw
        a = anno_xywh.float()
        return torch.stack([(a[:, 0] + a[:, 2] / 2) / w, (a[:, 1] + a[:, 3] / 2) / h, a[:, 2] / w, a[:, 3] / h], dim=1)

    def _burnin_training(self, burnin_data):
        hw = burnin_data["search_images"].shape[-2:]
        outputs = self.student_net(burnin_data=burnin_data)
        targets = self._build_targets(outputs["search_anno"], hw)
        loss_dict = self.objective(outputs, targets, tag="sup")
        wd = self.objective.weight_dict
        losses = sum(loss_dict[k] * wd[k + "_sup"] for k in loss_dict.keys() if k + "_sup" in wd)
        stats = {"Loss/total": losses.detach(), "Loss/ce": loss_dict["loss_ce"].detach(),
                 "Loss/bbox": loss_dict["loss_bbox"].detach(), "Loss/giou": loss_dict["loss_giou"].detach(),
                 "iou": loss_dict["iou"].detach()}
        if "loss_iou_pred" in loss_dict:
            stats["Loss/iou_pred"] = loss_dict["loss_iou_pred"].detach()
        return losses, stats

    def _sp_training(self, burnin_data, sp_data):
        hw = burnin_data["search_images"].shape[-2:]
        outputs = self.student_net(burnin_data=burnin_data, sp_data=sp_data)
        bs, bu = burnin_data["search_anno"].shape[0], sp_data["aug_anno"].shape[0]
        batch_size = bs + bu
        targets_sup = self._build_targets(outputs["search_anno"], hw)
        targets_unsup = self._build_targets(outputs["aug_anno"], hw)
        wd = self.objective.weight_dict
        ld_sup = self.objective({k: outputs[k] for k in outputs if k.endswith("_sup")}, targets_sup, tag="sup")
        loss_sup = sum(ld_sup[k] * wd[k + "_sup"] for k in ld_sup.keys() if k + "_sup" in wd)
        ld_unsup = self.objective({k: outputs[k] for k in outputs if k.endswith("_unsup")}, targets_unsup, tag="unsup")
        loss_unsup = sum(ld_unsup[k] * wd[k + "_unsup"] for k in ld_unsup.keys() if k + "_unsup" in wd)
        losses = (float(bs) * loss_sup + float(bu) * loss_unsup) / batch_size
        stats = {"Loss/total": losses.detach()}
        names = {"loss_ce": "Loss/ce", "loss_bbox": "Loss/bbox", "loss_giou": "Loss/giou",
                 "loss_iou_pred": "Loss/iou_pred", "iou": "iou"}
        for key, name in names.items():
            if key not in ld_sup:
                continue
            if key not in ld_unsup:
                stats[name] = ld_sup[key].detach()
            else:
                stats[name] = (float(bs) * ld_sup[key].detach() + float(bu) * ld_unsup[key].detach()) / batch_size
        stats["iou_sup"] = ld_sup["iou"].detach()
        stats["iou_unsup"] = ld_unsup["iou"].detach()
        return losses, stats

    @staticmethod
    def stats_to_float(stats):
        """One device->host transfer for all statistics."""
        keys = list(stats.keys())
        vals = torch.stack([stats[k].float().reshape(()) for k in keys]).tolist()
        return dict(zip(keys, vals))

    def __call__(self, mode, teacher_data=None, burnin_data=None, sp_data=None, keep_rate=0.998):
        if mode == "update":
            self._update_teacher_model(keep_rate=keep_rate)
        elif mode == "inference":
            return self._inference_with_teacher(teacher_data=teacher_data)
        elif mode == "burnin":
            return self._burnin_training(burnin_data=burnin_data)
        elif mode == "spsup":
            return self._sp_training(burnin_data=burnin_data, sp_data=sp_data)
        else:
            assert False, "No operation {} for TranstSparseActor".format(mode)
