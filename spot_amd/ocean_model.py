"""SP-Ocean: backbone, model wrapper and sparse actor of the Siamese variant on the HIP kernels.

Mirrors ltr/models/backbone/ocean_backbone.py (ResNet50_Ocean: ResNet-50 to layer3 with an unpadded 7x7 stem, a
3x3 / stride 2 shortcut in layer2 and a dilated layer3), ltr/models/tracking/ocean_model.py (class Ocean,
ocean_resnet50) and ltr/actor/ocean_sparse_actor.py (OceanSparseActor) with the same constructor arguments, forward
modes, online API and state-dict keys.  Every map of this network is odd-sized (125, 63, 31, 15 on the 255 / 127
crops), so all its convolutions run as patch matrix + GEMM at a padded row pitch and its BatchNorm layers take the
element-wise walk (spot_amd/ops.py:conv_general, batch_norm_act); the head is spot_amd/ocean.py."""
import numpy as np
import torch
import torch.nn as nn

from . import ocean, ops
from .actor import TranstSparseActor
from .box_ops import box_xyxy_to_xywh
from .tensordict import TensorDict
from .transforms_in_net import EllipseRotationAffine, GrayScale, IndependentTransform, JointTransform
from .transt import NecessaryPreprocess, TransT


class Bottleneck(nn.Module):
    """ocean_backbone.py:32-80: 1x1 -> 3x3 (stride or dilation) -> 1x1 with a residual; parameter holder, the
    arithmetic runs in the library's kernels."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, dilation=1):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        padding = 2 - stride
        if downsample is not None and dilation > 1:
            dilation = dilation // 2
            padding = dilation
        assert stride == 1 or dilation == 1, "stride and dilation must have one equals to zero at least"
        if dilation > 1:
            padding = dilation
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=stride, padding=padding, bias=False, dilation=dilation)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, kernel_size=1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        out = ops.batch_norm_act(ops.conv_general(x, self.conv1), self.bn1, relu=True)
        out = ops.conv_bn_act(out, self.conv2, self.bn2, relu=True)
        residual = x
        if self.downsample is not None:
            residual = ops.batch_norm_act(ops.conv_general(x, self.downsample[0]), self.downsample[1])
        return ops.batch_norm_act(ops.conv_general(out, self.conv3), self.bn3, relu=True, residual=residual)


class ResNet_plus2(nn.Module):
    """ocean_backbone.py:85-160."""

    def __init__(self, block, layers, used_layers):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=0, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.feature_size = 128 * block.expansion
        self.used_layers = used_layers
        self.layer3_use = 3 in used_layers
        if self.layer3_use:
            self.layer3 = self._make_layer(block, 256, layers[2], stride=1, dilation=2)
            self.feature_size = (256 + 128) * block.expansion
        else:
            self.layer3 = lambda x: x
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, float(np.sqrt(2.0 / n)))
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()

    def _make_layer(self, block, planes, blocks, stride=1, dilation=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            if stride == 1 and dilation == 1:
                conv = nn.Conv2d(self.inplanes, planes * block.expansion, kernel_size=1, stride=stride, bias=False)
            else:
                dd, padding = (dilation // 2, dilation // 2) if dilation > 1 else (1, 0)
                conv = nn.Conv2d(self.inplanes, planes * block.expansion, kernel_size=3, stride=stride, bias=False,
                                 padding=padding, dilation=dd)
            downsample = nn.Sequential(conv, nn.BatchNorm2d(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride=stride, downsample=downsample, dilation=dilation)]
        self.inplanes = planes * block.expansion
        layers += [block(self.inplanes, planes, dilation=dilation) for _ in range(1, blocks)]
        return nn.Sequential(*layers)

    def forward(self, x):
        x_ = ops.batch_norm_act(ops.conv_general(x, self.conv1), self.bn1, relu=True)
        x = ops.max_pool_3x3_s2(x_, self.maxpool)
        p1 = self.layer1(x)
        p2 = self.layer2(p1)
        p3 = self.layer3(p2)
        return [x_, p1, p2], p3


class ResNet50_Ocean(nn.Module):
    """ocean_backbone.py:15-26."""

    def __init__(self, used_layers=None):
        super().__init__()
        self.features = ResNet_plus2(Bottleneck, [3, 4, 6, 3], used_layers=used_layers if used_layers is not None else [2, 3, 4])

    def forward(self, x):
        return self.features(x)


def build_backbone_ocean(settings):
    return ResNet50_Ocean(used_layers=settings.used_layers)


class Ocean(nn.Module):
    """ocean_model.py:20-436: sparsely-supervised Ocean.  forward(burnin_data / teacher_data / sp_data) returns the
    reference's TensorDict (logits, ltrb maps, align logits, predicted IoUs, labels); template() / track() are the
    online API."""

    def __init__(self, backbone, neck, connect_model, align_head, search_size=255, score_size=25, batch_size=32,
                 training=True, necessary_pre=None, independent_trans=None, joint_trans=None, iou_prediction=False,
                 label_generator=None):
        super().__init__()
        self.features, self.neck = backbone, neck
        self.connect_model, self.align_head = connect_model, align_head
        self.zf = None
        self.search_size, self.score_size = search_size, score_size
        self.batch = batch_size if training else 1
        self.necessary_pre, self.independent_trans, self.joint_trans = necessary_pre, independent_trans, joint_trans
        self.iou_prediction, self.label_generator = iou_prediction, label_generator
        gx, gy = ocean.response_grid(score_size, search_size)
        self.register_buffer("grid_x", torch.tensor(gx, dtype=torch.float32), persistent=False)
        self.register_buffer("grid_y", torch.tensor(gy, dtype=torch.float32), persistent=False)

    # data decoding is the TransT wrapper's (transt_model.py:119-185 and ocean_model.py:215-280 are the same steps)
    _preprocess, _joint = TransT._preprocess, TransT._joint
    _decode_pairwise_data, _decode_sp_data = TransT._decode_pairwise_data, TransT._decode_sp_data
    _get_batch_size = staticmethod(TransT._get_batch_size)

    def feature_extractor(self, x):
        return self.features(x)

    def pred_to_image(self, bbox_pred):
        return ocean.pred_to_image(bbox_pred, self.grid_x, self.grid_y)

    def offset(self, boxes, featmap_sizes):
        return ocean.align_offsets(boxes, tuple(featmap_sizes[2:]), self.search_size)

    def template(self, z):
        _, zf = self.feature_extractor(z)
        if self.neck is not None:
            _, zf = self.neck(zf, crop=True)
        self.zf = zf.contiguous()

    def track(self, x):
        _, xf = self.feature_extractor(x)
        if self.neck is not None:
            xf = self.neck(xf)
        out = ocean.fusion_pipeline(self.connect_model, self.align_head, xf, self.zf, self.grid_x, self.grid_y,
                                    self.search_size, iou_prediction=False)
        return out["pred_logits"], out["pred_boxes"], out["align_logits"]

    def _branch(self, img, crop):
        _, f = self.feature_extractor(img)
        if self.neck is None:
            return f
        return self.neck(f, crop=True)[1] if crop else self.neck(f, crop=False)

    def _features_pair(self, template, search):
        """Backbone + neck of the template and the search crops (the reference runs them back to back, template
        first, ocean_model.py:414-419).  The template pass works on 63x63 ... 15x15 maps that cannot fill the chip, so
        on a GPU it runs on a second stream beside the search pass; the search pass leaves its BatchNorm
        running-statistics updates pending and they are applied after the join, which keeps the reference's order
        (template's update first)."""
        from . import fusion
        if not (fusion.TWO_STREAMS and search.is_cuda):
            return self._branch(template, True), self._branch(search, False)
        main = torch.cuda.current_stream()
        side = fusion.side_stream(search.device)
        ops.register_side_stream(side, main)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            zf = self._branch(template, True)
        with ops.defer_running_stats() as pending:
            xf = self._branch(search, False)
        main.wait_stream(side)
        zf.record_stream(main)
        pending.apply()
        return zf, xf

    def _fusion_pipeline(self, xf, zf, pred_to_img_tag=False, b_iou=0):
        return ocean.fusion_pipeline(self.connect_model, self.align_head, xf, zf, self.grid_x, self.grid_y, self.search_size,
                                     iou_prediction=self.iou_prediction, b_iou=b_iou, pred_to_img_tag=pred_to_img_tag)

    def _decode_output(self, outputs, batch_burnin=0):
        new = TensorDict()
        keys = ["pred_logits", "pred_boxes", "align_logits"] + (["pred_ious"] if self.iou_prediction else [])
        for key in keys:
            new[key + "_sup"] = outputs[key][:batch_burnin]
            if outputs[key].shape[0] > batch_burnin:
                new[key + "_unsup"] = outputs[key][batch_burnin:]
        return new

    def _refresh_anno(self, out, data, tag="search"):
        """ocean_model.py:345-372: labels from the (possibly augmented) boxes; align labels from the prediction."""
        if data is None:
            return out
        anno = data[tag + "_anno"]
        out[tag + "_anno"] = anno.detach()
        with torch.no_grad():
            cls_label = self.label_generator({"anno": anno}, tag="cls")
            reg_label, reg_weight = self.label_generator({"anno": anno}, tag="reg")
        sup = "_sup" if tag == "search" else "_unsup"
        align = self.label_generator({"bbox_pred": out["pred_boxes" + sup], "reg_label": reg_label,
                                      "reg_weight": reg_weight}, tag="align")
        if self.iou_prediction:
            with torch.no_grad():
                out["iou_weight" + sup] = self.label_generator({"anno": anno}, tag="iou_w")
        out["reg_label" + sup], out["reg_weight" + sup] = reg_label, reg_weight
        out["cls_label" + sup], out["align_label" + sup] = cls_label, align
        return out

    def forward(self, burnin_data=None, teacher_data=None, sp_data=None):
        burnin_data = self._decode_pairwise_data(burnin_data, "burnin", ["template", "search"])
        teacher_data = self._decode_pairwise_data(teacher_data, "sparse", ["key", "unlabeled"])
        sp_data = self._decode_sp_data(sp_data)
        templates, searches = [], []
        for data, t, s in ((teacher_data, "key", "unlabeled"), (burnin_data, "template", "search"), (sp_data, "relay", "aug")):
            if data is not None:
                searches.append(data[s + "_images"])
                templates.append(data[t + "_images"])
        template = (templates[0] if len(templates) == 1 else torch.cat(templates, dim=0)).detach()
        search = (searches[0] if len(searches) == 1 else torch.cat(searches, dim=0)).detach()
        zf, xf = self._features_pair(template, search)
        if teacher_data is not None:
            out = self._fusion_pipeline(xf, zf, pred_to_img_tag=True, b_iou=self._get_batch_size(teacher_data))
        else:
            nb = self._get_batch_size(burnin_data)
            out = self._decode_output(self._fusion_pipeline(xf, zf, pred_to_img_tag=True, b_iou=nb), batch_burnin=nb)
        out = self._refresh_anno(out, burnin_data, tag="search")
        out = self._refresh_anno(out, sp_data, tag="aug")
        return out


def ocean_resnet50(settings, network="student"):
    """ocean_model.py:439-489; weights are randomly initialised."""
    assert network in ["student", "teacher"], "Network construction type should be in 'student' or 'teacher'."
    aug = settings.aug_settings
    necessary_pre = NecessaryPreprocess(b_j=aug["jitter"], mean=settings.normalize_mean, std=settings.normalize_std)
    joint_trans = independent_trans = None
    if network != "teacher":
        joint_trans = JointTransform([GrayScale(gray_probability=aug["gray"])])
    if network != "teacher" and settings.strong_augmentation_tag:
        independent_trans = IndependentTransform([EllipseRotationAffine(
            max_angle=aug["angle"], lr_flip_prob=aug["lr_f_prob"], ud_flip_prob=aug["ud_f_prob"])])
    model = Ocean(backbone=build_backbone_ocean(settings),
                  neck=ocean.AdjustLayer(in_channels=settings.before_neck_dim, out_channels=settings.head_dim),
                  connect_model=ocean.box_tower(inchannels=settings.head_dim, outchannels=settings.head_dim,
                                                towernum=settings.head_tower_num, iou_prediction=settings.iou_prediction_tag),
                  align_head=ocean.AlignHead(settings.head_dim, settings.head_dim),
                  search_size=settings.search_sz, score_size=settings.output_sz, batch_size=settings.batch_size,
                  iou_prediction=settings.iou_prediction_tag, necessary_pre=necessary_pre, joint_trans=joint_trans,
                  independent_trans=independent_trans, label_generator=ocean.OceanLabelGenerator(settings))
    return model.to(torch.device(settings.device))


class OceanSparseActor(TranstSparseActor):
    """ocean_sparse_actor.py:10-243: modes "update" (EMA teacher: one streaming kernel over the flat parameter
    storage, as for SP-TransT), "inference" (teacher pseudo-labels), "burnin", "spsup".  The statistics are returned
    as device scalars (the reference calls .item() on each, ten host synchronisations per iteration)."""

    def __init__(self, settings, student_net, teacher_net, objective):
        super().__init__(settings, student_net, teacher_net, objective)
        self._window = None

    @torch.no_grad()
    def _inference_with_teacher(self, teacher_data):
        """:40-94: mix of classification and align scores, Hanning window penalty, arg-max cell, its box (corner
        form on the patch -> xywh, clamped) and its predicted IoU as confidence."""
        out = self.teacher_net(teacher_data=teacher_data)
        cs = self.settings.curriculum_settings
        ratio, penalty = cs["cls_ratio"], cs["window_penalty"]
        score = ratio * torch.sigmoid(out["pred_logits"]) + (1 - ratio) * torch.sigmoid(out["align_logits"])
        if self._window is None or self._window.device != score.device:
            h = torch.tensor(np.hanning(self.settings.output_sz), device=score.device, dtype=score.dtype)
            self._window = torch.outer(h, h)[None, None]
        B = score.shape[0]
        score = (penalty * self._window + (1 - penalty) * score).reshape(B, -1)
        best_score, best_idx = torch.max(score, dim=1)
        rows = torch.arange(B, device=score.device)
        boxes = out["bbox_pred_to_img"].reshape(B, 4, -1).transpose(1, 2)[rows, best_idx]
        bbox = box_xyxy_to_xywh(boxes)
        bbox[:, 0:2].clamp_min_(0)
        bbox[:, 2:4].clamp_min_(1)
        conf = out["pred_ious"].reshape(B, -1)[rows, best_idx]
        return TensorDict({"best_bbox": bbox.reshape(B, 4), "best_score": best_score, "best_conf": conf.reshape(B)})

    @staticmethod
    def _build_targets(data, outputs, tag):
        t = outputs[tag + "_anno"]
        h, w = data[tag + "_images"].shape[-2:]
        return {"boxes_cxcywh": torch.stack([(t[:, 0] + t[:, 2] / 2) / w, (t[:, 1] + t[:, 3] / 2) / h, t[:, 2] / w,
                                             t[:, 3] / h]).T,
                "boxes_orig_xyxy": torch.stack([t[:, 0], t[:, 1], t[:, 0] + t[:, 2], t[:, 1] + t[:, 3]]).T}

    def _weighted(self, loss_dict, suffix):
        wd = self.objective.weight_dict
        return sum(loss_dict[k] * wd[k + suffix] for k in loss_dict if k + suffix in wd)

    def _burnin_training(self, burnin_data):
        outputs = self.student_net(burnin_data=burnin_data)
        loss_dict = self.objective(outputs, self._build_targets(burnin_data, outputs, "search"), tag="sup")
        losses = self._weighted(loss_dict, "_sup")
        stats = {"Loss/total": losses.detach(), "Loss/ce": loss_dict["loss_ce"].detach(),
                 "Loss/bbox": loss_dict["loss_bbox"].detach(), "Loss/align": loss_dict["loss_align"].detach(),
                 "iou": loss_dict["iou"].detach()}
        if "loss_iou_pred" in loss_dict:
            stats["Loss/iou_pred"] = loss_dict["loss_iou_pred"].detach()
        return losses, stats

    def _sp_training(self, burnin_data, sp_data):
        outputs = self.student_net(burnin_data=burnin_data, sp_data=sp_data)
        n_sup, n_unsup = burnin_data["search_anno"].shape[0], sp_data["aug_anno"].shape[0]
        n = n_sup + n_unsup
        d_sup = self.objective({k: v for k, v in outputs.items() if k.endswith("_sup")},
                               self._build_targets(burnin_data, outputs, "search"), tag="sup")
        d_unsup = self.objective({k: v for k, v in outputs.items() if k.endswith("_unsup")},
                                 self._build_targets(sp_data, outputs, "aug"), tag="unsup")
        losses = (float(n_sup) * self._weighted(d_sup, "_sup") + float(n_unsup) * self._weighted(d_unsup, "_unsup")) / n
        stats = {"Loss/total": losses.detach(), "iou_sup": d_sup["iou"].detach(), "iou_unsup": d_unsup["iou"].detach()}
        for key, name in (("loss_ce", "Loss/ce"), ("loss_bbox", "Loss/bbox"), ("loss_align", "Loss/align"),
                          ("loss_iou_pred", "Loss/iou_pred"), ("iou", "iou")):
            if key not in d_sup:
                continue
            stats[name] = (d_sup[key].detach() if key not in d_unsup else
                           (float(n_sup) * d_sup[key].detach() + float(n_unsup) * d_unsup[key].detach()) / n)
        return losses, stats

    def __call__(self, mode, teacher_data=None, burnin_data=None, sp_data=None, keep_rate=0.998):
        if mode == "update":
            return self._update_teacher_model(keep_rate=keep_rate)
        if mode == "inference":
            return self._inference_with_teacher(teacher_data=teacher_data)
        if mode == "burnin":
            return self._burnin_training(burnin_data=burnin_data)
        if mode == "spsup":
            return self._sp_training(burnin_data=burnin_data, sp_data=sp_data)
        assert False, "No such operation for OceanSparseActor"
