"""The Siamese (SP-Ocean) head around the depthwise cross-correlation kernel: neck, feature encoders, correlation,
classification / regression / IoU towers, box decoding, label generator and criterion.

Mirrors the reference's classes (ltr/models/neck/ocean_fusion.py:14-235, ltr/models/tracking/ocean_model.py:79-191,
302-321, ltr/models/loss/ocean_label_gen.py:7-233, ltr/models/tracking/ocean_criterion.py:13-295) with the same
constructor arguments, parameter names (its checkpoints load) and outputs.  Every convolution runs as patch matrix +
GEMM, every BatchNorm / ReLU as the fused kernels, the three correlations and their weighted sum as one kernel, the
labels of a batch as one launch; the maps here are odd-sized (25x25, 29x29, 5x5), which those kernels handle by a
padded row pitch and an element-wise walk.  GPU only: the HIP library is required."""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import lib, ops
from .box_ops import box_xywh_to_xyxy  # noqa: F401  (part of the mirrored interface)
from .tensordict import TensorDict


def _conv_bn_relu(seq, x):
    """nn.Sequential of (conv, bn, relu) triples applied through the library's kernels."""
    mods = list(seq)
    assert len(mods) % 3 == 0
    for i in range(0, len(mods), 3):
        x = ops.conv_bn_act(x, mods[i], mods[i + 1], relu=True)
    return x


class AdjustLayer(nn.Module):
    """1x1 convolution + BatchNorm from the backbone's width to the head's; the template map is also cropped to its
    central 7x7 (ocean_fusion.py:14-32)."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.downsample = nn.Sequential(nn.Conv2d(in_channels, out_channels, kernel_size=1, bias=False),
                                        nn.BatchNorm2d(out_channels))

    def forward(self, x, crop=False):
        full = ops.batch_norm_act(ops.conv_general(x, self.downsample[0]), self.downsample[1])
        if not crop:
            return full
        if full.size(3) < 20:
            return full, full[:, :, 4:-4, 4:-4]
        raise RuntimeError("AdjustLayer: crop=True expects a template map narrower than 20 (as the reference, which "
                           "leaves the cropped map undefined otherwise)")


class matrix(nn.Module):
    """Three unpadded 3x3 conv-bn-relu encoders per input (template `_k`, search `_s`): plain, dilated (2, 1), dilated
    (1, 2) (ocean_fusion.py:117-171)."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        for name, cin, dil in (("matrix11", in_channels, 1), ("matrix12", out_channels, (2, 1)),
                               ("matrix21", out_channels, (1, 2))):
            for side in ("_k", "_s"):
                self.add_module(name + side, nn.Sequential(
                    nn.Conv2d(cin, out_channels, kernel_size=3, bias=False, dilation=dil),
                    nn.BatchNorm2d(out_channels), nn.ReLU(inplace=True)))

    def forward(self, z, x):
        zs = [_conv_bn_relu(getattr(self, n + "_k"), z) for n in ("matrix11", "matrix12", "matrix21")]
        xs = [_conv_bn_relu(getattr(self, n + "_s"), x) for n in ("matrix11", "matrix12", "matrix21")]
        return zs, xs


def xcorr_depthwise(x, kernel):
    """Per-(clip, channel) valid cross-correlation (ocean_fusion.py:238-247)."""
    return ops.xcorr_depthwise(x, kernel)


class GroupDW(nn.Module):
    """softmax(weight)-weighted sum of the three branch correlations (ocean_fusion.py:211-235), one kernel."""

    def __init__(self, in_channels=256):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(3))

    def forward(self, z, x):
        return ops.groupdw(z, x, self.weight)


class box_tower(nn.Module):
    """Encoders, correlation, regression / classification / IoU towers and their 3x3 prediction heads
    (ocean_fusion.py:35-114).  Returns (ltrb distances, classification logits, correlated classification features,
    regression tower features, extra outputs)."""

    def __init__(self, inchannels=512, outchannels=256, towernum=1, iou_prediction=True):
        super().__init__()
        self.iou_prediction = iou_prediction
        self.cls_encode = matrix(in_channels=inchannels, out_channels=outchannels)
        self.reg_encode = matrix(in_channels=inchannels, out_channels=outchannels)
        self.cls_dw = GroupDW(in_channels=inchannels)
        self.reg_dw = GroupDW(in_channels=inchannels)

        def tower():
            layers = []
            for _ in range(towernum):
                layers += [nn.Conv2d(outchannels, outchannels, kernel_size=3, stride=1, padding=1),
                           nn.BatchNorm2d(outchannels), nn.ReLU()]
            return nn.Sequential(*layers)

        self.add_module("bbox_tower", tower())
        self.add_module("cls_tower", tower())
        self.bbox_pred = nn.Conv2d(outchannels, 4, kernel_size=3, stride=1, padding=1)
        self.cls_pred = nn.Conv2d(outchannels, 1, kernel_size=3, stride=1, padding=1)
        self.adjust = nn.Parameter(0.1 * torch.ones(1))
        self.bias = nn.Parameter(torch.ones(1, 4, 1, 1))
        if self.iou_prediction:
            self.add_module("iou_pred_tower", tower())
            self.iou_pred = nn.Conv2d(outchannels, 1, kernel_size=3, stride=1, padding=1)

    def _cls_branch(self, search, kernel):
        cls_z, cls_x = self.cls_encode(kernel, search)
        cls_dw = self.cls_dw(cls_z, cls_x)
        c = _conv_bn_relu(self.cls_tower, cls_dw)
        return 0.1 * ops.conv_general(c, self.cls_pred), cls_dw

    def forward(self, search, kernel, iou_prediction=True, batch_iou=0):
        # The classification and the regression branch share nothing but their inputs (own encoders, correlation
        # weights, towers, BatchNorm layers): on a GPU the classification branch runs on a second stream beside
        # the regression branch, whose 25x25 ... 45x45 maps alone do not fill the chip at small batches.
        from . import fusion
        two = fusion.TWO_STREAMS and search.is_cuda
        if two:
            main = torch.cuda.current_stream()
            side = fusion.side_stream(search.device)
            ops.register_side_stream(side, main)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                cls, cls_dw = self._cls_branch(search, kernel)
        reg_z, reg_x = self.reg_encode(kernel, search)
        reg_dw = self.reg_dw(reg_z, reg_x)
        x_reg = _conv_bn_relu(self.bbox_tower, reg_dw)
        x = torch.exp(torch.clamp_max(self.adjust * ops.conv_general(x_reg, self.bbox_pred) + self.bias, max=4.7))
        if two:
            main.wait_stream(side)
            cls.record_stream(main)
            cls_dw.record_stream(main)
        else:
            cls, cls_dw = self._cls_branch(search, kernel)
        extra = TensorDict()
        if self.iou_prediction and iou_prediction and batch_iou > 0:
            t = _conv_bn_relu(self.iou_pred_tower, reg_dw.detach()[:batch_iou])
            extra["pred_ious"] = ops.conv_general(t, self.iou_pred).sigmoid()
        return x, cls, cls_dw, x_reg, extra


class DeformConv(nn.Module):
    """Bias-free deformable convolution v1 with externally supplied offsets (dcn/deform_conv.py:190-236)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 deformable_groups=1, bias=False):
        super().__init__()
        assert not bias and groups == 1
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size = (kernel_size, kernel_size) if isinstance(kernel_size, int) else tuple(kernel_size)
        self.stride, self.padding, self.dilation = stride, padding, dilation
        self.groups, self.deformable_groups = groups, deformable_groups
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels // groups, *self.kernel_size))
        bound = 1.0 / float(np.sqrt(in_channels * self.kernel_size[0] * self.kernel_size[1]))
        nn.init.uniform_(self.weight, -bound, bound)

    def forward(self, x, offset):
        return ops.deform_conv(x, offset, self.weight, self.stride, self.padding, self.dilation, self.groups,
                               self.deformable_groups)


class AdaptiveConv(nn.Module):
    """3x3 deformable convolution whose offsets come from predicted boxes, [N, H*W, 18] (ocean_fusion.py:174-192)."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv = DeformConv(in_channels, out_channels, 3, padding=1)

    def forward(self, x, offset):
        N, _, H, W = x.shape
        assert offset is not None and H * W == offset.shape[1]
        return self.conv(x, offset.permute(0, 2, 1).reshape(N, -1, H, W))


class AlignHead(nn.Module):
    """Classification on box-aligned features: adaptive convolution, ReLU, 1x1 convolution (ocean_fusion.py:195-208)."""

    def __init__(self, in_channels, feat_channels):
        super().__init__()
        self.rpn_conv = AdaptiveConv(in_channels, feat_channels)
        self.rpn_cls = nn.Conv2d(feat_channels, 1, 1)
        self.relu = nn.ReLU(inplace=True)

    def forward(self, x, offset):
        return ops.conv_general(torch.relu(self.rpn_conv(x, offset)), self.rpn_cls)


def fusion_pipeline(connect_model, align_head, xf, zf, grid_x, grid_y, search_size, iou_prediction=True, b_iou=0,
                    pred_to_img_tag=False):
    """Correlation head, box decoding, align offsets and aligned classification for neck features xf / zf
    (ocean_model.py:302-321)."""
    bbox_pred, cls_pred, _, reg_feature, extra = connect_model(xf, zf, iou_prediction=iou_prediction, batch_iou=b_iou)
    boxes = pred_to_image(bbox_pred, grid_x, grid_y)
    offsets = align_offsets(boxes.permute(0, 2, 3, 1).reshape(boxes.shape[0], -1, 4), bbox_pred.shape[2:], search_size)
    out = {"pred_logits": cls_pred, "pred_boxes": bbox_pred, "align_logits": align_head(reg_feature, offsets)}
    if iou_prediction and "pred_ious" in extra:
        out["pred_ious"] = extra["pred_ious"]
    if pred_to_img_tag:
        out["bbox_pred_to_img"] = boxes.detach()
    return out


def response_grid(score_size, search_size, stride=8):
    """Centres of the response-map cells on the search patch, as (x [S,S], y [S,S]) float64 arrays: cell (r, c) sits
    at ((c - S//2) stride + search//2, (r - S//2) stride + search//2) (ocean_model.py:79-96)."""
    idx = np.arange(0, score_size) - np.floor(float(score_size // 2))
    gx, gy = np.meshgrid(idx, idx)
    return gx * stride + search_size // 2, gy * stride + search_size // 2


def pred_to_image(bbox_pred, grid_x, grid_y):
    """ltrb distances [B,4,S,S] -> corner boxes on the search patch [B,4,S,S] (ocean_model.py:104-121)."""
    gx, gy = grid_x.to(bbox_pred.device), grid_y.to(bbox_pred.device)
    return torch.stack([gx - bbox_pred[:, 0], gy - bbox_pred[:, 1], gx + bbox_pred[:, 2], gy + bbox_pred[:, 3]], dim=1)


def align_offsets(boxes, feat_hw, search_size, stride=8, ks=3):
    """Sampling offsets of the aligned (deformable) convolution from predicted boxes [B, S*S, 4]: the 3x3 taps are
    spread over the box and centred on it (ocean_model.py:123-191), batched.  Returns [B, S*S, 2*ks*ks] in the
    (y0, x0, y1, x1, ...) order."""
    feat_h, feat_w = feat_hw
    dev, dt = boxes.device, boxes.dtype
    pad = (ks - 1) // 2
    idx = torch.arange(-pad, pad + 1, dtype=dt, device=dev)
    yy, xx = torch.meshgrid(idx, idx, indexing="ij")
    xx, yy = xx.reshape(-1), yy.reshape(-1)
    w = ((boxes[..., 2] - boxes[..., 0] + 1) / stride) / (ks - 1) - 1
    h = ((boxes[..., 3] - boxes[..., 1] + 1) / stride) / (ks - 1) - 1
    cx = ((boxes[..., 0] + boxes[..., 2]) * 0.5 - search_size * 0.5) / stride + feat_w // 2
    cy = ((boxes[..., 1] + boxes[..., 3]) * 0.5 - search_size * 0.5) / stride + feat_h // 2
    gy, gx = torch.meshgrid(torch.arange(0, feat_h, device=dev), torch.arange(0, feat_w, device=dev), indexing="ij")
    cx = cx - gx.reshape(-1).type_as(cx)
    cy = cy - gy.reshape(-1).type_as(cy)
    off_x = w[..., None] * xx + cx[..., None]
    off_y = h[..., None] * yy + cy[..., None]
    return torch.stack([off_y, off_x], dim=-1).reshape(boxes.shape[0], boxes.shape[1], -1)


class OceanLabelGenerator(nn.Module):
    """Classification / regression / IoU-weight / align labels from boxes (ocean_label_gen.py:7-233).  The first three
    come from one kernel launch for the whole batch; regression labels are float64 like the reference's (its grid is
    a float64 tensor, so every subtraction promotes)."""

    def __init__(self, settings):
        super().__init__()
        self.stride, self.output_sz = settings.stride, settings.output_sz
        self.batch, self.search_sz = settings.batch_size, settings.search_sz

    def forward(self, data_dict=None, tag="cls"):
        if tag == "align":
            return self._construct_align_labels(data_dict["bbox_pred"], data_dict["reg_label"], data_dict["reg_weight"])
        if tag not in ("cls", "reg", "iou_w"):
            return None
        anno = ops._chk(data_dict["anno"].detach().contiguous().float())
        B, S, dev = anno.shape[0], self.output_sz, anno.device
        cls = reg = reg_w = iou_w = None
        if tag == "cls":
            cls = torch.empty((B, S, S), dtype=torch.float32, device=dev)
        elif tag == "reg":
            reg = torch.empty((B, S, S, 4), dtype=torch.float64, device=dev)
            reg_w = torch.empty((B, S, S), dtype=torch.float32, device=dev)
        else:
            iou_w = torch.empty((B, S, S), dtype=torch.float32, device=dev)
        lib.call("spot_ocean_labels_f32", ops._p(anno), ops._p(cls), ops._p(reg), ops._p(reg_w), ops._p(iou_w), B, S,
                 int(self.stride), int(self.search_sz), 0.2, 2, ops._st())
        return {"cls": cls, "reg": (reg, reg_w), "iou_w": iou_w}[tag]

    def _construct_align_labels(self, bbox_pred, reg_label, reg_weight):
        """IoU of predicted and label ltrb distances, zero outside the box (ocean_label_gen.py:68-104); differentiable
        with respect to the prediction, as in the reference."""
        return _ltrb_iou(bbox_pred.permute(0, 2, 3, 1), reg_label, reg_weight)


def _ltrb_iou(pred, target, weight):
    area_t = (target[..., 0] + target[..., 2]) * (target[..., 1] + target[..., 3])
    area_p = (pred[..., 0] + pred[..., 2]) * (pred[..., 1] + pred[..., 3])
    w_i = torch.min(pred[..., 0], target[..., 0]) + torch.min(pred[..., 2], target[..., 2])
    h_i = torch.min(pred[..., 3], target[..., 3]) + torch.min(pred[..., 1], target[..., 1])
    inter = w_i * h_i
    ious = (inter + 1.0) / (area_t + area_p - inter + 1.0)
    return torch.clamp(torch.abs(weight * ious), max=1, min=0)


class SetCriterionOcean(nn.Module):
    """Losses of the Siamese head (ocean_criterion.py:13-284): class-balanced BCE on the classification map, BCE on the
    align map, IoU loss on the positive cells, L1 on the predicted IoU.  Written with masks instead of index
    selection, so nothing synchronises with the host; the reference's edge cases are kept (a class with exactly one
    cell contributes 0, ocean_criterion.py:46-47; NaN / inf losses become 0, :139-143,205,223)."""

    def __init__(self, settings, losses):
        super().__init__()
        self.weight_dict = settings.weight_dict
        self.losses = losses
        self.search_sz, self.stride = settings.search_sz, settings.stride
        self.score_size, self.batch = settings.output_sz, settings.batch_size
        self.register_buffer("empty_weight", torch.ones(2) * 0.5)

    @staticmethod
    def _finite_or_zero(v):
        return torch.where(torch.isfinite(v), v, torch.zeros_like(v))

    def _weighted_BCE(self, pred, label):
        pred, label = pred.reshape(-1), label.reshape(-1)
        bce = F.binary_cross_entropy_with_logits(pred, label, reduction="none")
        total = 0
        for value in (1, 0):
            m = label.eq(value)
            n = m.sum()
            part = torch.where(m, bce, torch.zeros_like(bce)).sum() / n
            total = total + 0.5 * torch.where(n == 1, torch.zeros_like(part), part)
        return total

    def add_iouloss(self, bbox_pred, reg_target, reg_weight, return_iou=False):
        pred = bbox_pred.permute(0, 2, 3, 1).reshape(-1, 4)
        target = reg_target.reshape(-1, 4)
        m = reg_weight.reshape(-1) > 0
        area_t = (target[:, 0] + target[:, 2]) * (target[:, 1] + target[:, 3])
        area_p = (pred[:, 0] + pred[:, 2]) * (pred[:, 1] + pred[:, 3])
        w_i = torch.min(pred[:, 0], target[:, 0]) + torch.min(pred[:, 2], target[:, 2])
        h_i = torch.min(pred[:, 3], target[:, 3]) + torch.min(pred[:, 1], target[:, 1])
        inter = w_i * h_i
        ious = (inter + 1.0) / (area_t + area_p - inter + 1.0)
        n = m.sum()
        loss = torch.where(m, -torch.log(ious), torch.zeros_like(ious)).sum() / n
        iou = torch.where(m, ious, torch.zeros_like(ious)).sum() / n
        bad = ~torch.isfinite(loss)
        loss = torch.where(bad, torch.zeros_like(loss), loss)
        if return_iou:
            return loss, torch.where(bad, torch.zeros_like(iou), iou)
        return loss

    def loss_cls(self, outputs, targets, tag):
        return {"loss_ce": self._finite_or_zero(self._weighted_BCE(outputs["pred_logits_" + tag],
                                                                   outputs["cls_label_" + tag]))}

    def loss_align_cls(self, outputs, targets, tag):
        label = outputs["align_label_" + tag].unsqueeze(1)
        logits = outputs["align_logits_" + tag]
        return {"loss_align": self._finite_or_zero(F.binary_cross_entropy_with_logits(logits.to(label.dtype), label))}

    def loss_boxes(self, outputs, targets, tag):
        reg_weight = outputs["reg_weight_" + tag]
        loss_bbox, ious = self.add_iouloss(outputs["pred_boxes_" + tag], outputs["reg_label_" + tag], reg_weight,
                                           return_iou=True)
        losses = {"loss_bbox": loss_bbox, "iou": ious}
        if "pred_ious_" + tag in outputs:
            iou_weight = outputs["iou_weight_" + tag]
            err = (outputs["align_label_" + tag].detach() - outputs["pred_ious_" + tag].squeeze(1)).abs()
            outer = (err * (iou_weight - reg_weight)).sum() / iou_weight.sum()
            inner = (err * reg_weight).sum() / iou_weight.sum()
            losses["loss_iou_pred"] = outer + inner
        return losses

    def get_loss(self, loss, outputs, targets, tag):
        table = {"cls": self.loss_cls, "boxes": self.loss_boxes, "align_cls": self.loss_align_cls}
        assert loss in table, "do you really want to compute {} loss?".format(loss)
        return table[loss](outputs, targets, tag=tag)

    def forward(self, outputs, targets, tag="sup"):
        losses = {}
        for loss in self.losses:
            losses.update(self.get_loss(loss, outputs, targets, tag=tag))
        return losses


def ocean_loss(settings):
    """ocean_criterion.py:287-295."""
    return SetCriterionOcean(settings=settings, losses=["cls", "align_cls", "boxes"]).to(torch.device(settings.device))
