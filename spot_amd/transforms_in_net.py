"""In-network augmentation of the student's crops: host mirror of
ltr/models/pre_input/transforms_in_net.py (IndependentTransform :7-60, JointTransform :63-118,
EllipseRotationAffine :121-359, GrayScale :362-398) with the same class names, constructor arguments and
forward signatures.

The reference transforms a batch sample by sample in Python (a flip, an affine grid, a grid_sample and a masked
fill per sample; a per-sample channel scaling for the grey-scale transform).  Here the random draws are made for
the whole batch with the same torch calls in the same order, and one kernel launch per transform applies them:
rotate + flip + fill of the images, the ellipse box transform, the channel weighting.

Reference behaviours kept on purpose:
  * GrayScale multiplies the three (already normalised) channels by 0.2989 / 0.5870 / 0.1140 without summing
    them, and replaces the fill colour by its weighted sum (:393-396); it ignores `masks`.
  * EllipseRotationAffine flips the box centre of every sample for which a left/right flip was drawn, but uses the
    transformed box only where masks[i] is set (:183-185,213).
"""
import math

import torch
from torch import nn

from . import lib, ops


class IndependentTransform(nn.Module):
    """Transforms applied to one patch type on its own (transforms_in_net.py:7-60)."""

    def __init__(self, aug_list):
        super().__init__()
        self.independent_augs = nn.ModuleList(list(aug_list))

    @torch.no_grad()
    def forward(self, data, tag="template", masks=None):
        mask = data["masks"] if masks is None else masks
        image, bbox, avg = data["{}_images".format(tag)], data["{}_anno".format(tag)], data["{}_avg".format(tag)]
        for aug in self.independent_augs:
            image, bbox, avg, mask = aug(image, bbox, avg, mask)
        return image, bbox


class JointTransform(nn.Module):
    """Transforms that share their random draws between the template and the search patch (:63-118)."""

    def __init__(self, aug_list):
        super().__init__()
        self.joint_augs = nn.ModuleList(list(aug_list))

    @torch.no_grad()
    def forward(self, data, tags=None, masks=None):
        if tags is None:
            tags = ["template", "search"]
        image_groups = [data[s + "_images"] for s in tags]
        anno_groups = [data[s + "_anno"] for s in tags]
        avg_groups = [data[s + "_avg"] for s in tags]
        mask = data["masks"] if masks is None else masks
        for aug in self.joint_augs:
            image_groups, anno_groups, avg_groups, mask = aug(image_groups, anno_groups, avg_groups, mask)
        return image_groups, anno_groups, avg_groups


def inverse_rotation_table():
    """[721, 6] float32: the inverse affine matrix of a rotation by a = -360..360 degrees about the point (0.5, 0.5)
    of the centred pixel frame, as _get_inverse_affine_matrix(center=[0.5, 0.5], angle=-a, translate=[0, 0],
    scale=1, shear=[0, 0]) yields it (:247-275; double precision on the host, then float32)."""
    rows = []
    for a in range(-360, 361):
        rot = math.radians(-a)
        ca, sa = math.cos(rot), math.sin(rot)          # shear = 0: a = cos, b = -sin, c = sin, d = cos
        m = [ca, sa, 0.0, -sa, ca, 0.0]                # [d, -b, 0, -c, a, 0]
        m[2] += m[0] * -0.5 + m[1] * -0.5
        m[5] += m[3] * -0.5 + m[4] * -0.5
        m[2] += 0.5
        m[5] += 0.5
        rows.append(m)
    return torch.tensor(rows, dtype=torch.float32)


class EllipseRotationAffine(nn.Module):
    """Probabilistic left/right flip, up/down flip (as a 180 degree turn) and a rotation by up to max_angle degrees,
    the box following through its inscribed ellipse (:121-216)."""

    def __init__(self, max_angle=0, lr_flip_prob=0.0, ud_flip_prob=0.0):
        super().__init__()
        self.max_angle = abs(int(max_angle))
        self.lr_flip_prob = lr_flip_prob
        self.ud_flip_prob = ud_flip_prob
        self._tables = {}

    def sample(self, batch, device):
        """The reference's draws, same calls in the same order (:161-166): (angle [B] float32, lr_flip [B] bool)."""
        base_angle = 180 * torch.where(torch.rand([batch], device=device) <= self.ud_flip_prob,
                                       torch.ones([batch], device=device), torch.zeros([batch], device=device))
        angle = torch.randint(-self.max_angle, self.max_angle + 1, [batch], device=device) + base_angle
        lr_flip_tag = torch.rand([batch], device=device) <= self.lr_flip_prob
        return angle, lr_flip_tag

    def apply(self, image, bbox, avg, mask, angle, lr_flip_tag):
        """The deterministic part: given the draws, transform images [B,C,H,W] and xywh boxes [B,4]."""
        image = ops._chk(image.contiguous(), name="image")
        B, Cn, H, W = image.shape
        dev = image.device
        if dev not in self._tables:
            self._tables[dev] = inverse_rotation_table().to(dev)
        angle = angle.to(torch.float32).contiguous()
        lr = lr_flip_tag.to(torch.uint8).contiguous()
        ap = mask.to(device=dev, dtype=torch.uint8).contiguous()
        fill = avg.to(torch.float32).contiguous()
        out = torch.empty_like(image)
        lib.call("spot_rotate_flip_nearest_f32", ops._p(image), ops._p(out), ops._p(self._tables[dev]), ops._p(angle),
                 ops._p(lr), ops._p(ap), ops._p(fill), B, Cn, H, W, ops._st())
        new_bbox = torch.empty((B, 4), dtype=torch.float32, device=dev)
        lib.call("spot_ellipse_rotate_boxes_f32", ops._p(bbox.to(torch.float32).contiguous()), ops._p(angle), ops._p(lr),
                 ops._p(ap), ops._p(new_bbox), B, W, H, ops._st())
        return out, new_bbox

    def forward(self, image, bbox, avg, mask):
        angle, lr_flip_tag = self.sample(image.shape[0], image.device)
        image, bbox = self.apply(image, bbox, avg, mask, angle, lr_flip_tag)
        return image, bbox, avg, mask


class GrayScale(nn.Module):
    """Grey-scale transform shared by the template and the search patch of a pair (:362-398)."""

    def __init__(self, gray_probability=0.05):
        super().__init__()
        self.gray_prob = gray_probability
        self._weights = {}

    def sample(self, batch, device):
        return torch.rand([batch], device=device) <= self.gray_prob          # :385

    def apply(self, image_groups, avg_groups, gray_tag):
        dev = image_groups[0].device
        if dev not in self._weights:
            self._weights[dev] = torch.tensor([0.2989, 0.5870, 0.1140], device=dev).float()
        w = self._weights[dev]
        scale = torch.where(gray_tag[:, None], w[None, :], torch.ones((), device=dev)).contiguous()     # [B, 3]
        new_avgs = []
        for img, avg in zip(image_groups, avg_groups):
            ops._chk(img, name="image")
            B, Cn, H, Wd = img.shape
            lib.call("spot_channel_scale_f32", ops._p(img), ops._p(scale), B * Cn, H * Wd, ops._st())
            a = avg.to(torch.float32)
            cvt = a[:, 0] * w[0] + a[:, 1] * w[1] + a[:, 2] * w[2]
            new_avgs.append(torch.where(gray_tag[:, None], cvt[:, None].expand(-1, 3), a))
        return image_groups, new_avgs

    def forward(self, image_groups, bbox_groups, avg_groups, masks):
        gray_tag = self.sample(image_groups[0].shape[0], image_groups[0].device)
        image_groups, avg_groups = self.apply(image_groups, avg_groups, gray_tag)
        return image_groups, bbox_groups, avg_groups, masks
