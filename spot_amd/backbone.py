"""Feature extraction: ResNet-50 truncated at layer3 + sine position encoding.

Mirror of ltr/models/backbone/resnet.py:56-92,95-205 and
ltr/models/backbone/transt_backbone.py:55-106 with identical state-dict keys
(`backbone.0.body.conv1.weight`, `...layer3.5.bn3.running_var`, ...).

The 1x1 convolutions are batched GEMMs of libspot_hip.so on the NCHW maps (ops.conv1x1), the
stride-1 3x3 convolutions its Winograd F(4x4,3x3) path (ops.conv3x3), the 7x7 stem and the one
stride-2 3x3 convolution a patch matrix + GEMM (ops.conv_general), the max-pool its own kernel.  BatchNorm is
fused with the ReLU / residual add that follow it (batchnorm.hip); everything
downstream (projection, fusion, heads, decode, filter) is libspot_hip.so.
The nn.BatchNorm2d modules remain as parameter/buffer holders so state-dict
keys match the reference.
"""
import math

import torch
from torch import nn

from . import lib, ops
from .tensordict import NestedTensor


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, dilation=1):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=stride, padding=dilation, bias=False,
                               dilation=dilation)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, kernel_size=1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        cb = ops.conv1x1_bn_act
        if self._bn1_rides_in_conv2(x):
            # training-mode bn1 + relu are taken by conv2's input transform: conv1's raw output goes straight in
            if self.downsample is None:
                raw, residual = ops.conv1x1(x, self.conv1, passthrough=True)
            else:
                raw = ops.conv1x1(x, self.conv1)
                residual = cb(x, self.downsample[0], self.downsample[1])
            out = ops.batch_norm_act(ops.bn_relu_conv3x3(raw, self.bn1, self.conv2), self.bn2, relu=True)
            return cb(out, self.conv3, self.bn3, relu=True, residual=residual)
        if self.downsample is None:
            # identity branch handed through conv1: its gradient joins conv1's input gradient in one launch
            out, residual = cb(x, self.conv1, self.bn1, relu=True, passthrough=True)
        else:
            out = cb(x, self.conv1, self.bn1, relu=True)
            residual = cb(x, self.downsample[0], self.downsample[1]) if self.downsample is not None else x
        out = ops.conv3x3_bn_act(out, self.conv2, self.bn2, relu=True)
        return cb(out, self.conv3, self.bn3, relu=True, residual=residual)

    def _bn1_rides_in_conv2(self, x):
        # conv1 is 1x1 / stride 1: its output has x's spatial size
        return (self.conv1.stride == (1, 1) and x.dtype == torch.float32
                and ops.bn_relu_conv3x3_applies(self.bn1, self.conv2, x.shape[0], x.shape[2], x.shape[3], x.is_cuda))


class ResNet50Layer3(nn.Module):
    """conv1 .. layer3: layer2 stride 2, layer3 stride 1 with dilation 2 in its first
    block, total stride 8, 1024 output channels (resnet.py:103-112,152-167)."""

    def __init__(self, inplanes=64):
        super().__init__()
        self.inplanes = inplanes
        self.conv1 = nn.Conv2d(3, inplanes, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(inplanes)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(inplanes, 3)
        self.layer2 = self._make_layer(inplanes * 2, 4, stride=2)
        self.layer3 = self._make_layer(inplanes * 4, 6, stride=1, dilation=2)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2. / n))
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()

    def _make_layer(self, planes, blocks, stride=1, dilation=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * 4:
            downsample = nn.Sequential(
                nn.Conv2d(self.inplanes, planes * 4, kernel_size=1, stride=stride, bias=False),
                nn.BatchNorm2d(planes * 4))
        layers = [Bottleneck(self.inplanes, planes, stride, downsample, dilation=dilation)]
        self.inplanes = planes * 4
        for _ in range(1, blocks):
            layers.append(Bottleneck(self.inplanes, planes))
        return nn.Sequential(*layers)

    def forward(self, x):
        x = ops.max_pool_3x3_s2(ops.conv_bn_act(x, self.conv1, self.bn1, relu=True), self.maxpool)
        return self.layer3(self.layer2(self.layer1(x)))


class PositionEmbeddingSine(nn.Module):
    """ltr/models/neck/position_encoding.py:11-45 (normalize=True, scale 2*pi, T=10000), one kernel
    (spot_posenc_sine_f32) that writes token rows.

    For the all-False masks of full-size crops the encoding is a constant of (H, W): it is computed once per
    device and kept as [H*W, 2*num_pos_feats] rows.  Masked inputs are encoded per call.
    """

    def __init__(self, num_pos_feats=64, temperature=10000, normalize=True, scale=None):
        super().__init__()
        if not normalize:
            raise RuntimeError("spot_amd: PositionEmbeddingSine is built for normalize=True (the reference's setting)")
        self.num_pos_feats = num_pos_feats
        self.temperature = temperature
        self.normalize = normalize
        self.scale = 2 * math.pi if scale is None else scale
        self._cache = {}

    def _rows(self, mask, B, H, W, device):
        out = torch.empty((B, H * W, 2 * self.num_pos_feats), dtype=torch.float32, device=device)
        m = None if mask is None else mask.to(torch.uint8).contiguous()
        lib.call("spot_posenc_sine_f32", ops._p(m), ops._p(out), B, H, W, self.num_pos_feats, float(self.temperature),
                 float(self.scale), ops._st())
        return out

    def rows(self, H, W, device):
        """Encoding of an unmasked H x W map as [H*W, C] rows."""
        key = (H, W, str(device))
        if key not in self._cache:
            self._cache[key] = self._rows(None, 1, H, W, device)[0]
        return self._cache[key]

    def forward(self, tensor_list: NestedTensor):
        """[B, C, H, W] as the reference returns it."""
        mask = tensor_list.mask
        assert mask is not None
        B, H, W = mask.shape
        return self._rows(mask, B, H, W, mask.device).view(B, H, W, -1).permute(0, 3, 1, 2)


class BackboneBase(nn.Module):
    def __init__(self, backbone, num_channels):
        super().__init__()
        self.body = backbone
        self.num_channels = num_channels

    def forward(self, tensor_list: NestedTensor):
        x = self.body(tensor_list.tensors)
        m = tensor_list.mask
        assert m is not None
        mask = torch.nn.functional.interpolate(m[None].float(), size=x.shape[-2:]).to(torch.bool)[0]
        return {"layer3": NestedTensor(x, mask)}


class Backbone(BackboneBase):
    def __init__(self, output_layers=("layer3",), pretrained=False, frozen_layers=()):
        if pretrained:
            raise RuntimeError("spot_amd: pretrained ImageNet weights are a network download in the reference "
                               "(resnet.py:256-257); load a checkpoint with load_state_dict instead")
        super().__init__(ResNet50Layer3(), 1024)


class Joiner(nn.Sequential):
    def __init__(self, backbone, position_embedding):
        super().__init__(backbone, position_embedding)

    def forward(self, tensor_list: NestedTensor):
        xs = self[0](tensor_list)
        out, pos = [], []
        for _, x in xs.items():
            out.append(x)
            pos.append(self[1](x).to(x.tensors.dtype))
        return out, pos


def build_backbone(settings, backbone_pretrained=False, frozen_backbone_layers=()):
    position_embedding = PositionEmbeddingSine(settings.hidden_dim // 2, normalize=True)
    backbone = Backbone(pretrained=backbone_pretrained, frozen_layers=frozen_backbone_layers)
    model = Joiner(backbone, position_embedding)
    model.num_channels = backbone.num_channels
    return model
