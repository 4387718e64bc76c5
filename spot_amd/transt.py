"""SP-TransT model on the HIP kernels: host mirror of
ltr/models/tracking/transt_model.py (class TransT, MLP, transt_resnet50).

Same constructor arguments, forward modes (burnin_data / teacher_data / sp_data),
online API (template / track) and state-dict keys as the reference.  uint8 crops
are normalised by the preprocess kernel, ResNet features are projected straight
from NCHW into batch-first tokens by the GEMM kernel (no permute pass), fusion
and heads run in libspot_hip.so.

The in-network augmentations (brightness jitter, GrayScale, EllipseRotationAffine) are batched device
kernels (transforms_in_net.py); their random draws come from torch's device generator.
"""
import torch
from torch import nn

from . import ops
from .backbone import build_backbone
from .fusion import build_featurefusion_network, _LinearParams
from .tensordict import NestedTensor, TensorDict
from .transforms_in_net import EllipseRotationAffine, GrayScale, IndependentTransform, JointTransform


class NecessaryPreprocess(nn.Module):
    """NormJitter + Normalize (ltr/models/pre_input/necessary_preprocess.py:6-91) as one kernel."""

    def __init__(self, b_j=0.2, mean=None, std=None, inplace=True):
        super().__init__()
        self.brightness_jitter = b_j
        self.mean = list(mean)
        self.std = list(std)
        self._dev_stats = {}

    def _stats(self, device):
        # cached per device: building them per call is a blocking pageable host-to-device copy
        if device not in self._dev_stats:
            self._dev_stats[device] = (torch.as_tensor(self.mean, dtype=torch.float32, device=device),
                                       torch.as_tensor(self.std, dtype=torch.float32, device=device))
        return self._dev_stats[device]

    def forward(self, data, tag="template"):
        images, avg = data[tag + "_images"], data[tag + "_avg"]
        B = images.shape[0]
        if images.dtype != torch.uint8:
            raise RuntimeError("spot_amd: %s_images must be uint8 crops" % tag)
        if self.brightness_jitter > 0:
            f = torch.zeros([B], device=images.device).uniform_(max(0, 1 - self.brightness_jitter),
                                                                 1 + self.brightness_jitter) / 255.0
        else:
            f = None
        im = ops.preprocess_u8(images.contiguous(), f, self.mean, self.std)
        fa = (f if f is not None else torch.full([B], 1.0 / 255.0, device=images.device))[:, None]
        m, s = self._stats(avg.device)
        avg_out = ((avg.float() * fa).clamp(0.0, 1.0) - m) / s
        return im, avg_out


class InputProjFn(torch.autograd.Function):
    """1x1 conv 1024 -> 256 (transt_model.py:49,274-275) from NCHW features to [B, HW, C] tokens."""

    @staticmethod
    def forward(ctx, feat, weight, bias):
        feat = feat.contiguous()
        B, Cin, H, W = feat.shape
        Cout, HW = weight.shape[0], H * W
        w2 = weight.view(Cout, Cin)
        out = torch.empty((B, HW, Cout), dtype=torch.float32, device=feat.device)
        ops.gemm(feat, w2, out, HW, Cout, Cin, transA=True, lda=HW, ldb=Cin, ldc=Cout, batch=B,
                 strideA=Cin * HW, strideC=HW * Cout, bias=bias)
        ctx.save_for_backward(feat, weight)
        ctx.params = (weight, bias)
        ops._sink_expect(weight if ctx.needs_input_grad[1] else None, bias if ctx.needs_input_grad[2] else None)
        return out

    @staticmethod
    def backward(ctx, dtok):
        feat, weight = ctx.saved_tensors
        B, Cin, H, W = feat.shape
        Cout, HW = weight.shape[0], H * W
        w2 = weight.view(Cout, Cin)
        dtok = dtok.contiguous()
        dfeat = dw = db = None
        if ctx.needs_input_grad[0]:
            dfeat = torch.empty_like(feat)
            ops.gemm(w2, dtok, dfeat, Cin, HW, Cout, transA=True, lda=Cin, ldb=Cout, ldc=HW, batch=B,
                     strideB=HW * Cout, strideC=Cin * HW)
        pw, pb = ctx.params
        if ctx.needs_input_grad[1]:
            sk = ops._sink(pw)
            dw = sk.view(Cout, Cin) if sk is not None else torch.zeros_like(w2)
            ops.gemm(dtok, feat, dw, Cout, Cin, B * HW, transA=True, lda=Cout, ldb=HW, ldc=Cin, kseg=HW,
                     segA=HW * Cout, segB=Cin * HW, splitk=ops._splitk_for(Cout, Cin, B * HW), accumulate=True)
            dw = None if sk is not None else dw.view_as(weight)
        if ctx.needs_input_grad[2]:
            sk = ops._sink(pb)
            db = ops.colsum(dtok.view(-1, Cout), out=sk)
            db = None if sk is not None else db
        ops._sink_done(pw, pb)
        return dfeat, dw, db


class MLP(nn.Module):
    """Very simple multi-layer perceptron (transt_model.py:296-310)."""

    def __init__(self, input_dim, hidden_dim, output_dim, num_layers):
        super().__init__()
        self.num_layers = num_layers
        h = [hidden_dim] * (num_layers - 1)
        self.layers = nn.ModuleList(_LinearParams(n, k) for n, k in zip([input_dim] + h, h + [output_dim]))

    def forward(self, x):
        return ops.mlp(x, self.layers)


class _Conv1x1Params(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        ref = nn.Conv2d(cin, cout, kernel_size=1)
        self.weight, self.bias = ref.weight, ref.bias


class TransT(nn.Module):
    def __init__(self, backbone, featurefusion_network, num_classes=1, iou_prediction=False,
                 necessary_pre=None, independent_trans=None, joint_trans=None):
        super().__init__()
        self.featurefusion_network = featurefusion_network
        hidden_dim = featurefusion_network.d_model
        self.class_embed = MLP(hidden_dim, hidden_dim, num_classes + 1, 3)
        self.bbox_embed = MLP(hidden_dim, hidden_dim, 4, 3)
        self.input_proj = _Conv1x1Params(backbone.num_channels, hidden_dim)
        self.backbone = backbone
        self.necessary_pre = necessary_pre
        self.independent_trans = independent_trans
        self.joint_trans = joint_trans
        self.iou_embed = MLP(hidden_dim, hidden_dim, 1, 3) if iou_prediction else None

    # -- feature extraction ------------------------------------------------
    def _features(self, img):
        """normalised crops [B,3,H,W] -> (tokens [B,HW,C], pos rows [HW,C])."""
        feat = self.backbone[0].body(img)
        tok = InputProjFn.apply(feat, self.input_proj.weight, self.input_proj.bias)
        pos = self.backbone[1].rows(feat.shape[2], feat.shape[3], feat.device)
        return tok, pos

    def _features_pair(self, search, template):
        """Search and template feature extraction (the reference runs them back to back, search first).  On a
        GPU the template pass, whose 16x16..64x64 maps cannot fill the chip, runs on a second stream beside the
        search pass; its BatchNorm running-statistics updates are applied afterwards so the buffers end up as
        in the sequential order."""
        from . import fusion
        if not (fusion.TWO_STREAMS and search.is_cuda):
            return self._features(search), self._features(template)
        main = torch.cuda.current_stream()
        side = fusion.side_stream(search.device)
        ops.register_side_stream(side, main)
        side.wait_stream(main)
        with torch.cuda.stream(side), ops.defer_running_stats() as pending:
            out_t = self._features(template)
        out_s = self._features(search)
        main.wait_stream(side)
        out_t[0].record_stream(main)
        pending.apply()
        return out_s, out_t

    def _heads(self, hs, use_iou=False, b_iou=0):
        out = {"pred_logits": self.class_embed(hs), "pred_boxes": self.bbox_embed(hs).sigmoid()}
        if self.iou_embed is not None and use_iou:
            out["pred_ious"] = self.iou_embed(hs[:b_iou].detach()).sigmoid()
        return out

    # -- online tracking API (transt_model.py:86-117) -----------------------
    def template(self, z):
        if isinstance(z, NestedTensor):
            z = z.tensors
        self.zf, self.pos_template = self._features(z)
        self.mask_template = None

    def track(self, search):
        if isinstance(search, NestedTensor):
            search = search.tensors
        xf, pos_search = self._features(search)
        hs = self.featurefusion_network.forward_tokens(self.zf, xf, self.pos_template, pos_search)
        return self._heads(hs)

    # -- training / pseudo-labelling forward (transt_model.py:229-293) -------
    @staticmethod
    def _get_batch_size(data):
        for k in ("template_anno", "relay_anno", "key_anno"):
            if data is not None and k in data:
                return data[k].shape[0]
        return 0

    def _preprocess(self, data, prefixes):
        if data is not None:
            for p in prefixes:
                data[p + "_images"], data[p + "_avg"] = self.necessary_pre(data, tag=p)
        return data

    def _joint(self, data, tags):
        """Joint transform of a (template, search) pair of groups, e.g. GrayScale (transt_model.py:128-133,157-165)."""
        image_groups, anno_groups, avg_groups = self.joint_trans(data, tags=tags, masks=data["masks"])
        for i, tag in enumerate(tags):
            data[tag + "_images"], data[tag + "_anno"], data[tag + "_avg"] = image_groups[i], anno_groups[i], avg_groups[i]

    def _decode_pairwise_data(self, data, stage, prefixes):
        """transt_model.py:119-145: necessary transforms, and for burn-in pairs the joint transform."""
        data = self._preprocess(data, prefixes)
        if data is not None and stage == "burnin" and self.joint_trans is not None:
            self._joint(data, prefixes)
        return data

    def _decode_sp_data(self, sp_data):
        """transt_model.py:147-185: necessary transforms, joint transform, then the strong augmentation of the
        search ("aug") crops, whose boxes follow."""
        sp_data = self._preprocess(sp_data, ["relay", "aug"])
        if sp_data is not None:
            if self.joint_trans is not None:
                self._joint(sp_data, ["relay", "aug"])
            if self.independent_trans is not None:
                sp_data["aug_images"], sp_data["aug_anno"] = self.independent_trans(sp_data, tag="aug")
        return sp_data

    def forward(self, burnin_data=None, teacher_data=None, sp_data=None):
        burnin_data = self._decode_pairwise_data(burnin_data, "burnin", ["template", "search"])
        teacher_data = self._decode_pairwise_data(teacher_data, "sparse", ["key", "unlabeled"])
        sp_data = self._decode_sp_data(sp_data)
        templates, searches = [], []
        if teacher_data is not None:
            searches.append(teacher_data["unlabeled_images"])
            templates.append(teacher_data["key_images"])
        if burnin_data is not None:
            searches.append(burnin_data["search_images"])
            templates.append(burnin_data["template_images"])
        if sp_data is not None:
            searches.append(sp_data["aug_images"])
            templates.append(sp_data["relay_images"])
        template = (templates[0] if len(templates) == 1 else torch.cat(templates, dim=0)).detach()
        search = (searches[0] if len(searches) == 1 else torch.cat(searches, dim=0)).detach()

        (src_search, pos_search), (src_template, pos_template) = self._features_pair(search, template)
        hs = self.featurefusion_network.forward_tokens(src_template, src_search, pos_template, pos_search)

        if teacher_data is not None:
            out = self._heads(hs, use_iou=True, b_iou=self._get_batch_size(teacher_data))
        else:
            nb = self._get_batch_size(burnin_data)
            raw = self._heads(hs, use_iou=True, b_iou=nb)
            out = TensorDict()
            for key in (["logits", "boxes", "ious"] if self.iou_embed is not None else ["logits", "boxes"]):
                out[key + "_sup"] = raw["pred_" + key][:nb]
                if raw["pred_" + key].shape[0] > nb:
                    out[key + "_unsup"] = raw["pred_" + key][nb:]
        if burnin_data is not None:
            out["search_anno"] = burnin_data["search_anno"].detach()
        if sp_data is not None:
            for s in ("unlabeled", "aug"):
                if s + "_anno" in sp_data:
                    out[s + "_anno"] = sp_data[s + "_anno"].detach()
        return out


def transt_resnet50(settings, network="student"):
    """Build an SP-TransT model (transt_model.py:313-358); weights are randomly initialised."""
    assert network in ["student", "teacher", "baseline"], \
        "Network construction type should be in 'student', 'teacher' or 'baseline'."
    backbone_net = build_backbone(settings, backbone_pretrained=False)
    featurefusion_network = build_featurefusion_network(settings)
    aug = getattr(settings, "aug_settings", {})
    necessary_pre = NecessaryPreprocess(b_j=aug.get("jitter", 0.0), mean=settings.normalize_mean,
                                        std=settings.normalize_std)
    # transt_model.py:333-345: the teacher takes neither transform; the strong augmentation needs its settings tag.
    # (A grey-scale probability of 0 never fires; the transform is then left out instead of drawing for nothing.)
    joint_trans = independent_trans = None
    if network != "teacher" and aug.get("gray", 0.0) > 0:
        joint_trans = JointTransform([GrayScale(gray_probability=aug["gray"])])
    if network != "teacher" and getattr(settings, "strong_augmentation_tag", False):
        independent_trans = IndependentTransform([EllipseRotationAffine(
            max_angle=aug["angle"], lr_flip_prob=aug["lr_f_prob"], ud_flip_prob=aug["ud_f_prob"])])
    model = TransT(backbone_net, featurefusion_network, num_classes=settings.num_classes,
                   iou_prediction=settings.iou_prediction_tag, necessary_pre=necessary_pre,
                   independent_trans=independent_trans, joint_trans=joint_trans)
    return model.to(torch.device(settings.device))
