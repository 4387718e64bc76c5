"""CPU restatement of SPOT's tracker forward/backward hot path.  TEST INFRASTRUCTURE ONLY.

This file is the parity oracle.  It is imported by tests/, by
__graft_entry__.smoke() and by bench.py's cpu_baseline leg, and by nothing
else: the product path (spot_amd/) never imports it and has no CPU fallback.

Every function restates, in plain functional torch-CPU / numpy arithmetic on a
state dict, what one piece of the reference computes, and cites the reference
file:line it follows (paths relative to the reference checkout).  The oracle is
pinned: tests/test_oracle_golden.py checks the functions here against golden
vectors produced by running the reference itself (tests/golden/make_golden.py).
Two functions are PARITY UNPINNED, because what they restate cannot run here and
the reference holds no fixture for it: `deform_conv` (the reference's compiled
deformable-convolution extension) and `resize_linear_u8` (cv2.resize; cv2 is not
installed).  Both say so where they are defined.

Layout conventions: images NCHW; token tensors batch-first [B, L, C] (the
reference uses [L, B, C]; the permutation is applied at the boundaries only).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------
# pre-input (ltr/models/pre_input/necessary_preprocess.py:24-39,54-67,86-91)
# --------------------------------------------------------------------------
def preprocess(images_u8, avg, mean, std, brightness=None):
    """uint8 crop -> float, * brightness/255, clamp [0,1], (x-mean)/std; same on avg."""
    B = images_u8.shape[0]
    if brightness is None:
        brightness = torch.ones(B)
    f = (brightness / 255.0).to(torch.float32)
    m = torch.as_tensor(mean, dtype=torch.float32).view(1, -1, 1, 1)
    s = torch.as_tensor(std, dtype=torch.float32).view(1, -1, 1, 1)
    img = (images_u8.float() * f[:, None, None, None]).clamp(0.0, 1.0)
    img = (img - m) / s
    a = (avg.float() * f[:, None]).clamp(0.0, 1.0)
    a = (a - m.view(1, -1)) / s.view(1, -1)
    return img, a


# --------------------------------------------------------------------------
# sine position encoding (ltr/models/neck/position_encoding.py:27-45)
# --------------------------------------------------------------------------
def position_embedding_sine(mask, num_pos_feats=128, temperature=10000.0):
    """mask [B,H,W] bool (True = padded) -> pos [B, 2*num_pos_feats, H, W]."""
    not_mask = (~mask).to(torch.float32)
    y = not_mask.cumsum(1)
    x = not_mask.cumsum(2)
    eps, scale = 1e-6, 2 * math.pi
    y = y / (y[:, -1:, :] + eps) * scale
    x = x / (x[:, :, -1:] + eps) * scale
    i = torch.arange(num_pos_feats, dtype=torch.float32)
    dim_t = temperature ** (2 * torch.div(i, 2, rounding_mode="floor") / num_pos_feats)
    px = x[..., None] / dim_t
    py = y[..., None] / dim_t
    px = torch.stack((px[..., 0::2].sin(), px[..., 1::2].cos()), dim=4).flatten(3)
    py = torch.stack((py[..., 0::2].sin(), py[..., 1::2].cos()), dim=4).flatten(3)
    return torch.cat((py, px), dim=3).permute(0, 3, 1, 2).contiguous()


# --------------------------------------------------------------------------
# backbone: ResNet-50 to layer3 (ltr/models/backbone/resnet.py:56-92,95-112,174-203)
# --------------------------------------------------------------------------
def _bn(x, sd, p, training, stats=None):
    """nn.BatchNorm2d (resnet.py:66-90 uses the module default: momentum 0.1, eps 1e-5).  In training mode with a
    `stats` dict, the running statistics move as the module's buffers do (copies kept in `stats`, `sd` untouched)."""
    if training and stats is not None:
        for leaf in (".running_mean", ".running_var", ".num_batches_tracked"):
            if p + leaf not in stats:
                stats[p + leaf] = sd[p + leaf].clone()
        stats[p + ".num_batches_tracked"] += 1
        return F.batch_norm(x, stats[p + ".running_mean"], stats[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"],
                            training=True, momentum=0.1, eps=1e-5)
    return F.batch_norm(x, None if training else sd[p + ".running_mean"],
                        None if training else sd[p + ".running_var"],
                        sd[p + ".weight"], sd[p + ".bias"], training=training, eps=1e-5)


def _bottleneck(x, sd, p, stride, dilation, training, stats=None):
    out = F.relu(_bn(F.conv2d(x, sd[p + ".conv1.weight"]), sd, p + ".bn1", training, stats))
    out = F.conv2d(out, sd[p + ".conv2.weight"], stride=stride, padding=dilation, dilation=dilation)
    out = F.relu(_bn(out, sd, p + ".bn2", training, stats))
    out = _bn(F.conv2d(out, sd[p + ".conv3.weight"]), sd, p + ".bn3", training, stats)
    if (p + ".downsample.0.weight") in sd:
        x = _bn(F.conv2d(x, sd[p + ".downsample.0.weight"], stride=stride), sd,
                p + ".downsample.1", training, stats)
    return F.relu(out + x)


def resnet50_layer3(x, sd, prefix="backbone.0.body", training=False, stats=None):
    """conv1/bn/relu/maxpool, layer1 (3), layer2 (4, stride 2), layer3 (6, stride 1;
    first block dilation 2, the rest dilation 1 — resnet.py:112,152-167) -> [B,1024,H/8,W/8]."""
    p = prefix
    x = F.relu(_bn(F.conv2d(x, sd[p + ".conv1.weight"], stride=2, padding=3), sd, p + ".bn1", training, stats))
    x = F.max_pool2d(x, 3, 2, 1)
    for name, nblk, stride, dil in (("layer1", 3, 1, 1), ("layer2", 4, 2, 1), ("layer3", 6, 1, 2)):
        for b in range(nblk):
            first = b == 0
            x = _bottleneck(x, sd, "%s.%s.%d" % (p, name, b), stride if first else 1,
                            dil if first else 1, training, stats)
    return x


# --------------------------------------------------------------------------
# attention and fusion (ltr/models/neck/transt_fusion.py)
# --------------------------------------------------------------------------
def mha(xq, pq, xk, pk, sd, p, nhead, key_padding_mask=None):
    """nn.MultiheadAttention as called at transt_fusion.py:145-148,212-230.

    q = (xq+pq) Wq^T + bq, k = (xk+pk) Wk^T + bk, v = xk Wv^T + bv (no pos on v);
    q scaled by 1/sqrt(d_head) after the projection; key-padding mask -> -inf;
    softmax over keys; out-projection.  Tensors are [B, L, C].
    """
    W, b = sd[p + ".in_proj_weight"], sd[p + ".in_proj_bias"]
    C = W.shape[1]
    dh = C // nhead
    q = F.linear(xq + pq if pq is not None else xq, W[:C], b[:C])
    k = F.linear(xk + pk if pk is not None else xk, W[C:2 * C], b[C:2 * C])
    v = F.linear(xk, W[2 * C:], b[2 * C:])
    B, Lq, _ = q.shape
    Lk = k.shape[1]
    q = q.view(B, Lq, nhead, dh).transpose(1, 2) * (1.0 / math.sqrt(dh))
    k = k.view(B, Lk, nhead, dh).transpose(1, 2)
    v = v.view(B, Lk, nhead, dh).transpose(1, 2)
    s = q @ k.transpose(-1, -2)                                  # [B,h,Lq,Lk]
    if key_padding_mask is not None:
        s = s.masked_fill(key_padding_mask[:, None, None, :], float("-inf"))
    a = torch.softmax(s, dim=-1)
    o = (a @ v).transpose(1, 2).reshape(B, Lq, C)
    return F.linear(o, sd[p + ".out_proj.weight"], sd[p + ".out_proj.bias"])


def _ln(x, sd, p):
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], 1e-5)


def _ffn(x, sd, p1, p2):
    return F.linear(F.relu(F.linear(x, sd[p1 + ".weight"], sd[p1 + ".bias"])),
                    sd[p2 + ".weight"], sd[p2 + ".bias"])


def fusion_layer(s1, s2, p1, p2, m1, m2, sd, p, nhead):
    """FeatureFusionLayer.forward_post (transt_fusion.py:204-244), dropout off."""
    s1 = _ln(s1 + mha(s1, p1, s1, p1, sd, p + ".self_attn1", nhead, m1), sd, p + ".norm11")
    s2 = _ln(s2 + mha(s2, p2, s2, p2, sd, p + ".self_attn2", nhead, m2), sd, p + ".norm21")
    a12 = mha(s1, p1, s2, p2, sd, p + ".multihead_attn1", nhead, m2)
    a22 = mha(s2, p2, s1, p1, sd, p + ".multihead_attn2", nhead, m1)
    s1 = _ln(s1 + a12, sd, p + ".norm12")
    s1 = _ln(s1 + _ffn(s1, sd, p + ".linear11", p + ".linear12"), sd, p + ".norm13")
    s2 = _ln(s2 + a22, sd, p + ".norm22")
    s2 = _ln(s2 + _ffn(s2, sd, p + ".linear21", p + ".linear22"), sd, p + ".norm23")
    return s1, s2


def decoder_layer(tgt, mem, pos_dec, pos_enc, mem_mask, sd, p, nhead):
    """DecoderCFALayer.forward_post (transt_fusion.py:138-154)."""
    tgt = _ln(tgt + mha(tgt, pos_dec, mem, pos_enc, sd, p + ".multihead_attn", nhead, mem_mask),
              sd, p + ".norm1")
    return _ln(tgt + _ffn(tgt, sd, p + ".linear1", p + ".linear2"), sd, p + ".norm2")


def fusion_network(src_t, mask_t, src_s, mask_s, pos_t, pos_s, sd, prefix, nhead, nlayers):
    """FeatureFusionNetwork.forward (transt_fusion.py:38-57).

    Inputs NCHW ([B,C,H,W]) + masks [B,H,W]; output hs [1, B, Ls, C].
    """
    tok = lambda t: t.flatten(2).transpose(1, 2)          # [B, HW, C]
    s1, s2, p1, p2 = tok(src_t), tok(src_s), tok(pos_t), tok(pos_s)
    m1 = mask_t.flatten(1) if mask_t is not None else None
    m2 = mask_s.flatten(1) if mask_s is not None else None
    pre = prefix + "." if prefix else ""
    for i in range(nlayers):
        s1, s2 = fusion_layer(s1, s2, p1, p2, m1, m2, sd, "%sencoder.layers.%d" % (pre, i), nhead)
    hs = decoder_layer(s2, s1, p2, p1, m1, sd, pre + "decoder.layers.0", nhead)
    hs = _ln(hs, sd, pre + "decoder.norm")
    return hs.unsqueeze(0)


def mlp(x, sd, p, nlayers=3):
    """MLP (transt_model.py:296-310)."""
    for i in range(nlayers):
        x = F.linear(x, sd["%s.layers.%d.weight" % (p, i)], sd["%s.layers.%d.bias" % (p, i)])
        if i < nlayers - 1:
            x = F.relu(x)
    return x


def transt_features(img, sd, training=False, stats=None):
    """Joiner + input_proj (transt_backbone.py:62-70,89-98; transt_model.py:270-275).

    Full-size crops: the padding mask is all False (misc.py:283-300 with equal sizes).
    """
    f = resnet50_layer3(img, sd, "backbone.0.body", training, stats)
    B, _, H, W = f.shape
    mask = torch.zeros(B, H, W, dtype=torch.bool)
    pos = position_embedding_sine(mask, sd["input_proj.weight"].shape[0] // 2)
    src = F.conv2d(f, sd["input_proj.weight"], sd["input_proj.bias"])
    return src, mask, pos


def transt_heads(hs, sd, b_iou=None):
    """_fusion_pipeline heads (transt_model.py:73-84)."""
    out = {"pred_logits": mlp(hs, sd, "class_embed")[-1],
           "pred_boxes": mlp(hs, sd, "bbox_embed").sigmoid()[-1]}
    if b_iou is not None and "iou_embed.layers.0.weight" in sd:
        out["pred_ious"] = mlp(hs[:, :b_iou].detach(), sd, "iou_embed").sigmoid()[-1]
    return out


def transt_forward(template_u8, search_u8, sd, nhead=8, nlayers=4, b_iou=None, training=False,
                   mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225), stats=None):
    """TransT.forward for one (template, search) batch (transt_model.py:229-293), no augmentation.  `training`:
    batch-statistics BatchNorm; with a `stats` dict the BatchNorm buffers after the pass are left in it (the search
    pass runs first, then the template pass, as transt_model.py:270-271 orders them)."""
    z, _ = preprocess(template_u8, torch.zeros(template_u8.shape[0], 3), mean, std)
    x, _ = preprocess(search_u8, torch.zeros(search_u8.shape[0], 3), mean, std)
    src_s, mask_s, pos_s = transt_features(x, sd, training, stats)
    src_t, mask_t, pos_t = transt_features(z, sd, training, stats)
    hs = fusion_network(src_t, mask_t, src_s, mask_s, pos_t, pos_s, sd, "featurefusion_network",
                        nhead, nlayers)
    return transt_heads(hs, sd, b_iou if b_iou is not None else search_u8.shape[0])


# --------------------------------------------------------------------------
# pseudo-label decode (ltr/actor/transt_sparse_actor.py:57-98)
# --------------------------------------------------------------------------
def hanning_window(feat_sz):
    h = torch.tensor(np.hanning(feat_sz), dtype=torch.float32)
    return torch.outer(h, h).flatten()


def decode_teacher(pred_logits, pred_boxes, pred_ious, window_penalty, feat_sz, img_w, img_h):
    """softmax[:, :, 0]; Hanning blend; arg-max token; gather box + IoU conf; cxcywh -> xywh px; clamp."""
    score = torch.softmax(pred_logits, dim=2)[:, :, 0]
    win = hanning_window(feat_sz)
    s = window_penalty * win[None, :] + (1 - window_penalty) * score
    best_score, best_idx = torch.max(s, dim=1)
    ar = torch.arange(pred_boxes.shape[0])
    bbox = pred_boxes[ar, best_idx].clone()
    conf = pred_ious[ar, best_idx].reshape(-1)
    out = torch.empty_like(bbox)
    out[:, 0] = ((bbox[:, 0] - bbox[:, 2] / 2) * img_w).clamp_min(0)
    out[:, 1] = ((bbox[:, 1] - bbox[:, 3] / 2) * img_h).clamp_min(0)
    out[:, 2] = (bbox[:, 2] * img_w).clamp_min(1)
    out[:, 3] = (bbox[:, 3] * img_h).clamp_min(1)
    return {"best_bbox": out, "best_score": best_score, "best_conf": conf, "best_idx": best_idx,
            "score_map": s}


# --------------------------------------------------------------------------
# middleware box transforms + filter
# (ltr/data/processing_utils.py:180-229, ltr/data/middleware.py:97-152)
# --------------------------------------------------------------------------
def recover_bbox_to_original(box_crop, box_extract, resize_factor, crop_sz):
    rf = resize_factor.reshape(-1, 1)
    ext_c = box_extract[:, 0:2] + 0.5 * box_extract[:, 2:4]
    crop_c = box_crop[:, 0:2] + 0.5 * box_crop[:, 2:4]
    wh = box_crop[:, 2:4] / rf
    c = (crop_c - (crop_sz - 1) / 2) / rf + ext_c
    return torch.cat((c - 0.5 * wh, wh), dim=1).clamp(min=0.0)


def transform_image_to_crop(box_in, box_extract, resize_factor, crop_sz):
    rf = resize_factor.reshape(-1, 1)
    ext_c = box_extract[:, 0:2] + 0.5 * box_extract[:, 2:4]
    in_c = box_in[:, 0:2] + 0.5 * box_in[:, 2:4]
    out_c = (crop_sz - 1) / 2 + (in_c - ext_c) * rf
    wh = box_in[:, 2:4] * rf
    return torch.cat((out_c - 0.5 * wh, wh), dim=1)


def middleware_filter(best_bbox, best_conf, unl_box_extract, unl_rf, aug_box_extract, aug_rf,
                      search_sz, tau):
    """record_relay_results (middleware.py:49-95): recover, keep = conf >= tau, re-project, clamp >= 0."""
    rec = recover_bbox_to_original(best_bbox, unl_box_extract.reshape(-1, 4), unl_rf, search_sz)
    keep = best_conf >= tau
    aug = transform_image_to_crop(rec, aug_box_extract.reshape(-1, 4), aug_rf, search_sz).clamp_min(0.0)
    return {"aug_anno": aug, "aug_success": keep, "bbox_recovered": rec}


# --------------------------------------------------------------------------
# compaction / refill (ltr/trainers/ltr_sparse_trainer.py:334-357)
# --------------------------------------------------------------------------
def compact_indices(keep, batch):
    """Rows kept in order, then refilled by repeatedly appending the first
    min(missing, have) rows: a cyclic repetition of the kept indices."""
    kept = [i for i in range(len(keep)) if bool(keep[i])]
    if not kept:
        return None                        # reference skips the iteration (:328-332)
    idx = list(kept)
    while len(idx) < batch:
        n = min(batch - len(idx), len(idx))
        idx = idx + idx[:n]
    return np.asarray(idx, dtype=np.int64)


# --------------------------------------------------------------------------
# boxes (ltr/utils/box_ops.py:9-32,105-156)
# --------------------------------------------------------------------------
def box_cxcywh_to_xyxy(x):
    cx, cy, w, h = x.unbind(-1)
    return torch.stack([cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h], dim=-1)


def box_xyxy_to_xywh(x):
    x0, y0, x1, y1 = x.unbind(-1)
    return torch.stack([x0, y0, x1 - x0, y1 - y0], dim=-1)


def box_xyxy_to_cxcywh(x):
    x0, y0, x1, y1 = x.unbind(-1)
    return torch.stack([(x0 + x1) / 2, (y0 + y1) / 2, x1 - x0, y1 - y0], dim=-1)


def box_iou_pair_xyxy(b1, b2):
    a1 = (b1[:, 2] - b1[:, 0]) * (b1[:, 3] - b1[:, 1])
    a2 = (b2[:, 2] - b2[:, 0]) * (b2[:, 3] - b2[:, 1])
    wh = (torch.min(b1[:, 2:], b2[:, 2:]) - torch.max(b1[:, :2], b2[:, :2])).clamp(min=0)
    inter = wh[:, 0] * wh[:, 1]
    union = a1 + a2 - inter
    return inter / union, union


def generalized_box_iou_pair_xyxy(b1, b2):
    iou, union = box_iou_pair_xyxy(b1, b2)
    wh = (torch.max(b1[:, 2:], b2[:, 2:]) - torch.min(b1[:, :2], b2[:, :2])).clamp(min=0)
    area = wh[:, 0] * wh[:, 1]
    return iou - (area - union) / area, iou


# --------------------------------------------------------------------------
# matcher + criterion (ltr/models/loss/transt_matcher.py:24-109,
#                      ltr/models/tracking/transt_criterion.py:40-109,136-170)
# --------------------------------------------------------------------------
def matcher_masks(gt_cxcywh, num_queries, with_neg, gap=3):
    """Positive-token mask [B,L] and negative-token multiplicity [B,L] (int).

    The reference's negative ring is the union of four rectangles that do not
    overlap, so multiplicities are 0/1; kept as counts to stay literal.
    """
    B = gt_cxcywh.shape[0]
    n = int(np.sqrt(num_queries))
    pos = np.zeros((B, n, n), bool)
    neg = np.zeros((B, n, n), np.int32)
    g = np.asarray(gt_cxcywh, dtype=np.float32)
    for i in range(B):
        cx, cy, w, h = (float(v) for v in g[i])
        X0 = max(int(np.ceil((cx - w / 2) * n)), 0)
        Y0 = max(int(np.ceil((cy - h / 2) * n)), 0)
        X1 = min(int(np.ceil((cx + w / 2) * n)), n)
        Y1 = min(int(np.ceil((cy + h / 2) * n)), n)
        if X0 == X1:
            X1 += 1
        if Y0 == Y1:
            Y1 += 1
        pos[i, Y0:Y1, X0:X1] = True
        if with_neg:
            Yo0, Yo1, Xo0, Xo1 = max(0, Y0 - gap), min(n, Y1 + gap), max(0, X0 - gap), min(n, X1 + gap)
            for (a, b, c, d) in ((Yo0, Yo1, Xo0, X0), (Yo0, Y0, X0, X1), (Y1, Yo1, X0, X1),
                                 (Yo0, Yo1, X1, Xo1)):
                neg[i, a:b, c:d] += 1
    return pos.reshape(B, -1), neg.reshape(B, -1)


def criterion(logits, boxes, ious, gt_cxcywh, eos_coef=0.0625, tag="sup"):
    """SetCriterionTransT.forward for one tag; returns the reference's loss dict."""
    B, L, _ = logits.shape
    pos, neg = matcher_masks(gt_cxcywh.detach().numpy(), L, with_neg=(tag == "sup"))
    pos_t = torch.from_numpy(pos)
    target = torch.where(pos_t, torch.zeros((), dtype=torch.int64), torch.ones((), dtype=torch.int64))
    w = torch.tensor([1.0, eos_coef])
    losses = {"loss_ce": F.cross_entropy(logits.transpose(1, 2), target, w)}
    num_pos = max(float(pos.sum()), 1.0)
    bi, ti = torch.nonzero(pos_t, as_tuple=True)
    src = boxes[bi, ti]
    tgt = gt_cxcywh[bi]
    giou, iou = generalized_box_iou_pair_xyxy(box_cxcywh_to_xyxy(src), box_cxcywh_to_xyxy(tgt))
    losses["loss_bbox"] = (src - tgt).abs().sum() / num_pos
    losses["loss_giou"] = (1 - giou).sum() / num_pos
    losses["iou"] = iou.sum() / num_pos
    if tag == "sup" and ious is not None:
        num_neg = max(float(neg.sum()), 1.0)
        l_pos = (ious.squeeze(2)[bi, ti] - iou.detach()).abs().sum() / num_pos
        neg_t = torch.from_numpy(neg)
        nb, nt = torch.nonzero(neg_t > 0, as_tuple=True)
        mult = neg_t[nb, nt].float()
        iou_n, _ = box_iou_pair_xyxy(box_cxcywh_to_xyxy(boxes[nb, nt]), box_cxcywh_to_xyxy(gt_cxcywh[nb]))
        l_neg = ((ious.squeeze(2)[nb, nt] - iou_n.detach()).abs() * mult).sum() / num_neg
        losses["loss_iou_pred"] = l_pos + l_neg
    return losses


def build_target(anno_xywh_px, w, h):
    """_build_targets (transt_sparse_actor.py:203-224): xywh px -> normalised cxcywh."""
    a = anno_xywh_px
    return torch.stack([(a[:, 0] + a[:, 2] / 2) / w, (a[:, 1] + a[:, 3] / 2) / h, a[:, 2] / w, a[:, 3] / h], dim=1)


def burnin_loss(out, search_anno, weight_dict, eos_coef=0.0625, search_hw=(256, 256)):
    """_burnin_training (transt_sparse_actor.py:100-138): the supervised criterion (with the IoU-prediction term) on
    labelled pairs only; returns the weighted loss and the reference's loss dict."""
    h, w = search_hw
    ld = criterion(out["logits"], out["boxes"], out.get("ious"), build_target(search_anno, w, h), eos_coef, "sup")
    return sum(ld[k] * weight_dict[k + "_sup"] for k in ld if k + "_sup" in weight_dict), ld


def sp_training_loss(out_sup, out_unsup, search_anno, aug_anno, weight_dict, eos_coef=0.0625,
                     search_hw=(256, 256)):
    """_sp_training (transt_sparse_actor.py:140-201): two criterion calls, batch-weighted sum."""
    h, w = search_hw
    bs, bu = search_anno.shape[0], aug_anno.shape[0]
    ls = criterion(out_sup["logits"], out_sup["boxes"], out_sup.get("ious"), build_target(search_anno, w, h),
                   eos_coef, "sup")
    lu = criterion(out_unsup["logits"], out_unsup["boxes"], None, build_target(aug_anno, w, h),
                   eos_coef, "unsup")
    loss_sup = sum(ls[k] * weight_dict[k + "_sup"] for k in ls if k + "_sup" in weight_dict)
    loss_unsup = sum(lu[k] * weight_dict[k + "_unsup"] for k in lu if k + "_unsup" in weight_dict)
    total = (float(bs) * loss_sup + float(bu) * loss_unsup) / (bs + bu)
    return total, ls, lu


# --------------------------------------------------------------------------
# depthwise cross-correlation head (ltr/models/neck/ocean_fusion.py:211-247)
# --------------------------------------------------------------------------
def xcorr_depthwise(x, z):
    """out[b,c,i,j] = sum_{u,v} x[b,c,i+u,j+v] * z[b,c,u,v] (valid correlation), numpy."""
    x = np.asarray(x, dtype=np.float32)
    z = np.asarray(z, dtype=np.float32)
    B, C, Hx, Wx = x.shape
    _, _, Hz, Wz = z.shape
    Ho, Wo = Hx - Hz + 1, Wx - Wz + 1
    out = np.zeros((B, C, Ho, Wo), np.float32)
    for u in range(Hz):
        for v in range(Wz):
            out += x[:, :, u:u + Ho, v:v + Wo] * z[:, :, u:u + 1, v:v + 1]
    return out


def groupdw(xs, zs, weight):
    """GroupDW.forward: sum_i softmax(weight)_i * xcorr(x_i, z_i)."""
    w = np.asarray(weight, dtype=np.float32)
    e = np.exp(w - w.max())
    sw = (e / e.sum()).astype(np.float32)
    s = 0
    for i in range(3):
        s = s + sw[i] * xcorr_depthwise(xs[i], zs[i])
    return s.astype(np.float32)


# --------------------------------------------------------------------------
# teacher EMA (ltr/actor/transt_sparse_actor.py:11-33)
# --------------------------------------------------------------------------
def ema_update(teacher_sd, student_sd, keep_rate):
    """new = student*(1-keep) + teacher*keep for every entry; the result is cast
    back to the teacher entry's dtype by load_state_dict (int64 buffers truncate)."""
    out = {}
    for k, v in teacher_sd.items():
        new = student_sd[k] * (1 - keep_rate) + v * keep_rate
        out[k] = new.to(v.dtype)
    return out


# --------------------------------------------------------------------------
# online tracker post-processing (pysot_toolkit/trackers/transt_tracker.py:21-39,146-193)
# --------------------------------------------------------------------------
def tracker_post(pred_logits, pred_boxes, window, window_penalty, center_pos, size, image_hw):
    """One track() step after the network: returns (bbox xywh, pscore, new center, new size, s_x)."""
    w_x = size[0] + 3 * ((size[0] + size[1]) * 0.5)
    h_x = size[1] + 3 * ((size[0] + size[1]) * 0.5)
    s_x = math.ceil(math.sqrt(w_x * h_x))
    lg = np.asarray(pred_logits, dtype=np.float32).reshape(-1, 2)
    e = np.exp(lg - lg.max(axis=1, keepdims=True))
    score = (e[:, 0] / e.sum(axis=1)).astype(np.float32)
    delta = np.asarray(pred_boxes, dtype=np.float32).reshape(-1, 4).T
    pscore = score * (1 - window_penalty) + window * window_penalty
    best = int(np.argmax(pscore))
    bbox = delta[:, best] * s_x
    cx = bbox[0] + center_pos[0] - s_x / 2
    cy = bbox[1] + center_pos[1] - s_x / 2
    H, W = image_hw
    cx = max(0, min(cx, W))
    cy = max(0, min(cy, H))
    width = max(10, min(bbox[2], W))
    height = max(10, min(bbox[3], H))
    return ([cx - width / 2, cy - height / 2, width, height], pscore,
            np.array([cx, cy]), np.array([width, height]), s_x)


# --------------------------------------------------------------------------
# in-network augmentation (ltr/models/pre_input/transforms_in_net.py:121-398)
# --------------------------------------------------------------------------
def gray_scale(image_groups, avg_groups, gray_tag):
    """GrayScale.forward (:378-398) for given draws: tagged samples get their channels weighted by
    0.2989 / 0.5870 / 0.1140 (not summed) and their fill colour replaced by its weighted sum."""
    w = torch.tensor([0.2989, 0.5870, 0.1140]).float()
    imgs = [g.clone() for g in image_groups]
    avgs = [a.clone().float() for a in avg_groups]
    for i in range(imgs[0].shape[0]):
        if gray_tag[i]:
            for g in range(len(imgs)):
                imgs[g][i].mul_(w[:, None, None])
                c = avgs[g][i][0] * w[0] + avgs[g][i][1] * w[1] + avgs[g][i][2] * w[2]
                avgs[g][i] = torch.stack([c, c, c])
    return imgs, avgs


def _inverse_rotation_matrix(angle_deg):
    """_get_inverse_affine_matrix (:247-275) for center (0.5, 0.5), no translation, scale 1, no shear."""
    rot = math.radians(angle_deg)
    a, b, c, d = math.cos(rot), -math.sin(rot), math.sin(rot), math.cos(rot)
    m = [d, -b, 0.0, -c, a, 0.0]
    m[2] += m[0] * -0.5 + m[1] * -0.5
    m[5] += m[3] * -0.5 + m[4] * -0.5
    m[2] += 0.5
    m[5] += 0.5
    return m


def _rotate_nearest(img, angle_deg, fill):
    """affine() + _apply_grid_transform() (:277-331): centred base grid, theta rescaled to [-1, 1], nearest
    grid_sample with zero padding, uncovered pixels (a ones channel sampled below 0.5) take `fill`."""
    Cn, H, W = img.shape
    theta = torch.tensor(_inverse_rotation_matrix(-angle_deg), dtype=torch.float32).reshape(1, 2, 3)
    base = torch.empty(1, H, W, 3, dtype=torch.float32)
    base[..., 0].copy_(torch.linspace(-W * 0.5 + 0.5, W * 0.5 + 0.5 - 1, steps=W))
    base[..., 1].copy_(torch.linspace(-H * 0.5 + 0.5, H * 0.5 + 0.5 - 1, steps=H).unsqueeze_(-1))
    base[..., 2].fill_(1)
    rescaled = theta.transpose(1, 2) / torch.tensor([0.5 * W, 0.5 * H], dtype=torch.float32)
    grid = base.view(1, H * W, 3).bmm(rescaled).view(1, H, W, 2)
    x = torch.cat((img[None], torch.ones((1, 1, H, W), dtype=img.dtype)), dim=1)
    x = F.grid_sample(x, grid, mode="nearest", padding_mode="zeros", align_corners=False)
    covered, x = x[:, -1:], x[:, :-1]
    fill_img = fill.to(img.dtype).view(1, Cn, 1, 1).expand_as(x)
    return torch.where(covered.expand_as(x) < 0.5, fill_img, x)[0]


def ellipse_rotation_affine(image, bbox, avg, mask, angle, lr_flip_tag):
    """EllipseRotationAffine.forward (:150-216) for given draws (angle [B] float degrees incl. the 180 of an
    up/down flip, lr_flip_tag [B] bool).  Returns (images, xywh boxes)."""
    B, Cn, H, W = image.shape
    out = image.clone()
    for i in range(B):
        if mask[i]:
            src = out[i].flip((2,)) if lr_flip_tag[i] else out[i]
            out[i] = _rotate_nearest(src, int(angle[i]), avg[i])
    rwrh = 1 / 2 * bbox[:, 2:]
    cxcy = bbox[:, :2] + rwrh
    cxcy[:, 0] = torch.where(lr_flip_tag, (W - 1) - cxcy[:, 0], cxcy[:, 0])
    a = torch.arange(0, 360, 1) * (math.pi / 180)
    xs = rwrh[:, 0][:, None] * torch.sin(a).expand([B, 360]) + cxcy[:, 0][:, None]
    ys = rwrh[:, 1][:, None] * torch.cos(a).expand([B, 360]) + cxcy[:, 1][:, None]
    rad = angle * (math.pi / 180)
    alpha, beta = torch.cos(rad), torch.sin(rad)
    x0, y0 = W * 0.5, H * 0.5
    M = torch.zeros(B, 2, 3)
    M[:, 0, 0], M[:, 0, 1], M[:, 0, 2] = alpha, beta, (1. - alpha) * x0 - beta * y0
    M[:, 1, 0], M[:, 1, 1], M[:, 1, 2] = -beta, alpha, beta * x0 + (1. - alpha) * y0
    pts = torch.cat([xs.unsqueeze(1), ys.unsqueeze(1), torch.ones(xs.shape).unsqueeze(1)], dim=1).float()
    tr = torch.bmm(M, pts)
    mx, mn = tr.max(dim=2).values, tr.min(dim=2).values
    new = torch.stack([mn[:, 0], mn[:, 1], mx[:, 0] - mn[:, 0], mx[:, 1] - mn[:, 1]], dim=1)
    return out, torch.clamp_min(torch.where(mask[:, None], new, bbox), min=0.0)


# --------------------------------------------------------------------------
# Siamese (SP-Ocean) head around the correlation
# (ltr/models/neck/ocean_fusion.py:14-114,117-171; ltr/models/tracking/ocean_model.py:79-191;
#  ltr/models/loss/ocean_label_gen.py:44-233; ltr/models/tracking/ocean_criterion.py:42-258)
# --------------------------------------------------------------------------
def _seq_bn(x, sd, p, training):
    """BatchNorm stored as entry p of an nn.Sequential; in training mode the running buffers in sd are updated."""
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"],
                        training=training, momentum=0.1, eps=1e-5)


def _conv_bn_relu(x, sd, p, training, padding=0, dilation=1, first=0):
    x = F.conv2d(x, sd["%s.%d.weight" % (p, first)], sd.get("%s.%d.bias" % (p, first)), padding=padding, dilation=dilation)
    return F.relu(_seq_bn(x, sd, "%s.%d" % (p, first + 1), training))


def ocean_adjust(x, sd, p, training, crop=False):
    """ocean_fusion.py:22-32: 1x1 conv + bn; crop=True also returns the central part [4:-4, 4:-4]."""
    y = _seq_bn(F.conv2d(x, sd[p + ".downsample.0.weight"]), sd, p + ".downsample.1", training)
    return (y, y[:, :, 4:-4, 4:-4]) if crop else y


def ocean_matrix(z, x, sd, p, training):
    """ocean_fusion.py:161-171: unpadded 3x3 conv-bn-relu, dilation 1, (2, 1), (1, 2), for template and search."""
    zs = [_conv_bn_relu(z, sd, "%s.%s_k" % (p, n), training, dilation=d)
          for n, d in (("matrix11", 1), ("matrix12", (2, 1)), ("matrix21", (1, 2)))]
    xs = [_conv_bn_relu(x, sd, "%s.%s_s" % (p, n), training, dilation=d)
          for n, d in (("matrix11", 1), ("matrix12", (2, 1)), ("matrix21", (1, 2)))]
    return zs, xs


def _torch_groupdw(zs, xs, w):
    def xc(x, k):
        b, c = k.shape[:2]
        o = F.conv2d(x.reshape(1, b * c, *x.shape[2:]), k.reshape(b * c, 1, *k.shape[2:]), groups=b * c)
        return o.reshape(b, c, *o.shape[2:])
    sw = F.softmax(w, 0)
    return sum(sw[i] * xc(xs[i], zs[i]) for i in range(3))


def ocean_box_tower(search, kernel, sd, training, towernum=1, batch_iou=0, p=""):
    """ocean_fusion.py:87-114.  Returns (ltrb, cls, cls_dw, x_reg, pred_ious or None)."""
    def tower(x, name):
        for i in range(towernum):
            x = _conv_bn_relu(x, sd, p + name, training, padding=1, first=3 * i)
        return x
    cls_z, cls_x = ocean_matrix(kernel, search, sd, p + "cls_encode", training)
    reg_z, reg_x = ocean_matrix(kernel, search, sd, p + "reg_encode", training)
    cls_dw = _torch_groupdw(cls_z, cls_x, sd[p + "cls_dw.weight"])
    reg_dw = _torch_groupdw(reg_z, reg_x, sd[p + "reg_dw.weight"])
    x_reg = tower(reg_dw, "bbox_tower")
    x = sd[p + "adjust"] * F.conv2d(x_reg, sd[p + "bbox_pred.weight"], sd[p + "bbox_pred.bias"], padding=1) + sd[p + "bias"]
    x = torch.exp(torch.clamp_max(x, 4.7))
    c = tower(cls_dw, "cls_tower")
    cls = 0.1 * F.conv2d(c, sd[p + "cls_pred.weight"], sd[p + "cls_pred.bias"], padding=1)
    ious = None
    if batch_iou > 0:
        t = tower(reg_dw.detach()[:batch_iou], "iou_pred_tower")
        ious = torch.sigmoid(F.conv2d(t, sd[p + "iou_pred.weight"], sd[p + "iou_pred.bias"], padding=1))
    return x, cls, cls_dw, x_reg, ious


def ocean_grid(score_size, search_size, stride=8):
    """ocean_model.py:79-96: float64 cell centres, x along columns, y along rows."""
    idx = np.arange(0, score_size) - np.floor(float(score_size // 2))
    gx, gy = np.meshgrid(idx, idx)
    return torch.tensor(gx * stride + search_size // 2), torch.tensor(gy * stride + search_size // 2)


def ocean_pred_to_image(bbox_pred, search_size, stride=8):
    """ocean_model.py:104-121 (its grid is float32 there)."""
    gx, gy = ocean_grid(bbox_pred.shape[-1], search_size, stride)
    gx, gy = gx.float(), gy.float()
    return torch.stack([gx - bbox_pred[:, 0], gy - bbox_pred[:, 1], gx + bbox_pred[:, 2], gy + bbox_pred[:, 3]], 1)


def ocean_align_offsets(boxes, feat_hw, search_size, stride=8):
    """ocean_model.py:123-191 for one clip after the other; boxes [B, H*W, 4] -> [B, H*W, 18] as (y, x) pairs."""
    fh, fw = feat_hw
    outs = []
    for b in boxes:
        idx = torch.arange(-1, 2, dtype=b.dtype)
        yy, xx = torch.meshgrid(idx, idx, indexing="ij")
        xx, yy = xx.reshape(-1), yy.reshape(-1)
        w = (b[:, 2] - b[:, 0] + 1) / stride / 2 - 1
        h = (b[:, 3] - b[:, 1] + 1) / stride / 2 - 1
        x = ((b[:, 0] + b[:, 2]) * 0.5 - search_size * 0.5) / stride + fw // 2
        y = ((b[:, 1] + b[:, 3]) * 0.5 - search_size * 0.5) / stride + fh // 2
        gy, gx = torch.meshgrid(torch.arange(0, fh), torch.arange(0, fw), indexing="ij")
        ox = w[:, None] * xx + (x - gx.reshape(-1).type_as(x))[:, None]
        oy = h[:, None] * yy + (y - gy.reshape(-1).type_as(y))[:, None]
        outs.append(torch.stack([oy, ox], -1).reshape(b.shape[0], -1))
    return torch.stack(outs)


def ocean_labels(anno_xywh, S, stride, search_sz, expand=0.2, radius=2):
    """ocean_label_gen.py:44-66,124-233 -> cls [B,S,S] f32, reg [B,S,S,4] f64, reg_w, iou_w [B,S,S] f32."""
    gx, gy = ocean_grid(S, search_sz, stride)                       # float64
    x1, y1 = anno_xywh[:, 0], anno_xywh[:, 1]
    x2, y2 = x1 + anno_xywh[:, 2], y1 + anno_xywh[:, 3]

    def ltrb(a1, b1, a2, b2):
        return torch.stack([gx[None] - a1[:, None, None], gy[None] - b1[:, None, None],
                            a2[:, None, None] - gx[None], b2[:, None, None] - gy[None]], -1)
    reg = ltrb(x1, y1, x2, y2)
    reg_w = (reg.min(-1).values > 0).float()
    w, h = torch.clamp_min(x2 - x1, 1), torch.clamp_min(y2 - y1, 1)
    iou_w = (ltrb(x1 - expand * w, y1 - expand * h, x2 + expand * w, y2 + expand * h).min(-1).values > 0).float()
    cx = S // 2 + (((x1 + x2) / 2 - search_sz / 2) / stride).long()
    cy = S // 2 + (((y1 + y2) / 2 - search_sz / 2) / stride).long()
    r = torch.arange(S)
    cls = ((r[None, None, :] - cx[:, None, None]).abs() + (r[None, :, None] - cy[:, None, None]).abs() <= radius).float()
    return cls, reg, reg_w, iou_w


def ocean_ltrb_iou(pred, target):
    """IoU of two boxes given as distances from a common point (ocean_label_gen.py:78-98, ocean_criterion.py:81-100)."""
    area_t = (target[..., 0] + target[..., 2]) * (target[..., 1] + target[..., 3])
    area_p = (pred[..., 0] + pred[..., 2]) * (pred[..., 1] + pred[..., 3])
    wi = torch.min(pred[..., 0], target[..., 0]) + torch.min(pred[..., 2], target[..., 2])
    hi = torch.min(pred[..., 3], target[..., 3]) + torch.min(pred[..., 1], target[..., 1])
    inter = wi * hi
    return (inter + 1.0) / (area_t + area_p - inter + 1.0)


def ocean_align_labels(bbox_pred, reg_label, reg_weight):
    """ocean_label_gen.py:68-104."""
    return torch.clamp(torch.abs(reg_weight * ocean_ltrb_iou(bbox_pred.permute(0, 2, 3, 1), reg_label)), 0, 1)


def ocean_losses(logits, cls_label, align_logits, align_label, bbox_pred, reg_label, reg_weight, pred_ious, iou_weight):
    """ocean_criterion.py:42-69,117-148,194-258 by index selection, as the reference does it."""
    bce = torch.nn.BCEWithLogitsLoss()

    def half(value):
        sel = cls_label.reshape(-1).eq(value).nonzero().squeeze()
        if sel.dim() == 0:
            return 0
        return bce(logits.reshape(-1)[sel], cls_label.reshape(-1)[sel])
    loss_ce = half(1) * 0.5 + half(0) * 0.5
    loss_align = bce(align_logits, align_label.unsqueeze(1))
    pos = torch.nonzero(reg_weight.reshape(-1) > 0).squeeze(1)
    ious = ocean_ltrb_iou(bbox_pred.permute(0, 2, 3, 1).reshape(-1, 4)[pos], reg_label.reshape(-1, 4)[pos])
    loss_bbox, iou = (-torch.log(ious)).mean(), ious.mean()
    err = (align_label.detach() - pred_ious.squeeze(1)).abs()
    loss_iou = ((err * (iou_weight - reg_weight)).sum() + (err * reg_weight).sum()) / iou_weight.sum()
    return {"loss_ce": loss_ce, "loss_align": loss_align, "loss_bbox": loss_bbox, "iou": iou, "loss_iou_pred": loss_iou}


# --------------------------------------------------------------------------
# deformable convolution v1 (ltr/models/neck/dcn/src/deform_conv_cuda_kernel.cu:82-113,189-241; gradients :115-186,
# 277-333,371-434 are what autograd derives from this forward wherever the sample point is not on a pixel row/column)
# PARITY UNPINNED: the reference holds no fixture for this operator and its extension cannot be built or run here;
# this restates the published DCNv1 algorithm as the reference's kernel source implements it.
# --------------------------------------------------------------------------
def deform_conv(x, offset, weight, stride=1, pad=0, dil=1, dgroups=1):
    """x [B,C,H,W], offset [B, dgroups*2*kh*kw, Ho, Wo] as (dy, dx) per tap, weight [Co,C,kh,kw] -> [B,Co,Ho,Wo]."""
    B, C, H, W = x.shape
    Co, _, kh, kw = weight.shape
    Ho = (H + 2 * pad - dil * (kh - 1) - 1) // stride + 1
    Wo = (W + 2 * pad - dil * (kw - 1) - 1) // stride + 1
    oy = (torch.arange(Ho, dtype=x.dtype) * stride - pad).view(1, Ho, 1)
    ox = (torch.arange(Wo, dtype=x.dtype) * stride - pad).view(1, 1, Wo)
    cpg = C // dgroups
    flat = x.reshape(B, C, H * W)
    groups = []
    for dg in range(dgroups):
        xs = flat[:, dg * cpg:(dg + 1) * cpg]
        taps = []
        for tap in range(kh * kw):
            ky, kx = divmod(tap, kw)
            h = oy + ky * dil + offset[:, dg * 2 * kh * kw + 2 * tap]
            w = ox + kx * dil + offset[:, dg * 2 * kh * kw + 2 * tap + 1]
            valid = (h > -1) & (w > -1) & (h < H) & (w < W)
            hl, wl = torch.floor(h.detach()), torch.floor(w.detach())
            lh, lw = h - hl, w - wl

            def at(hi, wi):
                inb = (valid & (hi >= 0) & (hi <= H - 1) & (wi >= 0) & (wi <= W - 1)).view(B, 1, -1)
                idx = (hi.clamp(0, H - 1) * W + wi.clamp(0, W - 1)).long().view(B, 1, -1).expand(B, cpg, -1)
                return xs.gather(2, idx) * inb
            lh_, lw_ = lh.reshape(B, 1, -1), lw.reshape(B, 1, -1)
            taps.append((1 - lh_) * (1 - lw_) * at(hl, wl) + (1 - lh_) * lw_ * at(hl, wl + 1)
                        + lh_ * (1 - lw_) * at(hl + 1, wl) + lh_ * lw_ * at(hl + 1, wl + 1))
        groups.append(torch.stack(taps, 2))                      # [B, cpg, taps, N]
    cols = torch.cat(groups, 1).reshape(B, C * kh * kw, Ho * Wo)
    return (weight.reshape(Co, -1) @ cols).reshape(B, Co, Ho, Wo)


# --------------------------------------------------------------------------
# crop + pad + resize (pysot_toolkit/trackers/transt_tracker.py:41-101; ltr/data/processing_utils.py:8-91)
# The window / padding part restates the reference line by line.  The resize part restates OpenCV's 8-bit
# INTER_LINEAR (and its 2:1 box-average substitution): PARITY UNPINNED for it, cv2 is not installed here and the
# reference holds no resized fixture.
# --------------------------------------------------------------------------
def crop_window(im, pos, original_sz, avg_chans):
    """The reference's padded window before any resize: uint8 [sz, sz, 3], plus its (x0, y0) in frame coordinates."""
    sz = original_sz
    r, c_, k = im.shape
    c = (original_sz + 1) / 2
    xmin = np.floor(pos[0] - c + 0.5)
    ymin = np.floor(pos[1] - c + 0.5)
    xmax, ymax = xmin + sz - 1, ymin + sz - 1
    left, top = int(max(0., -xmin)), int(max(0., -ymin))
    right, bottom = int(max(0., xmax - c_ + 1)), int(max(0., ymax - r + 1))
    te = np.zeros((r + top + bottom, c_ + left + right, k), np.uint8)
    te[top:top + r, left:left + c_, :] = im
    if top:
        te[0:top, left:left + c_, :] = avg_chans
    if bottom:
        te[r + top:, left:left + c_, :] = avg_chans
    if left:
        te[:, 0:left, :] = avg_chans
    if right:
        te[:, c_ + left:, :] = avg_chans
    patch = te[int(ymin + top):int(ymax + top + 1), int(xmin + left):int(xmax + left + 1), :]
    return patch, int(xmin), int(ymin)


def resize_linear_u8(patch, S):
    """uint8 [n, n, 3] -> [S, S, 3] by OpenCV's fixed-point bilinear rule (see the section header)."""
    n = patch.shape[0]
    if n == S:
        return patch.copy()
    p = patch.astype(np.int64)
    if n == 2 * S:
        return ((p[0::2, 0::2] + p[0::2, 1::2] + p[1::2, 0::2] + p[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    scale = n / S
    d = np.arange(S, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = f - s.astype(np.float32)
    fx, sx = f.copy(), s.copy()
    fx[sx < 0], sx[sx < 0] = 0, 0
    fx[sx >= n - 1], sx[sx >= n - 1] = 0, n - 1
    ax0 = np.rint((np.float32(1) - fx) * np.float32(2048)).astype(np.int64)
    ax1 = np.rint(fx * np.float32(2048)).astype(np.int64)
    sx1 = np.minimum(sx + 1, n - 1)
    ay0 = np.rint((np.float32(1) - f) * np.float32(2048)).astype(np.int64)
    ay1 = np.rint(f * np.float32(2048)).astype(np.int64)
    sy0, sy1 = np.clip(s, 0, n - 1), np.clip(s + 1, 0, n - 1)
    rows = p[:, sx] * ax0[None, :, None] + p[:, sx1] * ax1[None, :, None]             # [n, S, 3]
    out = (((ay0[:, None, None] * (rows[sy0] >> 4)) >> 16) + ((ay1[:, None, None] * (rows[sy1] >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def ocean_decode_teacher(pred_logits, align_logits, pred_ious, boxes_xyxy, cls_ratio, window_penalty):
    """ltr/actor/ocean_sparse_actor.py:40-94: score mix, Hanning penalty, arg-max cell -> (xywh box, score, IoU)."""
    B, S = pred_logits.shape[0], pred_logits.shape[-1]
    score = cls_ratio * torch.sigmoid(pred_logits) + (1 - cls_ratio) * torch.sigmoid(align_logits)
    h = torch.tensor(np.hanning(S), dtype=score.dtype)
    score = (window_penalty * torch.outer(h, h)[None, None] + (1 - window_penalty) * score).reshape(B, -1)
    best, idx = score.max(1)
    b = boxes_xyxy.reshape(B, 4, -1).transpose(1, 2)[torch.arange(B), idx]
    xywh = torch.stack([b[:, 0].clamp_min(0), b[:, 1].clamp_min(0), (b[:, 2] - b[:, 0]).clamp_min(1),
                        (b[:, 3] - b[:, 1]).clamp_min(1)], 1)
    return xywh, best, pred_ious.reshape(B, -1)[torch.arange(B), idx]


def ocean_tracker_update(cls, ltrb, align, target_pos, target_sz, window, scale_z, ratio=0.94, penalty_k=0.062,
                         window_influence=0.38, lr=0.765, stride=8, instance_size=255):
    """pysot_toolkit/trackers/ocean_tracker.py:71-130 on numpy maps cls / align [S,S] float32, ltrb [4,S,S] float32;
    target_sz is already scaled by scale_z, as the caller passes it (:155-156)."""
    S = cls.shape[-1]
    gx, gy = ocean_grid(S, instance_size, stride)
    gx, gy = gx.numpy(), gy.numpy()
    sig = lambda a: (1.0 / (1.0 + np.exp(-a.astype(np.float32)))).astype(np.float32)
    score = np.float32(ratio) * sig(cls) + np.float32(1 - ratio) * sig(align)
    x1, y1, x2, y2 = gx - ltrb[0], gy - ltrb[1], gx + ltrb[2], gy + ltrb[3]
    change = lambda r: np.maximum(r, 1.0 / r)

    def sz(w, h):
        pad = (w + h) * 0.5
        return np.sqrt((w + pad) * (h + pad))
    s_c = change(sz(x2 - x1, y2 - y1) / sz(target_sz[0], target_sz[1]))
    r_c = change((target_sz[0] / target_sz[1]) / ((x2 - x1) / (y2 - y1)))
    penalty = np.exp(-(r_c * s_c - 1) * penalty_k)
    pscore = penalty * score * (1 - window_influence) + window * window_influence
    r, c = np.unravel_index(pscore.argmax(), pscore.shape)
    w, h = (x2 - x1)[r, c] / scale_z, (y2 - y1)[r, c] / scale_z
    dx = ((x1 + x2)[r, c] / 2 - instance_size // 2) / scale_z
    dy = ((y1 + y2)[r, c] / 2 - instance_size // 2) / scale_z
    tsz = np.asarray(target_sz, dtype=np.float64) / scale_z
    rate = penalty[r, c] * score[r, c] * lr
    res = np.array([w * rate + (1 - rate) * tsz[0], h * rate + (1 - rate) * tsz[1]])
    return np.array([target_pos[0] + dx, target_pos[1] + dy]), tsz * (1 - rate) + rate * res, score[r, c]
