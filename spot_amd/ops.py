"""Torch-facing wrappers of the C-ABI kernels (PyTorch supplies device memory and
streams; every arithmetic step below runs in libspot_hip.so).

Functional wrappers take/return torch CUDA tensors; the autograd Functions glue
forward and backward kernels together so that the host modules (fusion.py,
transt.py) read like the reference's nn.Modules.
"""
import ctypes as C
import math

import torch

from . import lib
from .lib import GemmDesc


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_current_device = getattr(torch._C, "_cuda_getDevice", None)


def _st():
    """The current HIP stream of the current device as a C pointer.  Asked for once per kernel launch, so it takes the
    framework's raw-handle query (a fraction of a microsecond) where that exists instead of building a Stream object."""
    if _raw_stream is not None and _current_device is not None:
        return C.c_void_p(_raw_stream(_current_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class KernelTimer:
    """Optional per-launch timing with HIP events on the launch stream (used by bench.py's roofline leg)."""

    def __init__(self, shapes=False, only=None):
        self.records = []          # (key, work, start_event, end_event)
        self.shapes = shapes       # key GEMM launches by problem shape as well
        self.only = set(only) if only else None     # kernel classes to time (None: all); the rest launch untimed

    def wants(self, cls):
        return self.only is None or cls in self.only

    def run(self, key, work, launch):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        launch()
        e.record()
        self.records.append((key, work, s, e))

    def summary(self):
        torch.cuda.synchronize()
        agg = {}
        for key, work, s, e in self.records:
            a = agg.setdefault(key, {"launches": 0, "ms": 0.0, "work": 0.0})
            a["launches"] += 1
            a["ms"] += s.elapsed_time(e)
            a["work"] += work
        return agg


TIMER = None   # set to a KernelTimer to time gemm / attention launches


class _Seeds:
    """Seeds for the counter-based dropout masks: a base (set by manual_seed, mix the rank in for
    data-parallel runs) and a per-call counter, so a run is reproducible without device RNG state."""
    base = 0x5107
    counter = 0


def manual_seed(seed):
    _Seeds.base = int(seed) & 0xffffffff
    _Seeds.counter = 0


def next_seed():
    _Seeds.counter += 1
    return (_Seeds.base * 0x9E3779B1 + _Seeds.counter * 0x85EBCA6B) & 0xffffffff


def _p(t):
    if t is None:
        return None
    if lib._keeping is not None:
        lib._keeping.append(t)
    return C.c_void_p(t.data_ptr())


def _chk(t, dtype=torch.float32, name="tensor"):
    if not t.is_cuda:
        raise RuntimeError("spot_amd: %s must live on the GPU (no CPU path exists)" % name)
    if t.dtype != dtype:
        raise RuntimeError("spot_amd: %s must be %s, got %s" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise RuntimeError("spot_amd: %s must be contiguous" % name)
    return t


# ---------------------------------------------------------------------------
# pre-input / pseudo-label path
# ---------------------------------------------------------------------------
def preprocess_u8(images, scale, mean, std):
    """(clamp(u8 * scale[b], 0, 1) - mean) / std  ->  float32 [B,3,H,W]."""
    _chk(images, torch.uint8, "images")
    B, Cn, H, W = images.shape
    out = torch.empty((B, Cn, H, W), dtype=torch.float32, device=images.device)
    m = (C.c_float * 3)(*[float(v) for v in mean])
    s = (C.c_float * 3)(*[float(v) for v in std])
    lib.call("spot_preprocess_u8_f32", _p(images), _p(scale), m, s, _p(out), B, Cn, H, W, _st())
    return out


_WINDOWS = {}


def hanning_window(feat_sz, device):
    """outer(hanning, hanning) formed in float32, as transt_sparse_actor.py:65-68 does."""
    key = (feat_sz, str(device))
    if key not in _WINDOWS:
        import numpy as np
        h = torch.tensor(np.hanning(feat_sz), dtype=torch.float32)
        _WINDOWS[key] = torch.outer(h, h).flatten().contiguous().to(device)
    return _WINDOWS[key]


def decode_teacher(pred_logits, pred_boxes, pred_ious, window_penalty, feat_sz, img_w, img_h):
    B, L, _ = pred_logits.shape
    dev = pred_logits.device
    _chk(pred_logits), _chk(pred_boxes), _chk(pred_ious)
    bbox = torch.empty((B, 4), dtype=torch.float32, device=dev)
    score = torch.empty((B,), dtype=torch.float32, device=dev)
    conf = torch.empty((B,), dtype=torch.float32, device=dev)
    idx = torch.empty((B,), dtype=torch.int32, device=dev)
    lib.call("spot_decode_teacher_f32", _p(pred_logits), _p(pred_boxes), _p(pred_ious),
             _p(hanning_window(feat_sz, dev)), float(window_penalty), int(img_w), int(img_h),
             _p(bbox), _p(score), _p(conf), _p(idx), B, L, _st())
    return bbox, score, conf, idx


def middleware_filter(best_bbox, best_conf, unl_ext, unl_rf, aug_ext, aug_rf, search_sz, tau):
    B = best_bbox.shape[0]
    dev = best_bbox.device
    for t in (best_bbox, best_conf, unl_ext, unl_rf, aug_ext, aug_rf):
        _chk(t)
    rec = torch.empty((B, 4), dtype=torch.float32, device=dev)
    aug = torch.empty((B, 4), dtype=torch.float32, device=dev)
    keep = torch.empty((B,), dtype=torch.uint8, device=dev)
    lib.call("spot_middleware_filter_f32", _p(best_bbox), _p(best_conf), _p(unl_ext), _p(unl_rf), _p(aug_ext),
             _p(aug_rf), int(search_sz), float(tau), _p(rec), _p(aug), _p(keep), B, _st())
    return rec, aug, keep


def decode_filter(pred_logits, pred_boxes, pred_ious, window_penalty, feat_sz, img_w, img_h,
                  unl_ext, unl_rf, aug_ext, aug_rf, search_sz, tau):
    """Fused decode + recover + keep mask + re-projection: one launch, no host sync."""
    B, L, _ = pred_logits.shape
    dev = pred_logits.device
    for t in (pred_logits, pred_boxes, pred_ious, unl_ext, unl_rf, aug_ext, aug_rf):
        _chk(t)
    f = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
    bbox, score, conf, rec, aug = f(B, 4), f(B), f(B), f(B, 4), f(B, 4)
    idx = torch.empty((B,), dtype=torch.int32, device=dev)
    keep = torch.empty((B,), dtype=torch.uint8, device=dev)
    lib.call("spot_decode_filter_f32", _p(pred_logits), _p(pred_boxes), _p(pred_ious),
             _p(hanning_window(feat_sz, dev)), float(window_penalty), int(img_w), int(img_h), _p(unl_ext),
             _p(unl_rf), _p(aug_ext), _p(aug_rf), int(search_sz), float(tau), _p(bbox), _p(score), _p(conf),
             _p(idx), _p(rec), _p(aug), _p(keep), B, L, _st())
    return {"best_bbox": bbox, "best_score": score, "best_conf": conf, "best_idx": idx,
            "bbox_recovered": rec, "aug_anno": aug, "aug_success": keep}


def compact_indices(keep_u8, n_out):
    _chk(keep_u8, torch.uint8, "keep")
    dev = keep_u8.device
    idx = torch.empty((n_out,), dtype=torch.int32, device=dev)
    n = torch.empty((1,), dtype=torch.int32, device=dev)
    lib.call("spot_compact_indices", _p(keep_u8), keep_u8.numel(), _p(idx), int(n_out), _p(n), _st())
    return idx, n


def gather_rows(src, idx):
    """dst[r] = src[idx[r]] along dim 0 (any dtype)."""
    if not src.is_contiguous():
        raise RuntimeError("spot_amd: gather_rows source must be contiguous")
    _chk(idx, torch.int32, "gather index")
    n = idx.numel()
    dst = torch.empty((n,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    row_bytes = src[0].numel() * src.element_size() if src.shape[0] > 0 else 0
    if n and row_bytes:
        lib.call("spot_gather_rows", _p(src), _p(dst), _p(idx), n, row_bytes, _st())
    return dst


def ema_lerp_(teacher_flat, student_flat, keep_rate):
    _chk(teacher_flat), _chk(student_flat)
    lib.call("spot_ema_lerp_f32", _p(teacher_flat), _p(student_flat), teacher_flat.numel(), float(keep_rate), _st())


def ema_lerp_i64_(teacher_flat, student_flat, keep_rate):
    _chk(teacher_flat, torch.int64), _chk(student_flat, torch.int64)
    lib.call("spot_ema_lerp_i64", _p(teacher_flat), _p(student_flat), teacher_flat.numel(), float(keep_rate), _st())


# ---------------------------------------------------------------------------
# boxes
# ---------------------------------------------------------------------------
def box_iou_pair(b1, b2, cxcywh=False, want_giou=True):
    _chk(b1), _chk(b2)
    P = b1.shape[0]
    f = lambda: torch.empty((P,), dtype=torch.float32, device=b1.device)
    giou, iou, uni = (f() if want_giou else None), f(), f()
    if P:
        lib.call("spot_box_iou_pair_f32", _p(b1), _p(b2), int(cxcywh), _p(giou), _p(iou), _p(uni), P, _st())
    return giou, iou, uni



_GRAD_AT_APPLY = True


class _Fn(torch.autograd.Function):
    """autograd.Function whose forward can tell whether a gradient can be asked for at all: inside forward() grad mode
    is always off and ctx.needs_input_grad reflects requires_grad only (True for parameters under torch.no_grad()),
    so the grad mode at the time of apply() is recorded here.  _wants(ctx[, i]) is what forward() should consult
    before it keeps or computes anything for backward (log-sum-exp, sign bytes, transformed inputs)."""

    @classmethod
    def apply(cls, *args, **kwargs):
        global _GRAD_AT_APPLY
        prev, _GRAD_AT_APPLY = _GRAD_AT_APPLY, torch.is_grad_enabled()
        try:
            return super().apply(*args, **kwargs)
        finally:
            _GRAD_AT_APPLY = prev


def _wants(ctx, i=None):
    if not _GRAD_AT_APPLY:
        return False
    return any(ctx.needs_input_grad) if i is None else bool(ctx.needs_input_grad[i])


class GIoUPairFn(_Fn):
    """(giou, iou) of paired cxcywh boxes; gradient flows to the first set only."""

    @staticmethod
    def forward(ctx, src, tgt):
        src, tgt = src.contiguous(), tgt.contiguous()
        giou, iou, _ = box_iou_pair(src, tgt, cxcywh=True)
        ctx.save_for_backward(src, tgt)
        return giou, iou

    @staticmethod
    def backward(ctx, dg, di):
        src, tgt = ctx.saved_tensors
        P = src.shape[0]
        out = torch.empty_like(src)
        if P:
            dg = dg.contiguous() if dg is not None else None
            di = di.contiguous() if di is not None else None
            lib.call("spot_box_giou_pair_bwd_f32", _p(src), _p(tgt), _p(dg), _p(di), _p(out), P, _st())
        return out, None


class TranstLossFn(_Fn):
    """(loss_ce, loss_bbox, loss_giou, iou, loss_iou_pred) of the set criterion as five 0-dim tensors from two launches;
    backward is one launch that reads the five upstream gradients on the device.  iou_pred / neg may be None (no IoU
    head); `iou` carries no gradient (transt_criterion.py:86 reports it only)."""

    @staticmethod
    def forward(ctx, logits, boxes, iou_pred, gt, pos, neg, w_obj, w_eos):
        logits, boxes, gt = logits.contiguous(), boxes.contiguous(), gt.contiguous()
        iou_pred = iou_pred.contiguous() if iou_pred is not None else None
        _chk(logits), _chk(boxes), _chk(gt), _chk(pos, torch.uint8, "pos")
        B, L = logits.shape[:2]
        if logits.shape != (B, L, 2) or boxes.shape != (B, L, 4) or gt.shape != (B, 4) or pos.shape != (B, L):
            raise RuntimeError("spot_amd: set criterion expects logits [B,L,2], boxes [B,L,4], targets [B,4], pos [B,L]")
        dev = logits.device
        out = torch.empty((8,), dtype=torch.float32, device=dev)
        ws = torch.empty((lib.load().spot_transt_loss_ws_floats(),), dtype=torch.float32, device=dev)
        lib.call("spot_transt_loss_f32", _p(logits), _p(boxes), _p(iou_pred), _p(gt), _p(pos), _p(neg), float(w_obj),
                 float(w_eos), B, L, _p(out), out.data_ptr() + 20, _p(ws), _st())
        ctx.w = (float(w_obj), float(w_eos))
        ctx.has_pred = iou_pred is not None
        if _wants(ctx):
            ctx.save_for_backward(logits, boxes, iou_pred, gt, pos, neg, out)
        losses = tuple(out[i] for i in range(5))
        ctx.mark_non_differentiable(losses[3])
        return losses

    @staticmethod
    def backward(ctx, g_ce, g_bbox, g_giou, _g_iou, g_pred):
        logits, boxes, iou_pred, gt, pos, neg, out = ctx.saved_tensors
        B, L = logits.shape[:2]
        need = ctx.needs_input_grad
        d_logits = torch.empty_like(logits) if need[0] else None
        d_boxes = torch.empty_like(boxes) if need[1] else None
        d_pred = torch.empty_like(iou_pred) if (ctx.has_pred and need[2]) else None
        lib.call("spot_transt_loss_bwd_f32", _p(logits), _p(boxes), _p(iou_pred), _p(gt), _p(pos), _p(neg), ctx.w[0],
                 ctx.w[1], out.data_ptr() + 20, _p(g_ce), _p(g_bbox), _p(g_giou), _p(g_pred), B, L, _p(d_logits),
                 _p(d_boxes), _p(d_pred), _st())
        return d_logits, d_boxes, d_pred, None, None, None, None, None


# ---------------------------------------------------------------------------
# crop + pad + resize of frames
# ---------------------------------------------------------------------------
def _crop_desc(frame, x0, y0, crop, pad):
    assert frame.is_cuda and frame.dtype == torch.uint8 and frame.dim() == 3 and frame.shape[2] == 3 and frame.is_contiguous()
    d = lib.CropDesc()
    d.image, d.H, d.W = frame.data_ptr(), frame.shape[0], frame.shape[1]
    d.x0, d.y0, d.crop = int(x0), int(y0), int(crop)
    for i in range(3):
        d.pad[i] = int(pad[i])
    return d


def crop_resize_u8(frame, x0, y0, crop, out_sz, pad, out=None):
    """Window [y0, y0+crop) x [x0, x0+crop) of a uint8 HWC device frame, `pad` outside the frame, resized to
    out_sz: uint8 [1, 3, out_sz, out_sz] (get_subwindow / sample_target of the reference, on the device); written into
    `out` when one of that shape is given."""
    if out is None:
        out = torch.empty((1, 3, out_sz, out_sz), dtype=torch.uint8, device=frame.device)
    elif tuple(out.shape) != (1, 3, out_sz, out_sz) or out.dtype != torch.uint8 or not out.is_contiguous() or out.device != frame.device:
        raise RuntimeError("spot_amd: crop_resize_u8(out=...) must be a contiguous uint8 [1, 3, %d, %d] tensor on the frame's device" % (out_sz, out_sz))
    d = _crop_desc(frame, x0, y0, crop, pad)
    lib.call("spot_crop_resize_u8", C.byref(d), _p(out), int(out_sz), _st())
    return out


def crop_resize_batch_u8(frames, windows, pads, out_sz):
    """One launch for a batch: frames[i] (uint8 HWC device tensors of any sizes), windows[i] = (x0, y0, crop),
    pads[i] = fill colour -> uint8 [B, 3, out_sz, out_sz]."""
    B = len(frames)
    dev = frames[0].device if B else "cuda"
    out = torch.empty((B, 3, out_sz, out_sz), dtype=torch.uint8, device=dev)
    if B == 0:
        return out
    arr = (lib.CropDesc * B)()
    for i, (f, (x0, y0, crop), pad) in enumerate(zip(frames, windows, pads)):
        arr[i] = _crop_desc(f, x0, y0, crop, pad)
    table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
    lib.call("spot_crop_resize_batch_u8", _p(table), B, _p(out), int(out_sz), _st())
    return out


# ---------------------------------------------------------------------------
# depthwise cross-correlation
# ---------------------------------------------------------------------------
class XCorrDepthwiseFn(_Fn):
    @staticmethod
    def forward(ctx, x, kernel):
        x, kernel = _chk(x.contiguous()), _chk(kernel.contiguous())
        B, Cn, hx, wx = x.shape
        _, _, hz, wz = kernel.shape
        out = torch.empty((B, Cn, hx - hz + 1, wx - wz + 1), dtype=torch.float32, device=x.device)
        lib.call("spot_xcorr_depthwise_f32", _p(x), _p(kernel), _p(out), B * Cn, hx, wx, hz, wz, _st())
        ctx.save_for_backward(x, kernel)
        return out

    @staticmethod
    def backward(ctx, g):
        x, kernel = ctx.saved_tensors
        B, Cn, hx, wx = x.shape
        _, _, hz, wz = kernel.shape
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dk = torch.empty_like(kernel) if ctx.needs_input_grad[1] else None
        lib.call("spot_xcorr_depthwise_bwd_f32", _p(x), _p(kernel), _p(g.contiguous()), _p(dx), _p(dk), B * Cn,
                 hx, wx, hz, wz, _st())
        return dx, dk


def xcorr_depthwise(x, kernel):
    """Drop-in for ocean_fusion.xcorr_depthwise (differentiable)."""
    return XCorrDepthwiseFn.apply(x, kernel)


class GroupDWFn(_Fn):
    @staticmethod
    def forward(ctx, z11, z12, z21, x11, x12, x21, weight):
        ts = [_chk(t.contiguous()) for t in (z11, z12, z21, x11, x12, x21, weight)]
        z11, z12, z21, x11, x12, x21, weight = ts
        B, Cn, hx, wx = x11.shape
        _, _, hz, wz = z11.shape
        assert tuple(x12.shape[2:]) == (hx - 2, wx) and tuple(x21.shape[2:]) == (hx, wx - 2), "branch shapes"
        assert tuple(z12.shape[2:]) == (hz - 2, wz) and tuple(z21.shape[2:]) == (hz, wz - 2), "branch shapes"
        out = torch.empty((B, Cn, hx - hz + 1, wx - wz + 1), dtype=torch.float32, device=x11.device)
        lib.call("spot_groupdw_f32", _p(x11), _p(x12), _p(x21), _p(z11), _p(z12), _p(z21), _p(weight), _p(out),
                 B * Cn, hx, wx, hz, wz, _st())
        ctx.save_for_backward(*ts)
        return out

    @staticmethod
    def backward(ctx, g):
        z11, z12, z21, x11, x12, x21, weight = ctx.saved_tensors
        B, Cn, hx, wx = x11.shape
        _, _, hz, wz = z11.shape
        dzs = [torch.empty_like(t) for t in (z11, z12, z21)]
        dxs = [torch.empty_like(t) for t in (x11, x12, x21)]
        part = torch.empty((B * Cn, 3), dtype=torch.float32, device=g.device)
        lib.call("spot_groupdw_bwd_f32", _p(x11), _p(x12), _p(x21), _p(z11), _p(z12), _p(z21), _p(weight),
                 _p(g.contiguous()), _p(dxs[0]), _p(dxs[1]), _p(dxs[2]), _p(dzs[0]), _p(dzs[1]), _p(dzs[2]),
                 _p(part), B * Cn, hx, wx, hz, wz, _st())
        dw = torch.empty_like(weight)
        lib.call("spot_groupdw_weight_grad_f32", _p(part), B * Cn, _p(weight), _p(dw), _st())
        return dzs[0], dzs[1], dzs[2], dxs[0], dxs[1], dxs[2], dw


def groupdw(zs, xs, weight):
    """Fused GroupDW.forward (differentiable): softmax(weight)-weighted sum of three depthwise correlations."""
    return GroupDWFn.apply(zs[0], zs[1], zs[2], xs[0], xs[1], xs[2], weight)


# ---------------------------------------------------------------------------
# gradient sinks
# ---------------------------------------------------------------------------
# A parameter whose .grad is a preallocated, zero-initialised view of a flat gradient buffer
# (arena.ParamArena.attach_grads marks it with ``_spot_grad_sink``) receives its gradient
# inside the producing kernel: the split-K GEMM / column-sum / LayerNorm reduction add
# straight into .grad and the autograd function returns None for it.  That removes one
# zero-fill and one add kernel per parameter per step.  The sink is told when a parameter's
# gradient is complete so that bucketed all-reduce can start while backward continues.
def _sink(p):
    if p is None or getattr(p, "_spot_grad_sink", None) is None:
        return None
    return p.grad


def _sink_expect(*params):
    """Forward-side twin of _sink_done: the op being recorded will deliver these parameters' gradients itself."""
    for p in params:
        if p is not None and p.requires_grad:
            s = getattr(p, "_spot_grad_sink", None)
            if s is not None and p.grad is not None:
                s.expect(p)


def _sink_done(*params):
    told = False
    for p in params:
        if p is not None:
            s = getattr(p, "_spot_grad_sink", None)
            if s is not None and p.grad is not None:
                s.grad_ready(p)
                told = True
    if told:
        _join_after_backward()


# Side streams (fusion.py runs the template stream of a layer on one).  Autograd only joins the streams of
# AccumulateGrad nodes with the caller's stream when backward ends; a gradient written into a sink on a side
# stream has no such node, so the join is queued explicitly, once per backward pass and side stream.
_SIDE_TO_MAIN = {}
_JOIN_QUEUED = set()


def register_side_stream(side, main):
    _SIDE_TO_MAIN[side.cuda_stream] = (side, main)


_NORM_WS = {}


def clip_grad_norm_(grad_flat, max_norm):
    """torch.nn.utils.clip_grad_norm_ over a flat gradient buffer, in place; returns the norm as a device tensor [1]
    (max_norm <= 0: the norm only)."""
    _chk(grad_flat, name="grad_flat")
    dev = grad_flat.device
    if dev not in _NORM_WS:
        _NORM_WS[dev] = torch.empty((lib.load().spot_clip_grad_norm_ws_floats() + 1,), dtype=torch.float32, device=dev)
    ws = _NORM_WS[dev]
    norm = torch.empty((1,), dtype=torch.float32, device=dev)
    lib.call("spot_clip_grad_norm_f32", _p(grad_flat), grad_flat.numel(), float(max_norm), _p(norm), _p(ws), _st())
    return norm


def compute_streams():
    """(side, main) stream pairs the network passes have used so far."""
    return list(_SIDE_TO_MAIN.values())


def _join_after_backward():
    cur = torch.cuda.current_stream().cuda_stream
    pair = _SIDE_TO_MAIN.get(cur)
    if pair is None or cur in _JOIN_QUEUED:
        return
    _JOIN_QUEUED.add(cur)
    side, main = pair

    def join():
        _JOIN_QUEUED.discard(cur)
        main.wait_stream(side)

    torch.autograd.Variable._execution_engine.queue_callback(join)


# ---------------------------------------------------------------------------
# dense contractions
# ---------------------------------------------------------------------------
_GEMM_TILE = 0   # workgroup tile handed to the library: 0 = it picks per problem; 64 / 128 force one (same results)


# Experiment switch, off in the product path ("fp32": every product exact).  "bf16x3-linear-fwd" takes the forward
# products of nn.Linear layers, "bf16x3" every GEMM launched from here (linear layers forward, input and weight
# gradients, 1x1 and patch-matrix convolutions, input_proj; not the Winograd products, which the library launches
# itself) on the bf16 matrix instruction with operands cut in three bf16 pieces (spot_gemm_desc.precision = 1), wherever
# the problem is made of whole 128 x 128 x 32 tiles of aligned operands; every other launch stays exact fp32.
MATMUL_PRECISION = "fp32"


def _bf16x3_ok(A, B, M, N, K, transA, transB, d, rowsum):
    """The library's conditions for precision = 1 (it refuses anything else)."""
    if M % 128 or N % 128 or K % 32 or rowsum is not None or (d.kseg and d.kseg % 32):
        return False
    if A.data_ptr() % 16 or B.data_ptr() % 16 or d.lda % 4 or d.ldb % 4 or d.strideA % 4 or d.strideB % 4 or d.segA % 4 or d.segB % 4:
        return False
    kseg = d.kseg or K
    ext_a = kseg * d.lda + M if transA else M * d.lda
    ext_b = kseg * d.ldb + N if transB else N * d.ldb
    return ext_a < 2 ** 31 and ext_b < 2 ** 31


_KS_MAX_TILES = 128         # 256 and 512 measure the same on the tracker (569 / 583 / 568 frames/s)


def _small_grid_ok(A, B, Cout, M, N, K, d, rowsum):
    """Launches of at most 128 tiles of 64x64 (half the CUs) that the library's tile = 32 kernel accepts."""
    if ((M + 63) // 64) * ((N + 63) // 64) * d.batch > _KS_MAX_TILES or M % 32 or N % 32 or K % 128 or (d.kseg and d.kseg % 128):
        return False
    if rowsum is not None or d.splitk > 1 or d.precision:
        return False
    ptrs = [A.data_ptr(), B.data_ptr(), Cout.data_ptr(), d.bias or 0, d.rowbias or 0, d.mask or 0, d.residual or 0]
    lds = [d.lda, d.ldb, d.ldc, d.strideA, d.strideB, d.strideC, d.segA, d.segB, d.ldrb, d.ldmask, d.ldres, d.strideRes,
           d.rb_cols]
    return all(p % 16 == 0 for p in ptrs) and all(v % 4 == 0 for v in lds)


def gemm(A, B, Cout, M, N, K, *, transA=False, transB=False, lda=None, ldb=None, ldc=None, batch=1,
         strideA=0, strideB=0, strideC=0, kseg=0, segA=0, segB=0, splitk=1, alpha=1.0, accumulate=False,
         bias=None, rowbias=None, rb_rows=0, ldrb=0, relu=False, mask=None, ldmask=0, residual=None,
         ldres=0, strideRes=0, drop_p=0.0, drop_seed=0, tile=0, rowsum=None, rb_cols=0, bn=None, precision=None):
    """Raw access to spot_gemm_f32; tensors are only used for their base pointers."""
    d = GemmDesc()
    if lib._keeping is not None:
        lib.keep(A, B, Cout, bias, rowbias, mask, residual, rowsum)
    d.A, d.B, d.C = A.data_ptr(), B.data_ptr(), Cout.data_ptr()
    d.M, d.N, d.K, d.batch = M, N, K, batch
    d.transA, d.transB = int(transA), int(transB)
    d.lda = lda if lda is not None else (M if transA else K)
    d.ldb = ldb if ldb is not None else (N if transB else K)
    d.ldc = ldc if ldc is not None else N
    d.strideA, d.strideB, d.strideC = strideA, strideB, strideC
    d.kseg, d.segA, d.segB = kseg, segA, segB
    d.splitk, d.alpha, d.accumulate = splitk, alpha, int(accumulate)
    d.bias = bias.data_ptr() if bias is not None else None
    d.rowbias = rowbias.data_ptr() if rowbias is not None else None
    d.rb_rows, d.ldrb = rb_rows, ldrb
    d.relu = int(relu)
    d.mask = mask.data_ptr() if mask is not None else None
    d.ldmask = ldmask
    d.residual = residual.data_ptr() if residual is not None else None
    d.ldres, d.strideRes = ldres, strideRes
    d.drop_p, d.drop_seed = drop_p, drop_seed
    d.tile = tile or _GEMM_TILE
    d.rowsum = rowsum.data_ptr() if rowsum is not None else None
    d.rb_cols = rb_cols
    # precision: 0 / 1 as given (the library refuses a 1 it cannot do); "try": bf16x3 if the problem qualifies;
    # None: by MATMUL_PRECISION
    if precision is None:
        precision = "try" if MATMUL_PRECISION == "bf16x3" else 0
    if precision == "try":
        precision = 1 if _bf16x3_ok(A, B, M, N, K, transA, transB, d, rowsum) else 0
    d.precision = precision
    if not d.tile and frozen_weights._active and not torch.is_grad_enabled() and _small_grid_ok(A, B, Cout, M, N, K, d, rowsum):
        d.tile = 32         # one clip (the online trackers): k-splitting 32x32 tile, see spot_gemm_desc.tile
    if bn is not None:          # (gamma, beta, running_mean, running_var, eps): inference BatchNorm over the rows
        d.bn_gamma, d.bn_beta, d.bn_mean, d.bn_var = (t.data_ptr() for t in bn[:4])
        d.bn_eps = float(bn[4])
    key = "gemm_%s%s" % ("t" if transA else "n", "n" if transB else "t") + ("_bf16x3" if precision else "")
    if TIMER is not None and TIMER.wants(key):
        if TIMER.shapes:
            key += " M%d N%d K%d b%d sk%d" % (M, N, K, batch, splitk)
        TIMER.run(key, 2.0 * M * N * K * batch,
                  lambda: lib.call("spot_gemm_f32", C.byref(d), _st()))
    else:
        lib.call("spot_gemm_f32", C.byref(d), _st())
    return Cout


def _view2d(t):
    """[..., C] contiguous -> (2-D view, leading dim)."""
    return t.reshape(-1, t.shape[-1])


def linear_fwd(x2, W, bias=None, relu=False, residual=None, rowbias=None, out=None, drop_p=0.0, drop_seed=0):
    """y[M,N] = x2[M,K] W[N,K]^T + bias (+ rowbias[m % R], over its own columns) -> relu -> dropout -> + residual."""
    M, K = x2.shape
    N = W.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=x2.device)
    precision = "try" if MATMUL_PRECISION == "bf16x3-linear-fwd" else None
    gemm(x2, W, out, M, N, K, lda=x2.stride(0), ldb=W.stride(0), ldc=out.stride(0), bias=bias, relu=relu,
         rowbias=rowbias, rb_rows=rowbias.shape[0] if rowbias is not None else 0,
         ldrb=rowbias.stride(0) if rowbias is not None else 0,
         rb_cols=rowbias.shape[1] if rowbias is not None else 0,
         residual=residual, ldres=residual.stride(0) if residual is not None else 0, drop_p=drop_p,
         drop_seed=drop_seed, precision=precision)
    return out


def dropout_apply(x2, p, seed):
    """keep(seed,row,col) ? x/(1-p) : 0 with the mask function of the GEMM epilogue."""
    if p <= 0:
        return x2
    out = torch.empty_like(x2)
    lib.call("spot_dropout_f32", _p(x2), _p(out), x2.shape[0], x2.shape[1], float(p), seed, _st())
    return out


# The gradient behind a block's output dropout, made by the LayerNorm that follows the block.  An attention or
# feed-forward block ends in  z = x + dropout(y);  its backward starts from  dropout_mask * dz,  and dz is what the
# backward of the LayerNorm applied to z returns.  The block notes (address of z, p, seed) when it has run; a LayerNorm
# called on exactly that tensor asks its backward kernel for the masked copy as a second output and files it under the
# address of the gradient it returns; the block's backward picks it up if the gradient it receives is that very
# tensor, unmodified (address, version and mask parameters match), and otherwise masks the gradient itself.
_OUT_DROP = None
_DROPPED = {}


def _note_out_drop(z, p, seed, wanted):
    """Called by a block's forward with its result z, its output dropout and whether a backward will follow."""
    global _OUT_DROP
    _OUT_DROP = (z.data_ptr(), float(p), int(seed)) if (p > 0 and wanted and z.is_cuda) else None


def _out_drop_of(x):
    """(p, seed) of the output dropout of the block that produced x, if x is the latest such block's result."""
    global _OUT_DROP
    hint, _OUT_DROP = _OUT_DROP, None
    if hint is not None and hint[0] == x.data_ptr():
        return hint[1], hint[2]
    return 0.0, 0


def _dropped_grad(dz2, p, seed):
    """mask(seed) * dz2 / (1 - p): from the LayerNorm backward that produced dz2, else by a pass over dz2."""
    hit = _DROPPED.pop(dz2.data_ptr(), None) if p > 0 else None
    if (hit is not None and hit[2] == dz2._version and hit[3] == float(p) and hit[4] == int(seed)
            and hit[1].shape == dz2.shape):
        return hit[1]
    return dropout_apply(dz2, p, seed)


def linear_dgrad(dy2, W, mask=None, out=None, accumulate=False, residual=None, alpha=1.0):
    """dx[M,K] = dy2[M,N] W[N,K]  (optionally zeroed where mask <= 0, plus residual, or added into out)."""
    M, N = dy2.shape
    K = W.shape[1]
    if out is None:
        out = torch.empty((M, K), dtype=torch.float32, device=dy2.device)
    gemm(dy2, W, out, M, K, N, transB=True, lda=dy2.stride(0), ldb=W.stride(0), ldc=out.stride(0), mask=mask,
         ldmask=mask.stride(0) if mask is not None else 0, accumulate=accumulate, residual=residual,
         ldres=residual.stride(0) if residual is not None else 0, alpha=alpha)
    return out


def _splitk_for(rows_out, cols_out, k):
    """K ranges of a weight-gradient product (its reduction runs over the token rows of the batch).

    Measured on MI355X (tools/probes/splitk_scan.py): the launch is fastest when the 64x64 output tiles times the
    ranges make about four workgroups per CU in whole multiples of the 256 CUs (a ragged last round leaves CUs with
    one workgroup more than their neighbours, and they set the time), and when a range still holds twenty k tiles
    (the atomic adds of a range's 64x64 partial tile cost as much as ten k tiles of products)."""
    t64 = ((rows_out + 63) // 64) * ((cols_out + 63) // 64)
    nkt = max(1, k // 32)
    sk = min(max(1, 1024 // t64), max(1, nkt // 20))
    step = 256 // math.gcd(t64, 256)
    if sk >= step:
        sk -= sk % step
    # never 1: with two or more ranges the kernel adds into the gradient with atomics, which is what makes it safe
    # for the template and the search pass to deliver a shared weight's gradient from two streams at once
    return max(2, sk)


def linear_wgrad(dy2, x2, out=None, accumulate=False, bias_out=None):
    """dW[N,K] = dy2[M,N]^T x2[M,K], reduction over the M token rows split across workgroups.

    bias_out [N] (+)= column sums of dy2 (the bias gradient), taken from the dy2 tiles the kernel stages anyway."""
    M, N = dy2.shape
    K = x2.shape[1]
    sk = _splitk_for(N, K, M)
    if out is None:
        out = torch.zeros((N, K), dtype=torch.float32, device=dy2.device)
    elif not accumulate:
        out.zero_()
    gemm(dy2, x2, out, N, K, M, transA=True, transB=True, lda=dy2.stride(0), ldb=x2.stride(0), ldc=out.stride(0),
         splitk=sk, accumulate=True, rowsum=bias_out)
    return out


def _bias_grad_buffer(sink, n, device):
    """Where a bias gradient is accumulated: the parameter's slice of the flat gradient buffer, or a cleared vector."""
    return sink if sink is not None else torch.zeros((n,), dtype=torch.float32, device=device)


def batch_sum(x2, R):
    """[nb*R, C] token rows (a column window of a wider buffer is fine) -> [R, C] summed over the nb clips."""
    M, Cn = x2.shape
    if M == R:
        return x2
    out = torch.empty((R, Cn), dtype=torch.float32, device=x2.device)
    lib.call("spot_batch_sum_f32", _p(x2), x2.stride(0), M // R, R, Cn, _p(out), _st())
    return out


def channel_sum(x4):
    """[B,C,H,W] -> [C] (the bias gradient of a convolution)."""
    x4 = x4.contiguous()
    B, Cn, H, W = x4.shape
    out = torch.empty((Cn,), dtype=torch.float32, device=x4.device)
    lib.call("spot_channel_sum_f32", _p(x4), B, Cn, H * W, _p(out), _st())
    return out


def colsum(x2, out=None):
    M, N = x2.shape
    if out is None:
        out = torch.zeros((N,), dtype=torch.float32, device=x2.device)
    lib.call("spot_colsum_f32", _p(x2), x2.stride(0), _p(out), M, N, _st())
    return out


def _linear_param_grads(pW, pb, need_W, need_b, dy2, x2):
    """(dW, db) of y = x W^T + b, written into the parameters' slices of the flat gradient buffer when they have one
    (then None is returned for that gradient); the bias gradient rides on the weight-gradient launch."""
    dW = db = None
    want_b = pb is not None and need_b
    gb = _sink(pb) if want_b else None
    if want_b and need_W:
        db = _bias_grad_buffer(gb, pW.shape[0], dy2.device)
    if need_W:
        gW = _sink(pW)
        dW = linear_wgrad(dy2, x2, out=gW, accumulate=gW is not None, bias_out=db)
        if gW is not None:
            dW = None
            _sink_done(pW)
    if want_b:
        if db is None:
            db = colsum(dy2, out=gb)
        if gb is not None:
            db = None
            _sink_done(pb)
    return dW, db


class LinearFn(_Fn):
    """nn.Linear on [..., K] inputs."""

    @staticmethod
    def forward(ctx, x, W, b):
        x2 = _view2d(x.contiguous())
        y = linear_fwd(x2, W, b)
        ctx.save_for_backward(x2, W)
        ctx.params = (W, b)
        _sink_expect(W if ctx.needs_input_grad[1] else None, b if ctx.needs_input_grad[2] else None)
        return y.view(*x.shape[:-1], W.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, W = ctx.saved_tensors
        pW, pb = ctx.params
        dy2 = _view2d(dy.contiguous())
        dx = linear_dgrad(dy2, W) if ctx.needs_input_grad[0] else None
        dW, db = _linear_param_grads(pW, pb, ctx.needs_input_grad[1], ctx.needs_input_grad[2], dy2, x2)
        return (dx.view(*dy.shape[:-1], W.shape[1]) if dx is not None else None), dW, db


class MLPFn(_Fn):
    """Linear -> ReLU -> ... -> Linear on [..., K] inputs (the prediction heads, ltr/models/tracking/transt_model.py:296-310).
    ReLU runs in each GEMM's epilogue; in backward its derivative is the mask operand of the next layer's input-gradient
    GEMM (zero where the stored activation is <= 0), so no element-wise pass exists in either direction."""

    @staticmethod
    def forward(ctx, x, *params):
        n = len(params) // 2
        hs = [_view2d(x.contiguous())]
        for i in range(n):
            hs.append(linear_fwd(hs[-1], params[2 * i], params[2 * i + 1], relu=i < n - 1))
        ctx.params = params
        if _wants(ctx):
            ctx.save_for_backward(*hs[:-1])
        for i in range(n):
            _sink_expect(params[2 * i] if ctx.needs_input_grad[1 + 2 * i] else None,
                         params[2 * i + 1] if (params[2 * i + 1] is not None and ctx.needs_input_grad[2 + 2 * i]) else None)
        return hs[-1].view(*x.shape[:-1], params[2 * n - 2].shape[0])

    @staticmethod
    def backward(ctx, dy):
        hs = ctx.saved_tensors
        params = ctx.params
        n = len(params) // 2
        need = ctx.needs_input_grad
        grads = [None] * (2 * n)
        d = _view2d(dy.contiguous())
        for i in range(n - 1, -1, -1):
            W, b = params[2 * i], params[2 * i + 1]
            grads[2 * i], grads[2 * i + 1] = _linear_param_grads(W, b, need[1 + 2 * i], need[2 + 2 * i], d, hs[i])
            if i > 0:
                d = linear_dgrad(d, W, mask=hs[i])
            else:
                d = linear_dgrad(d, W) if need[0] else None
        dx = d.view(*dy.shape[:-1], params[0].shape[1]) if d is not None else None
        return (dx,) + tuple(grads)


def linear(x, W, b=None):
    return LinearFn.apply(x, W, b)


def mlp(x, layers):
    """layers: the nn.Linear modules in order; ReLU between them."""
    return MLPFn.apply(x, *[t for l in layers for t in (l.weight, l.bias)])


class FFNFn(_Fn):
    """z = x + drop(linear2(drop(relu(linear1(x)))))  — the FFN sublayer with its residual and both
    dropouts (transt_fusion.py:151-152,234-235,240-241).  Dropout masks are regenerated from their
    seeds in backward; the hidden mask and relu' collapse into one test on the stored activation."""

    @staticmethod
    def forward(ctx, x, W1, b1, W2, b2, p):
        x2 = _view2d(x.contiguous())
        s1, s2 = (next_seed(), next_seed()) if p > 0 else (0, 0)
        h = linear_fwd(x2, W1, b1, relu=True, drop_p=p, drop_seed=s1)
        z = linear_fwd(h, W2, b2, residual=x2, drop_p=p, drop_seed=s2)
        ctx.save_for_backward(x2, h, W1, W2)
        ctx.drop = (p, s2)
        ctx.params = (W1, b1, W2, b2)
        _sink_expect(*ctx.params)
        _note_out_drop(z, p, s2, _wants(ctx))
        return z.view_as(x)

    @staticmethod
    def backward(ctx, dz):
        x2, h, W1, W2 = ctx.saved_tensors
        p, s2 = ctx.drop
        dz2 = _view2d(dz.contiguous())
        dzd = _dropped_grad(dz2, p, s2)                      # gradient behind the output dropout
        # h > 0 <=> relu passed and the hidden mask kept: one epilogue test, scaled by 1/(1-p)
        dh = linear_dgrad(dzd, W2, mask=h, alpha=1.0 / (1.0 - p))
        dx = linear_dgrad(dh, W1, residual=dz2)              # FFN path + residual path in one epilogue
        grads = []
        for (pw, pb), dyv, xv in zip(((ctx.params[0], ctx.params[1]), (ctx.params[2], ctx.params[3])), (dh, dzd), (x2, h)):
            gw, gb = _sink(pw), _sink(pb)
            db = _bias_grad_buffer(gb, pb.shape[0], dyv.device)
            dw = linear_wgrad(dyv, xv, out=gw, accumulate=gw is not None, bias_out=db)
            grads += [None if gw is not None else dw, None if gb is not None else db]
        _sink_done(*ctx.params)
        return (dx.view_as(dz), *grads, None)


# ---------------------------------------------------------------------------
# LayerNorm
# ---------------------------------------------------------------------------
class LayerNormFn(_Fn):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps, drop_p=0.0, drop_seed=0):
        x2 = _view2d(x.contiguous())
        rows, Cn = x2.shape
        y = torch.empty_like(x2)
        need = _wants(ctx)              # the row statistics are kept for backward only
        ctx.out_drop = (drop_p, drop_seed)      # output dropout of the block that produced x (see _dropped_grad)
        mean = torch.empty((rows,), dtype=torch.float32, device=x.device) if need else None
        rstd = torch.empty((rows,), dtype=torch.float32, device=x.device) if need else None
        lib.call("spot_layernorm_fwd_f32", _p(x2), _p(gamma), _p(beta), _p(y), _p(mean), _p(rstd), rows, Cn,
                 float(eps), _st())
        if need:
            ctx.save_for_backward(x2, mean, rstd, gamma)
            ctx.params = (gamma, beta)
            if _sink(gamma) is not None and _sink(beta) is not None:
                _sink_expect(gamma, beta)
        return y.view_as(x)

    @staticmethod
    def backward(ctx, dy):
        x2, mean, rstd, gamma = ctx.saved_tensors
        dy2 = _view2d(dy.contiguous())
        rows, Cn = x2.shape
        dx = torch.empty_like(x2)
        pg, pb = ctx.params
        sg, sb = _sink(pg), _sink(pb)
        direct = sg is not None and sb is not None
        dg = sg if direct else torch.empty_like(gamma)
        db = sb if direct else torch.empty_like(gamma)
        ws = torch.empty((lib.load().spot_layernorm_bwd_ws_floats(rows, Cn),), dtype=torch.float32, device=dy.device)
        dp, dseed = ctx.out_drop
        dxd = torch.empty_like(x2) if (dp > 0 and ctx.needs_input_grad[0]) else None
        lib.call("spot_layernorm_bwd_f32", _p(dy2), _p(x2), _p(mean), _p(rstd), _p(gamma), _p(dx), _p(dg), _p(db),
                 _p(ws), rows, Cn, int(direct), _p(dxd), float(dp), int(dseed), _st())
        if dxd is not None:
            _DROPPED[dx.data_ptr()] = (dx, dxd, dx._version, float(dp), int(dseed))
        if direct:
            _sink_done(pg, pb)
            return dx.view_as(dy), None, None, None, None, None
        return dx.view_as(dy), dg, db, None, None, None


def layer_norm(x, gamma, beta, eps=1e-5):
    dp, dseed = _out_drop_of(x) if torch.is_grad_enabled() else (0.0, 0)
    return LayerNormFn.apply(x, gamma, beta, eps, dp, dseed)


def layer_norm_infer(x, gamma, beta, eps=1e-5):
    x2 = _view2d(x)
    y = torch.empty_like(x2)
    lib.call("spot_layernorm_fwd_f32", _p(x2), _p(gamma), _p(beta), _p(y), None, None, x2.shape[0], x2.shape[1],
             float(eps), _st())
    return y.view_as(x)


# ---------------------------------------------------------------------------
# 1x1 convolutions of the ResNet bottleneck as GEMMs on the NCHW maps
# ---------------------------------------------------------------------------
class Conv1x1Fn(_Fn):
    """y[b] = W x[b] for a bias-free 1x1 convolution, NCHW in and out (resnet.py:77,83,131-134).

    Per clip the feature map is already a [C, H*W] row-major matrix, so forward, input gradient and weight
    gradient are batched GEMMs on it as it lies: no NHWC round trip (the convolution library transposes
    activations for its weight-gradient kernels), and the weight gradient is summed over the batch by
    split-K straight into the flat gradient buffer.

    With passthrough=True the input is handed on as a second output (the bottleneck's identity branch,
    resnet.py:86-90): the gradient that arrives for it is then added to W^T dy in that GEMM's epilogue instead of
    by a separate pass over the map after both gradients exist."""

    @staticmethod
    def forward(ctx, x, weight, passthrough=False):
        x = _chk(x.contiguous())
        B, Ci, H, W = x.shape
        Co, HW = weight.shape[0], H * W
        w2 = weight.view(Co, Ci)
        y = torch.empty((B, Co, H, W), dtype=torch.float32, device=x.device)
        gemm(w2, x, y, Co, HW, Ci, transB=True, lda=Ci, ldb=HW, ldc=HW, batch=B, strideB=Ci * HW, strideC=Co * HW)
        ctx.save_for_backward(x, weight)
        ctx.params = (weight,)
        if ctx.needs_input_grad[1]:
            _sink_expect(weight)
        if passthrough:
            return y, x.view_as(x)
        return y

    @staticmethod
    def backward(ctx, dy, dpass=None):
        x, weight = ctx.saved_tensors
        (pw,) = ctx.params
        B, Ci, H, W = x.shape
        Co, HW = weight.shape[0], H * W
        w2 = weight.view(Co, Ci)
        dy = dy.contiguous()
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            if dpass is not None:
                dpass = dpass.contiguous()
            gemm(w2, dy, dx, Ci, HW, Co, transA=True, transB=True, lda=Ci, ldb=HW, ldc=HW, batch=B,
                 strideB=Co * HW, strideC=Ci * HW, residual=dpass, ldres=HW, strideRes=Ci * HW)
        if ctx.needs_input_grad[1]:
            sk = _sink(pw)
            dw = sk.view(Co, Ci) if sk is not None else torch.zeros_like(w2)
            gemm(dy, x, dw, Co, Ci, B * HW, lda=HW, ldb=HW, ldc=Ci, kseg=HW, segA=Co * HW, segB=Ci * HW,
                 splitk=_splitk_for(Co, Ci, B * HW), accumulate=True)
            if sk is not None:
                dw = None
                _sink_done(pw)
            else:
                dw = dw.view_as(weight)
        return dx, dw, None


def conv1x1(x, conv, passthrough=False):
    """Bias-free 1x1 nn.Conv2d `conv` applied to x through the GEMM kernel (stride by subsampling first).
    passthrough=True (stride 1 only) returns (y, x'), x' being x for the identity branch of a residual block."""
    assert conv.kernel_size == (1, 1) and conv.bias is None and conv.groups == 1 and conv.padding == (0, 0)
    if conv.stride != (1, 1):
        assert not passthrough
        x = x[:, :, ::conv.stride[0], ::conv.stride[1]]
    if x.shape[2] * x.shape[3] % 32 != 0:       # the batch-summed weight gradient walks K in whole 32-wide tiles per clip:
        y = ConvIm2colFn.apply(x, conv.weight, 1, 0, (1, 1), None)      # other maps go through a copy at a padded pitch
        return (y, x) if passthrough else y
    return Conv1x1Fn.apply(x, conv.weight, passthrough)


def conv1x1_bn_act(x, conv, bn, relu=False, residual=None, passthrough=False):
    """[relu](bn(conv x) [+ residual]) for a bias-free 1x1 convolution.  With `bn` in inference mode (and no
    gradient wanted) the normalisation is a per-output-channel affine map and rides in the GEMM's epilogue together
    with the residual add and the ReLU: one launch, the map is written once.  Otherwise conv1x1 -> batch_norm_act.
    passthrough=True returns (result, x') with x' = x for the block's identity branch (see Conv1x1Fn)."""
    if passthrough:
        if torch.is_grad_enabled() and x.requires_grad:
            y, xres = conv1x1(x, conv, passthrough=True)
            return batch_norm_act(y, bn, relu=relu, residual=residual), xres
        return conv1x1_bn_act(x, conv, bn, relu=relu, residual=residual), x
    fold = (not bn.training and bn.track_running_stats and x.is_cuda and not torch.is_grad_enabled()
            and conv.kernel_size == (1, 1) and conv.bias is None and conv.groups == 1 and conv.padding == (0, 0))
    xs = x
    if fold:
        if conv.stride != (1, 1):
            xs = x[:, :, ::conv.stride[0], ::conv.stride[1]]
        xs = _chk(xs.contiguous())
        B, Ci, H, W = xs.shape
        Co, HW = conv.weight.shape[0], H * W
        fold = HW % 4 == 0
    if not fold:
        return batch_norm_act(conv1x1(x, conv), bn, relu=relu, residual=residual)
    x = xs
    if residual is not None:
        residual = _chk(residual.contiguous())
    y = torch.empty((B, Co, H, W), dtype=torch.float32, device=x.device)
    gemm(conv.weight.view(Co, Ci), x, y, Co, HW, Ci, transB=True, lda=Ci, ldb=HW, ldc=HW, batch=B, strideB=Ci * HW,
         strideC=Co * HW, bn=(bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps), residual=residual,
         ldres=HW, strideRes=Co * HW,
         relu=(2 if residual is not None else 1) if relu else 0)
    return y


class ConvIm2colFn(_Fn):
    """Convolution as patch matrix + GEMM, for what lies outside the Winograd path: the ResNet's 7x7 / stride 2 stem
    and layer2's 3x3 / stride 2 (resnet.py:61,79,195-196), and the Siamese head's 3x3 convolutions on odd-sized maps,
    some with a bias, some dilated along one axis (ocean_fusion.py:56-85,126-158).  Forward, input gradient (GEMM
    into patch space, then a gather) and weight gradient (one split-K GEMM over all clips, added into the flat
    gradient buffer) all run on the library's GEMM; the patch matrix of the forward pass is kept for the weight
    gradient.  Maps whose position count is not a multiple of 32 are worked on at a row pitch rounded up to one
    (zero-filled), so every GEMM keeps whole, aligned tiles; the result is narrowed on the way out."""

    @staticmethod
    def forward(ctx, x, weight, stride, pad, dil, bias=None):
        x = _chk(x.contiguous())
        dh, dw_ = (dil, dil) if isinstance(dil, int) else dil
        B, Ci, H, W = x.shape
        Co, _, kh, kw = weight.shape
        K = Ci * kh * kw
        Kpad = (K + 31) // 32 * 32
        Ho = (H + 2 * pad - dh * (kh - 1) - 1) // stride + 1
        Wo = (W + 2 * pad - dw_ * (kw - 1) - 1) // stride + 1
        N = Ho * Wo
        Np = (N + 31) // 32 * 32
        dev = x.device
        cols = torch.empty((B, Kpad, Np), dtype=torch.float32, device=dev)
        lib.call("spot_im2col_f32", _p(x), _p(cols), B, Ci, H, W, kh, kw, stride, pad, dh, dw_, Kpad, Np, _st())
        if Kpad == K:
            wp = weight.view(Co, K)
        else:
            wp = torch.zeros((Co, Kpad), dtype=torch.float32, device=dev)
            wp[:, :K] = weight.view(Co, K)
        y = torch.empty((B, Co, Np), dtype=torch.float32, device=dev)
        gemm(wp, cols, y, Co, Np, Kpad, transB=True, lda=Kpad, ldb=Np, ldc=Np, batch=B, strideB=Kpad * Np, strideC=Co * Np)
        if Np != N:
            y = y[:, :, :N] + bias.view(1, Co, 1) if bias is not None else y[:, :, :N].contiguous()
        elif bias is not None:
            y += bias.view(1, Co, 1)
        ctx.geo = (B, Ci, H, W, Co, kh, kw, stride, pad, dh, dw_, K, Kpad, N, Np)
        ctx.save_for_backward(cols if ctx.needs_input_grad[1] else None, wp, weight)
        ctx.params = (weight,)
        if ctx.needs_input_grad[1]:
            _sink_expect(weight)
        return y.view(B, Co, Ho, Wo)

    @staticmethod
    def backward(ctx, dy):
        cols, wp, weight = ctx.saved_tensors
        (pw,) = ctx.params
        B, Ci, H, W, Co, kh, kw, stride, pad, dh, dw_, K, Kpad, N, Np = ctx.geo
        dev = dy.device
        dbias = channel_sum(dy) if ctx.needs_input_grad[5] else None
        if Np != N:
            dyp = torch.zeros((B, Co, Np), dtype=torch.float32, device=dev)
            dyp[:, :, :N] = dy.reshape(B, Co, N)
            dy = dyp
        else:
            dy = dy.contiguous()
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dcols = torch.empty((B, Kpad, Np), dtype=torch.float32, device=dev)
            gemm(wp, dy, dcols, Kpad, Np, Co, transA=True, transB=True, lda=Kpad, ldb=Np, ldc=Np, batch=B,
                 strideB=Co * Np, strideC=Kpad * Np)
            dx = torch.empty((B, Ci, H, W), dtype=torch.float32, device=dev)
            lib.call("spot_col2im_f32", _p(dcols), _p(dx), B, Ci, H, W, kh, kw, stride, pad, dh, dw_, Kpad, Np, _st())
        if ctx.needs_input_grad[1]:
            # only the first K rows of the patch matrix carry weights: the product is taken over those and added
            # (split-K, atomically) straight into the gradient, padded or not
            sk = _sink(pw)
            dwk = sk.view(Co, K) if sk is not None else torch.zeros((Co, K), dtype=torch.float32, device=dev)
            gemm(dy, cols, dwk, Co, K, B * Np, lda=Np, ldb=Np, ldc=K, kseg=Np, segA=Co * Np, segB=Kpad * Np,
                 splitk=_splitk_for(Co, K, B * Np), accumulate=True)
            if sk is not None:
                _sink_done(pw)
            else:
                dw = dwk.reshape(weight.shape)
        return dx, dw, None, None, None, dbias


class DeformConvFn(_Fn):
    """Deformable convolution v1, bias-free, one group (dcn/deform_conv.py:12-110 as the aligned head uses it,
    ocean_fusion.py:174-192): the bilinearly sampled patch matrix is built by one kernel, after which forward,
    input-gradient and weight-gradient products are the GEMMs of ConvIm2colFn; one more kernel scatters the patch
    gradient into the input gradient and reduces it into the offset gradient."""

    @staticmethod
    def forward(ctx, x, offset, weight, stride, pad, dil, dgroups):
        x, offset = _chk(x.contiguous()), _chk(offset.contiguous())
        B, Ci, H, W = x.shape
        Co, Cw, kh, kw = weight.shape
        if Cw != Ci:
            raise RuntimeError("spot_amd: deformable convolution with groups > 1 is not supported")
        K = Ci * kh * kw
        Kpad = (K + 31) // 32 * 32
        Ho = (H + 2 * pad - dil * (kh - 1) - 1) // stride + 1
        Wo = (W + 2 * pad - dil * (kw - 1) - 1) // stride + 1
        if tuple(offset.shape) != (B, dgroups * 2 * kh * kw, Ho, Wo):
            raise RuntimeError("spot_amd: deformable convolution offsets must be [B, groups*2*kh*kw, Ho, Wo], got %s"
                               % (tuple(offset.shape),))
        N = Ho * Wo
        Np = (N + 31) // 32 * 32
        dev = x.device
        cols = torch.empty((B, Kpad, Np), dtype=torch.float32, device=dev)
        lib.call("spot_deform_im2col_f32", _p(x), _p(offset), _p(cols), B, Ci, H, W, kh, kw, stride, pad, dil, dgroups,
                 Kpad, Np, _st())
        if Kpad == K:
            wp = weight.view(Co, K)
        else:
            wp = torch.zeros((Co, Kpad), dtype=torch.float32, device=dev)
            wp[:, :K] = weight.view(Co, K)
        y = torch.empty((B, Co, Np), dtype=torch.float32, device=dev)
        gemm(wp, cols, y, Co, Np, Kpad, transB=True, lda=Kpad, ldb=Np, ldc=Np, batch=B, strideB=Kpad * Np, strideC=Co * Np)
        if Np != N:
            y = y[:, :, :N].contiguous()
        ctx.geo = (B, Ci, H, W, Co, kh, kw, stride, pad, dil, dgroups, K, Kpad, N, Np)
        ctx.save_for_backward(x, offset, cols if ctx.needs_input_grad[2] else None, wp, weight)
        ctx.params = (weight,)
        if ctx.needs_input_grad[2]:
            _sink_expect(weight)
        return y.view(B, Co, Ho, Wo)

    @staticmethod
    def backward(ctx, dy):
        x, offset, cols, wp, weight = ctx.saved_tensors
        (pw,) = ctx.params
        B, Ci, H, W, Co, kh, kw, stride, pad, dil, dgroups, K, Kpad, N, Np = ctx.geo
        dev = dy.device
        if Np != N:
            dyp = torch.zeros((B, Co, Np), dtype=torch.float32, device=dev)
            dyp[:, :, :N] = dy.reshape(B, Co, N)
            dy = dyp
        else:
            dy = dy.contiguous()
        dx = doff = dw = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            dcols = torch.empty((B, Kpad, Np), dtype=torch.float32, device=dev)
            gemm(wp, dy, dcols, Kpad, Np, Co, transA=True, transB=True, lda=Kpad, ldb=Np, ldc=Np, batch=B,
                 strideB=Co * Np, strideC=Kpad * Np)
            dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
            doff = ws = None
            if ctx.needs_input_grad[1]:
                doff = torch.empty_like(offset)
                ws = torch.empty((lib.load().spot_deform_col2im_ws_floats(B, Ci, H, W, kh, kw, stride, pad, dil, dgroups),),
                                 dtype=torch.float32, device=dev)
            lib.call("spot_deform_col2im_f32", _p(dcols), _p(x), _p(offset), _p(dx), _p(doff), _p(ws), B, Ci, H, W, kh, kw,
                     stride, pad, dil, dgroups, Kpad, Np, _st())
        if ctx.needs_input_grad[2]:
            # only the first K rows of the patch matrix carry weights: the product is taken over those and added
            # (split-K, atomically) straight into the gradient, padded or not
            sk = _sink(pw)
            dwk = sk.view(Co, K) if sk is not None else torch.zeros((Co, K), dtype=torch.float32, device=dev)
            gemm(dy, cols, dwk, Co, K, B * Np, lda=Np, ldb=Np, ldc=K, kseg=Np, segA=Co * Np, segB=Kpad * Np,
                 splitk=_splitk_for(Co, K, B * Np), accumulate=True)
            if sk is not None:
                _sink_done(pw)
            else:
                dw = dwk.reshape(weight.shape)
        return dx, doff, dw, None, None, None, None


def deform_conv(x, offset, weight, stride=1, padding=0, dilation=1, groups=1, deformable_groups=1):
    """Drop-in for dcn.deform_conv (deform_conv.py:186) for square stride / padding / dilation and groups = 1."""
    one = lambda v: v if isinstance(v, int) else (v[0] if v[0] == v[1] else None)
    stride, padding, dilation = one(stride), one(padding), one(dilation)
    if groups != 1 or None in (stride, padding, dilation):
        raise RuntimeError("spot_amd: deformable convolution supports groups = 1 and square stride / padding / dilation")
    if x.shape[0] == 0:                 # an empty batch: nothing to launch
        kh, kw = weight.shape[2:]
        return x.new_empty((0, weight.shape[0], (x.shape[2] + 2 * padding - dilation * (kh - 1) - 1) // stride + 1,
                            (x.shape[3] + 2 * padding - dilation * (kw - 1) - 1) // stride + 1))
    return DeformConvFn.apply(x, offset, weight, stride, padding, dilation, deformable_groups)


class Conv3x3GenFn(_Fn):
    """3x3 / stride 1 convolution with dilation 1 or 2 per axis and padding = pad * dilation, pad 0, 1 or 2, on a map
    of any size (optionally with bias) by the Winograd F(4x4, 3x3) path with masked partial tiles: the Siamese head's
    and backbone's 3x3 convolutions on 15x15 ... 63x63 maps (ocean_fusion.py:56-85,126-158; ocean_backbone.py:50-51).
    The input gradient is the same path on the output gradient with the filter rotated and pad 2 - pad; the weight
    gradient is summed in the transformed domain from the kept input transform and added straight into the flat
    gradient buffer."""

    @staticmethod
    def forward(ctx, x, weight, bias, pad, dil=(1, 1)):
        x = _chk(x.contiguous())
        B, Ci, H, W = x.shape
        Co = weight.shape[0]
        dh, dw_ = int(dil[0]), int(dil[1])
        L = lib.load()
        dev = x.device
        keep = _wants(ctx, 1)
        y = torch.empty((B, Co, H + (2 * pad - 2) * dh, W + (2 * pad - 2) * dw_), dtype=torch.float32, device=dev)
        ws = torch.empty((L.spot_conv3x3_winograd_general_ws_floats(B, Ci, Co, H, W, pad, dh, dw_),), dtype=torch.float32,
                         device=dev)
        v = (torch.empty((L.spot_conv3x3_winograd_general_v_floats(B, Ci, H, W, pad, dh, dw_),), dtype=torch.float32,
                         device=dev) if keep else None)
        opts = None
        if v is not None:
            opts = lib.WinogradOpts()
            opts.v_keep = v.data_ptr()
        lib.call("spot_conv3x3_winograd_general_f32", _p(x), _p(weight), _p(bias), _p(y), _p(ws),
                 C.byref(opts) if opts is not None else None, B, Ci, Co, H, W, int(pad), dh, dw_, 0, _st())
        ctx.save_for_backward(weight, v)
        ctx.params = (weight,)
        ctx.geo = (B, Ci, Co, H, W, pad, dh, dw_)
        if keep:
            _sink_expect(weight)
        return y

    @staticmethod
    def backward(ctx, dy):
        weight, v = ctx.saved_tensors
        (pw,) = ctx.params
        B, Ci, Co, H, W, pad, dh, dw_ = ctx.geo
        dy = dy.contiguous()
        L = lib.load()
        dev = dy.device
        dx = dw = dbias = None
        if ctx.needs_input_grad[2]:
            dbias = channel_sum(dy)
        if ctx.needs_input_grad[0]:
            Ho, Wo = dy.shape[2:]
            dx = torch.empty((B, Ci, H, W), dtype=torch.float32, device=dev)
            ws = torch.empty((L.spot_conv3x3_winograd_general_ws_floats(B, Co, Ci, Ho, Wo, 2 - pad, dh, dw_),),
                             dtype=torch.float32, device=dev)
            lib.call("spot_conv3x3_winograd_general_f32", _p(dy), _p(weight), None, _p(dx), _p(ws), None, B, Co, Ci, Ho, Wo,
                     int(2 - pad), dh, dw_, 1, _st())
        if ctx.needs_input_grad[1]:
            sk = _sink(pw)
            dw = sk if sk is not None else torch.empty_like(weight)
            ws = torch.empty((L.spot_conv3x3_winograd_general_ws_floats(B, Ci, Co, H, W, pad, dh, dw_),), dtype=torch.float32,
                             device=dev)
            lib.call("spot_conv3x3_winograd_general_wgrad_f32", None, _p(v), _p(dy), _p(dw), _p(ws), B, Ci, Co, H, W, int(pad),
                     dh, dw_, int(sk is not None), _st())
            if sk is not None:
                dw = None
                _sink_done(pw)
        return dx, dw, dbias, None, None


def _winograd_general_applies(x, conv):
    s, p, d = conv.stride, conv.padding, conv.dilation
    return (x.is_cuda and x.dtype == torch.float32 and conv.kernel_size == (3, 3) and conv.groups == 1
            and s == (1, 1) and isinstance(p, tuple) and p[0] == p[1] and conv.padding_mode == "zeros"
            and d[0] in (1, 2) and d[1] in (1, 2) and (p[0] == 0 or (d[0] == d[1] and p[0] in (d[0], 2 * d[0])))
            and x.shape[2] + 2 * p[0] > 2 * d[0] and x.shape[3] + 2 * p[0] > 2 * d[1]
            and conv.weight.shape[0] >= 16 and conv.weight.shape[1] >= 16)


def conv_bn_act(x, conv, bn, relu=False):
    """[relu](bn(conv x)) for any nn.Conv2d this library runs.  With `bn` in inference mode and no gradient wanted, a
    3x3 convolution on the any-size Winograd path takes the normalisation (and ReLU) in its output transform, and
    keeps its transformed weights inside `with frozen_weights():`; otherwise conv_general -> batch_norm_act."""
    infer = x.shape[0] > 0 and not bn.training and bn.track_running_stats and not torch.is_grad_enabled() and x.is_cuda
    fold = infer and _winograd_general_applies(x, conv)
    if not fold:
        if infer and _im2col_fold_applies(x, conv):
            return _conv_im2col_bn_act(x, conv, bn, relu)
        return batch_norm_act(conv_general(x, conv), bn, relu=relu)
    x = _chk(x.contiguous())
    B, Ci, H, W = x.shape
    weight = conv.weight
    Co = weight.shape[0]
    dh, dw_ = conv.dilation
    pad = conv.padding[0] // dh
    L = lib.load()
    y = torch.empty((B, Co, H + (2 * pad - 2) * dh, W + (2 * pad - 2) * dw_), dtype=torch.float32, device=x.device)
    ws = torch.empty((L.spot_conv3x3_winograd_general_ws_floats(B, Ci, Co, H, W, pad, dh, dw_),), dtype=torch.float32,
                     device=x.device)
    opts = lib.WinogradOpts()
    if frozen_weights._active:
        key = ("winograd", weight.data_ptr(), tuple(weight.shape), 0)
        hit = frozen_weights._tables.get(key)
        opts.u_valid = int(hit is not None)
        if hit is None:
            hit = frozen_weights._tables[key] = (torch.empty((36 * Co * Ci,), dtype=torch.float32, device=x.device), weight)
        opts.u_keep = hit[0].data_ptr()
    opts.bn_gamma, opts.bn_beta = bn.weight.data_ptr(), bn.bias.data_ptr()
    opts.bn_mean, opts.bn_var = bn.running_mean.data_ptr(), bn.running_var.data_ptr()
    opts.bn_eps, opts.relu = float(bn.eps), int(relu)
    lib.call("spot_conv3x3_winograd_general_f32", _p(x), _p(weight), _p(conv.bias), _p(y), _p(ws), C.byref(opts), B, Ci, Co,
             H, W, int(pad), int(dh), int(dw_), 0, _st())
    return y


def _im2col_fold_applies(x, conv):
    """Inference through the patch-matrix path with the normalisation in the product's epilogue: bias-free convolutions
    outside both Winograd paths (the 7x7 / stride 2 stem, the strided 3x3) whose output map is a whole number of
    32-position groups."""
    s, p, d = conv.stride, conv.padding, conv.dilation
    if not (conv.groups == 1 and conv.bias is None and s[0] == s[1] and isinstance(p, tuple) and p[0] == p[1]
            and conv.padding_mode == "zeros" and x.dtype == torch.float32):
        return False
    if winograd_applies(x, conv) or _winograd_general_applies(x, conv):
        return False
    kh, kw = conv.kernel_size
    Ho = (x.shape[2] + 2 * p[0] - d[0] * (kh - 1) - 1) // s[0] + 1
    Wo = (x.shape[3] + 2 * p[0] - d[1] * (kw - 1) - 1) // s[0] + 1
    return Ho > 0 and Wo > 0 and (Ho * Wo) % 32 == 0


def _conv_im2col_bn_act(x, conv, bn, relu):
    """[relu](bn(conv x)) in inference as patch matrix + one product whose epilogue normalises (per output channel =
    per row of the product) and clamps; inside `frozen_weights` the zero-padded weight matrix is kept between calls."""
    x = _chk(x.contiguous())
    B, Ci, H, W = x.shape
    weight = conv.weight
    Co, _, kh, kw = weight.shape
    stride, pad = conv.stride[0], conv.padding[0]
    dh, dw_ = conv.dilation
    K = Ci * kh * kw
    Kpad = (K + 31) // 32 * 32
    Ho = (H + 2 * pad - dh * (kh - 1) - 1) // stride + 1
    Wo = (W + 2 * pad - dw_ * (kw - 1) - 1) // stride + 1
    N = Ho * Wo
    dev = x.device
    cols = torch.empty((B, Kpad, N), dtype=torch.float32, device=dev)
    lib.call("spot_im2col_f32", _p(x), _p(cols), B, Ci, H, W, kh, kw, stride, pad, dh, dw_, Kpad, N, _st())
    if Kpad == K:
        wp = weight.view(Co, K)
    else:
        key = ("im2col weights", weight.data_ptr(), tuple(weight.shape))
        hit = frozen_weights._tables.get(key) if frozen_weights._active else None
        if hit is None:
            wp = torch.zeros((Co, Kpad), dtype=torch.float32, device=dev)
            wp[:, :K] = weight.view(Co, K)
            if frozen_weights._active:
                frozen_weights._tables[key] = (wp, weight)
        else:
            wp = hit[0]
    y = torch.empty((B, Co, Ho, Wo), dtype=torch.float32, device=dev)
    gemm(wp, cols, y, Co, N, Kpad, transB=True, lda=Kpad, ldb=N, ldc=N, batch=B, strideB=Kpad * N, strideC=Co * N,
         bn=(bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps), relu=1 if relu else 0)
    return y


def conv_general(x, conv):
    """nn.Conv2d `conv` applied to x: the general-size Winograd path for 3x3 / stride 1 convolutions with at least 16
    channels on both sides, patch matrix + GEMM for the rest of what has one group, square stride and padding and zero
    padding mode.  Anything else is not a convolution of this hot path and raises."""
    s, p, d = conv.stride, conv.padding, conv.dilation
    _chk(x, name="convolution input") if x.is_contiguous() else _chk(x.contiguous(), name="convolution input")
    if not (conv.groups == 1 and s[0] == s[1] and isinstance(p, tuple) and p[0] == p[1] and conv.padding_mode == "zeros"):
        raise RuntimeError("spot_amd: unsupported convolution %r (one group, square stride / padding, zero padding only)"
                           % (conv,))
    kh, kw = conv.kernel_size
    Ho = (x.shape[2] + 2 * p[0] - d[0] * (kh - 1) - 1) // s[0] + 1
    Wo = (x.shape[3] + 2 * p[0] - d[1] * (kw - 1) - 1) // s[0] + 1
    if Ho <= 0 or Wo <= 0:
        raise RuntimeError("spot_amd: convolution %r has no output on a %dx%d map" % (conv, x.shape[2], x.shape[3]))
    if x.shape[0] == 0:                 # an empty batch: nothing to launch
        return x.new_zeros((0, conv.weight.shape[0], Ho, Wo))
    if _winograd_general_applies(x, conv):
        return Conv3x3GenFn.apply(x, conv.weight, conv.bias, p[0] // d[0], (d[0], d[1]))
    return ConvIm2colFn.apply(x, conv.weight, s[0], p[0], (d[0], d[1]), conv.bias)


class MaxPool3x3s2Fn(_Fn):
    """nn.MaxPool2d(3, stride=2, padding=1) (resnet.py:113,199): one byte per output remembers the winning window
    position, the backward gathers through it."""

    @staticmethod
    def forward(ctx, x):
        x = _chk(x.contiguous())
        B, Cn, H, W = x.shape
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        y = torch.empty((B, Cn, Ho, Wo), dtype=torch.float32, device=x.device)
        win = torch.empty((B, Cn, Ho, Wo), dtype=torch.uint8, device=x.device) if _wants(ctx, 0) else None
        lib.call("spot_maxpool3x3s2_fwd_f32", _p(x), _p(y), _p(win), B * Cn, H, W, _st())
        ctx.save_for_backward(win)
        ctx.shape = (B, Cn, H, W)
        return y

    @staticmethod
    def backward(ctx, dy):
        (win,) = ctx.saved_tensors
        B, Cn, H, W = ctx.shape
        dx = torch.empty((B, Cn, H, W), dtype=torch.float32, device=dy.device)
        lib.call("spot_maxpool3x3s2_bwd_f32", _p(dy.contiguous()), _p(win), _p(dx), B * Cn, H, W, _st())
        return dx


def max_pool_3x3_s2(x, pool):
    """nn.MaxPool2d `pool` applied to x: this library's kernels for the stems' 3x3 / stride 2 / padding 1 pool; other
    pools are not part of this hot path and raise."""
    ks = pool.kernel_size if isinstance(pool.kernel_size, tuple) else (pool.kernel_size,) * 2
    st = pool.stride if isinstance(pool.stride, tuple) else (pool.stride,) * 2
    pd = pool.padding if isinstance(pool.padding, tuple) else (pool.padding,) * 2
    dl = pool.dilation if isinstance(pool.dilation, tuple) else (pool.dilation,) * 2
    if (x.is_cuda and ks == (3, 3) and st == (2, 2) and pd == (1, 1) and dl == (1, 1) and not pool.ceil_mode
            and not pool.return_indices and x.dtype == torch.float32):
        return MaxPool3x3s2Fn.apply(x)
    raise RuntimeError("spot_amd: unsupported max-pool %r on a %s %s map (3x3 / stride 2 / padding 1 on float32 GPU maps "
                       "only)" % (pool, tuple(x.shape), x.dtype))


# ---------------------------------------------------------------------------
# 3x3 convolutions of the ResNet bottleneck
# ---------------------------------------------------------------------------
def _winograd(inp, weight, Cout, flip, dilation=1, keep_v=False, bn=None, relu=False):
    """keep_v: also return the transformed input [36, Cin, tiles] (for the weight gradient).
    bn (an eval()-mode nn.BatchNorm2d) / relu: folded into the output transform (inference only)."""
    B, Cin, H, W = inp.shape
    out = torch.empty((B, Cout, H, W), dtype=torch.float32, device=inp.device)
    ws = torch.empty((lib.load().spot_conv3x3_winograd_ws_floats(B, Cin, Cout, H, W),), dtype=torch.float32,
                     device=inp.device)
    v = torch.empty((36, Cin, B * (H // 4) * (W // 4)), dtype=torch.float32, device=inp.device) if keep_v else None
    opts = None
    frozen = frozen_weights._active and not torch.is_grad_enabled()
    if keep_v or bn is not None or frozen:
        opts = lib.WinogradOpts()
        opts.v_keep = v.data_ptr() if v is not None else None
        if frozen:                                   # transformed weights kept across calls (see frozen_weights)
            key = ("winograd", weight.data_ptr(), tuple(weight.shape), int(flip))
            hit = frozen_weights._tables.get(key)
            opts.u_valid = int(hit is not None)
            if hit is None:
                hit = frozen_weights._tables[key] = (torch.empty((36 * Cout * Cin,), dtype=torch.float32,
                                                                 device=inp.device), weight)
            opts.u_keep = hit[0].data_ptr()
        if bn is not None:
            opts.bn_gamma, opts.bn_beta = bn.weight.data_ptr(), bn.bias.data_ptr()
            opts.bn_mean, opts.bn_var = bn.running_mean.data_ptr(), bn.running_var.data_ptr()
            opts.bn_eps, opts.relu = float(bn.eps), int(relu)
    lib.call("spot_conv3x3_winograd_f32", _p(inp), _p(weight), _p(out), _p(ws), C.byref(opts) if opts is not None else None,
             B, Cin, Cout, H, W, int(dilation), int(flip), _st())
    return (out, v) if keep_v else out


class Conv3x3Fn(_Fn):
    """Bias-free 3x3 convolution, stride 1, padding 1 (resnet.py:79-81): forward and input gradient by the
    Winograd F(4x4, 3x3) path of the library; the weight gradient likewise (summed in the transformed domain and
    added straight into the flat gradient buffer)."""

    @staticmethod
    def forward(ctx, x, weight, dilation=1):
        x = _chk(x.contiguous())
        keep = _wants(ctx, 1)                            # the transformed input serves the weight gradient too
        y = _winograd(x, weight, weight.shape[0], False, dilation, keep_v=keep)
        y, v = y if keep else (y, None)
        ctx.save_for_backward(None if keep else x, weight, v)
        ctx.params = (weight,)
        ctx.dilation = dilation
        ctx.xshape = tuple(x.shape)
        if keep:
            _sink_expect(weight)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, v = ctx.saved_tensors
        (pw,) = ctx.params
        dy = dy.contiguous()
        dil = ctx.dilation
        dx = _winograd(dy, weight, weight.shape[1], True, dil) if ctx.needs_input_grad[0] else None
        dw = None
        if ctx.needs_input_grad[1]:
            B, Cin, H, W = ctx.xshape
            Cout = weight.shape[0]
            sk = _sink(pw)
            dw = sk if sk is not None else torch.empty_like(weight)
            ws = torch.empty((lib.load().spot_conv3x3_winograd_wgrad_ws_floats(B, Cin, Cout, H, W),), dtype=torch.float32,
                             device=dy.device)
            lib.call("spot_conv3x3_winograd_wgrad_f32", _p(x), _p(v), _p(dy), _p(dw), _p(ws), B, Cin, Cout, H, W, dil,
                     int(sk is not None), _st())
            if sk is not None:
                dw = None
                _sink_done(pw)
        return dx, dw, None


class BnReluConv3x3Fn(_Fn):
    """conv2(relu(bn1(x))) of a bottleneck in training mode (resnet.py:77-81), x = the raw output of conv1.

    The normalised map has one reader, the 3x3 convolution's Winograd input transform, so it is never written: a
    statistics pass over x, then the input transform normalises and clamps each element on its way in (and writes the
    ReLU sign bytes).  One pass over the map and one launch less than batch_norm_act -> conv3x3, same values bit for
    bit.  Backward: the convolution's two gradients as Conv3x3Fn, then the BatchNorm backward as BatchNormActFn."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, tracked, eps, momentum, weight, deferred=None):
        x = _chk(x.contiguous())
        B, Cin, H, W = x.shape
        Cout = weight.shape[0]
        dev = x.device
        L = lib.load()
        need = _wants(ctx)
        keep_v = _wants(ctx, 8)                           # the transformed input serves the weight gradient too
        stats_ws = torch.empty((L.spot_batchnorm_ws_floats(B, Cin),), dtype=torch.float32, device=dev)
        lib.call("spot_batchnorm_stats_f32", _p(x), _p(stats_ws), B, Cin, H * W, _st())
        mean = torch.empty((Cin,), dtype=torch.float32, device=dev)
        rstd = torch.empty((Cin,), dtype=torch.float32, device=dev)
        sign = torch.empty((x.numel() // 4,), dtype=torch.uint8, device=dev) if need else None
        ib = lib.BnTrainIn()
        ib.stats_ws, ib.gamma, ib.beta = stats_ws.data_ptr(), gamma.data_ptr(), beta.data_ptr()
        ib.running_mean = running_mean.data_ptr() if running_mean is not None else None
        ib.running_var = running_var.data_ptr() if running_var is not None else None
        ib.num_batches_tracked = tracked.data_ptr() if tracked is not None else None
        ib.save_mean, ib.save_rstd = mean.data_ptr(), rstd.data_ptr()
        ib.relu_sign = sign.data_ptr() if sign is not None else None
        ib.deferred_stats = deferred.data_ptr() if deferred is not None else None
        ib.eps, ib.momentum = float(eps), float(momentum)
        lib.keep(stats_ws, gamma, beta, running_mean, running_var, tracked, mean, rstd, sign, deferred)
        out = torch.empty((B, Cout, H, W), dtype=torch.float32, device=dev)
        ws = torch.empty((L.spot_conv3x3_winograd_ws_floats(B, Cin, Cout, H, W),), dtype=torch.float32, device=dev)
        v = torch.empty((36, Cin, B * (H // 4) * (W // 4)), dtype=torch.float32, device=dev) if keep_v else None
        opts = lib.WinogradOpts()
        opts.v_keep = v.data_ptr() if v is not None else None
        opts.in_bn = C.pointer(ib)
        lib.call("spot_conv3x3_winograd_f32", _p(x), _p(weight), _p(out), _p(ws), C.byref(opts), B, Cin, Cout, H, W, 1, 0,
                 _st())
        if need:
            ctx.save_for_backward(x, sign, mean, rstd, gamma, weight, v)
            ctx.params = (gamma, beta, weight)
            if _sink(gamma) is not None and _sink(beta) is not None:
                _sink_expect(gamma, beta)
            if keep_v:
                _sink_expect(weight)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, sign, mean, rstd, gamma, weight, v = ctx.saved_tensors
        pg, pb, pw = ctx.params
        B, Cin, H, W = x.shape
        Cout = weight.shape[0]
        dev = dout.device
        dout = dout.contiguous()
        L = lib.load()
        dw = None
        if ctx.needs_input_grad[8]:
            sk = _sink(pw)
            dw = sk if sk is not None else torch.empty_like(weight)
            ws = torch.empty((L.spot_conv3x3_winograd_wgrad_ws_floats(B, Cin, Cout, H, W),), dtype=torch.float32, device=dev)
            lib.call("spot_conv3x3_winograd_wgrad_f32", None, _p(v), _p(dout), _p(dw), _p(ws), B, Cin, Cout, H, W, 1,
                     int(sk is not None), _st())
            if sk is not None:
                dw = None
                _sink_done(pw)
        dx = dg = db = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            dy = _winograd(dout, weight, Cin, True, 1)       # gradient behind the ReLU
            dx = torch.empty_like(x)
            sg, sb = _sink(pg), _sink(pb)
            direct = sg is not None and sb is not None
            dg = sg if direct else torch.empty_like(gamma)
            db = sb if direct else torch.empty_like(gamma)
            ws = torch.empty((L.spot_batchnorm_ws_floats(B, Cin),), dtype=torch.float32, device=dev)
            lib.call("spot_batchnorm_bwd_f32", _p(dy), _p(x), _p(sign), _p(mean), _p(rstd), _p(gamma), _p(dx), None,
                     _p(dg), _p(db), _p(ws), B, Cin, H * W, int(direct), _st())
            if direct:
                _sink_done(pg, pb)
                dg = db = None
        return dx, dg, db, None, None, None, None, None, dw, None


FUSE_BN_INTO_CONV3X3 = True     # scheduling switch: False runs batch_norm_act -> conv3x3 (same values)


def bn_relu_conv3x3_applies(bn, conv, B, H, W, is_cuda):
    """Training-mode BatchNorm + ReLU in front of a bias-free 3x3 / stride 1 / padding 1 convolution on a B x . x H x W
    map of the Winograd path, dilation 1 (the dilated one works on polyphase sub-maps whose packs of four are not the
    map's)."""
    return (FUSE_BN_INTO_CONV3X3 and is_cuda and B > 0 and (bn.training or not bn.track_running_stats)
            and conv.kernel_size == (3, 3) and conv.stride == (1, 1) and conv.dilation == (1, 1) and conv.padding == (1, 1)
            and conv.groups == 1 and conv.bias is None and H % 4 == 0 and W % 4 == 0)


def bn_relu_conv3x3(x, bn, conv):
    """conv(relu(bn(x))) with bn in training mode: see BnReluConv3x3Fn (statistics handling as batch_norm_act)."""
    if bn.momentum is None:
        raise RuntimeError("spot_amd: cumulative-average BatchNorm (momentum=None) is not supported")
    track = bn.track_running_stats
    pending = defer_running_stats._active
    if track and pending is not None:
        return BnReluConv3x3Fn.apply(x, bn.weight, bn.bias, None, None, None, bn.eps, bn.momentum, conv.weight,
                                     pending.buffer_for(bn))
    return BnReluConv3x3Fn.apply(x, bn.weight, bn.bias, bn.running_mean if track else None,
                                 bn.running_var if track else None, bn.num_batches_tracked if track else None, bn.eps,
                                 bn.momentum, conv.weight)


def conv3x3(x, conv):
    """nn.Conv2d `conv` applied to x: the Winograd path for bias-free 3x3 / stride 1 convolutions with dilation d = 1
    or 2 and padding d on maps whose sides are multiples of 4d; everything else goes to conv_general (any-size
    Winograd, or patch matrix + GEMM for strided ones)."""
    if winograd_applies(x, conv):
        return Conv3x3Fn.apply(x, conv.weight, conv.dilation[0])
    return conv_general(x, conv)


def winograd_applies(x, conv):
    B, _, H, W = x.shape
    d = conv.dilation[0]
    return (x.is_cuda and B > 0 and conv.kernel_size == (3, 3) and conv.stride == (1, 1)
            and d in (1, 2) and conv.dilation == (d, d) and conv.padding == (d, d) and conv.groups == 1
            and conv.bias is None and H % (4 * d) == 0 and W % (4 * d) == 0)


def conv3x3_bn_act(x, conv, bn, relu=False):
    """[relu](bn(conv x)) for the bottleneck's 3x3 convolution.  With `bn` in inference mode and no gradient wanted
    the normalisation and the ReLU ride in the Winograd output transform; otherwise conv3x3 -> batch_norm_act."""
    if (not bn.training and bn.track_running_stats and not torch.is_grad_enabled() and winograd_applies(x, conv)):
        return _winograd(_chk(x.contiguous()), conv.weight, conv.weight.shape[0], False, conv.dilation[0], bn=bn, relu=relu)
    if winograd_applies(x, conv):
        return batch_norm_act(conv3x3(x, conv), bn, relu=relu)
    return conv_bn_act(x, conv, bn, relu=relu)


# ---------------------------------------------------------------------------
# BatchNorm2d + ReLU (+ residual) of the ResNet bottleneck
# ---------------------------------------------------------------------------
class BatchNormActFn(_Fn):
    """y = [relu](batch_norm(x) [+ residual]) in training mode (batch statistics, running stats updated)."""

    @staticmethod
    def forward(ctx, x, residual, gamma, beta, running_mean, running_var, tracked, eps, momentum, relu,
                deferred=None):
        x = _chk(x.contiguous())
        N, Cn, H, W = x.shape
        HW = H * W
        if residual is not None:
            residual = _chk(residual.contiguous())
        dev = x.device
        y = torch.empty_like(x)
        mean = torch.empty((Cn,), dtype=torch.float32, device=dev)
        rstd = torch.empty((Cn,), dtype=torch.float32, device=dev)
        need = _wants(ctx)
        # one byte per pack of four elements, or per element on maps whose H*W is not a multiple of 4
        sign = (torch.empty((x.numel() // 4 if HW % 4 == 0 else x.numel(),), dtype=torch.uint8, device=dev)
                if (relu and need) else None)
        ws = torch.empty((lib.load().spot_batchnorm_ws_floats(N, Cn),), dtype=torch.float32, device=dev)
        lib.call("spot_batchnorm_fwd_train_f32", _p(x), _p(residual), _p(gamma), _p(beta), _p(running_mean),
                 _p(running_var), _p(tracked), _p(y), _p(mean), _p(rstd), _p(sign), _p(deferred), _p(ws), N, Cn, HW,
                 float(eps), float(momentum), int(relu), _st())
        if need:
            ctx.save_for_backward(x, sign, mean, rstd, gamma)
            ctx.params = (gamma, beta)
            ctx.has_res = residual is not None
            if _sink(gamma) is not None and _sink(beta) is not None:
                _sink_expect(gamma, beta)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, sign, mean, rstd, gamma = ctx.saved_tensors
        pg, pb = ctx.params
        N, Cn, H, W = x.shape
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        dres = torch.empty_like(x) if (ctx.has_res and ctx.needs_input_grad[1]) else None
        sg, sb = _sink(pg), _sink(pb)
        direct = sg is not None and sb is not None
        dg = sg if direct else torch.empty_like(gamma)
        db = sb if direct else torch.empty_like(gamma)
        ws = torch.empty((lib.load().spot_batchnorm_ws_floats(N, Cn),), dtype=torch.float32, device=dy.device)
        lib.call("spot_batchnorm_bwd_f32", _p(dy), _p(x), _p(sign), _p(mean), _p(rstd), _p(gamma), _p(dx), _p(dres),
                 _p(dg), _p(db), _p(ws), N, Cn, H * W, int(direct), _st())
        if direct:
            _sink_done(pg, pb)
            dg = db = None
        return dx, dres, dg, db, None, None, None, None, None, None, None


class _BnDeferred(C.Structure):
    _fields_ = [("running_mean", C.c_void_p), ("running_var", C.c_void_p), ("num_batches_tracked", C.c_void_p),
                ("stats", C.c_void_p), ("C", C.c_int), ("momentum", C.c_float)]


class defer_running_stats:
    """Context in which training-mode batch_norm_act leaves the running-statistics update pending.

    For a forward pass that runs on a second stream beside another pass through the same BatchNorm layers:
    both would read-modify-write the running buffers.  The deferred pass stores its batch statistics in a
    per-layer buffer instead; ``apply()`` (on the main stream, after the streams have joined) performs all
    pending updates in one launch, so the buffers end up as if the passes had run one after the other."""
    _active = None
    _tables = {}

    def __enter__(self):
        self.layers = []
        self._prev = defer_running_stats._active
        defer_running_stats._active = self
        return self

    def __exit__(self, *exc):
        defer_running_stats._active = self._prev
        return False

    def buffer_for(self, bn):
        buf = getattr(bn, "_spot_deferred_stats", None)
        if buf is None or buf.device != bn.weight.device:
            buf = torch.empty((2 * bn.num_features,), dtype=torch.float32, device=bn.weight.device)
            bn._spot_deferred_stats = buf
        self.layers.append(bn)
        return buf

    def apply(self):
        if not self.layers:
            return
        key = tuple((id(bn), bn.running_mean.data_ptr(), bn._spot_deferred_stats.data_ptr()) for bn in self.layers)
        table = defer_running_stats._tables.get(key)
        if table is None:
            arr = (_BnDeferred * len(self.layers))()
            for e, bn in zip(arr, self.layers):
                e.running_mean, e.running_var = bn.running_mean.data_ptr(), bn.running_var.data_ptr()
                e.num_batches_tracked = bn.num_batches_tracked.data_ptr()
                e.stats, e.C, e.momentum = bn._spot_deferred_stats.data_ptr(), bn.num_features, bn.momentum
            host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
            table = host.to(self.layers[0].weight.device)
            defer_running_stats._tables[key] = table
        lib.call("spot_batchnorm_apply_deferred", _p(table), len(self.layers), _st())


def batch_norm_act(x, bn, relu=False, residual=None):
    """nn.BatchNorm2d `bn` applied to x, then (+ residual) and (ReLU), as one fused op.
    Training mode uses batch statistics and updates bn's running buffers like the module would."""
    if bn.training or not bn.track_running_stats:
        if bn.momentum is None:
            raise RuntimeError("spot_amd: cumulative-average BatchNorm (momentum=None) is not supported")
        track = bn.track_running_stats
        pending = defer_running_stats._active
        if track and pending is not None:
            return BatchNormActFn.apply(x, residual, bn.weight, bn.bias, None, None, None, bn.eps, bn.momentum, relu,
                                        pending.buffer_for(bn))
        return BatchNormActFn.apply(x, residual, bn.weight, bn.bias, bn.running_mean if track else None,
                                    bn.running_var if track else None, bn.num_batches_tracked if track else None,
                                    bn.eps, bn.momentum, relu)
    x = _chk(x.contiguous())
    if torch.is_grad_enabled() and (x.requires_grad or bn.weight.requires_grad):
        raise RuntimeError("spot_amd: BatchNorm in eval mode is an inference path here; wrap it in torch.no_grad()")
    N, Cn, H, W = x.shape
    y = torch.empty_like(x)
    if residual is not None:
        residual = _chk(residual.contiguous())
    lib.call("spot_batchnorm_fwd_infer_f32", _p(x), _p(residual), _p(bn.weight), _p(bn.bias), _p(bn.running_mean),
             _p(bn.running_var), _p(y), N, Cn, H * W, float(bn.eps), int(relu), _st())
    return y


# ---------------------------------------------------------------------------
# attention
# ---------------------------------------------------------------------------
def _mha_key_splits(B, Lq, Lk, nhead):
    """Key ranges per query tile for an inference forward whose grid of (query tile, head, clip) workgroups would
    leave most of the 256 CUs idle (the online tracker: one clip): enough ranges of whole 64-key tiles to reach a few
    hundred workgroups, at most 8."""
    wgs = (Lq + 127) // 128 * nhead * B
    if wgs >= 128:
        return 1
    return max(1, min((Lk + 63) // 64, 8, 384 // wgs))


def mha_drop_bits(B, nhead, Lq, Lk, device):
    """Buffer for the keep bits an attention forward with dropout hands to its backward (one bit per query and key)."""
    return torch.empty((lib.load().spot_mha_drop_bits_words(B, nhead, Lq, Lk),), dtype=torch.int32, device=device)


def mha_core_fwd(q, ldq, k, ldk, v, ldv, B, Lq, Lk, nhead, dh, kpm, want_lse, out=None, drop_p=0.0, drop_seed=0,
                 bf16=False, drop_bits=None):
    dev = q.device
    if out is None:
        out = torch.empty((B * Lq, nhead * dh), dtype=torch.float32, device=dev)
    nsplit = _mha_key_splits(B, Lq, Lk, nhead) if not (want_lse or bf16 or drop_p > 0) else 1
    if nsplit > 1:
        ws = torch.empty((lib.load().spot_mha_fwd_splitk_ws_floats(B, Lq, nhead, nsplit),), dtype=torch.float32, device=dev)
        lib.call("spot_mha_fwd_splitk_f32", _p(q), ldq, _p(k), ldk, _p(v), ldv, _p(out), out.stride(0), _p(kpm), _p(ws), nsplit,
                 B, Lq, Lk, nhead, dh, 1.0 / math.sqrt(dh), _st())
        return out, None
    lse = torch.empty((B, nhead, Lq), dtype=torch.float32, device=dev) if want_lse else None
    launch = lambda: lib.call("spot_mha_fwd_bf16" if bf16 else "spot_mha_fwd_f32", _p(q), ldq, _p(k), ldk, _p(v), ldv, _p(out), out.stride(0), _p(lse),
                              _p(kpm), B, Lq, Lk, nhead, dh, 1.0 / math.sqrt(dh), float(drop_p), drop_seed, _p(drop_bits), _st())
    if TIMER is not None and TIMER.wants("mha_fwd"):
        TIMER.run("mha_fwd Lq%d Lk%d B%d" % (Lq, Lk, B) if TIMER.shapes else "mha_fwd", 4.0 * B * nhead * Lq * Lk * dh, launch)
    else:
        launch()
    return out, lse


def _timed_mha_bwd(B, Lq, Lk, nhead, dh, launch):
    if TIMER is not None and TIMER.wants("mha_bwd"):
        TIMER.run("mha_bwd Lq%d Lk%d B%d" % (Lq, Lk, B) if TIMER.shapes else "mha_bwd",
                  10.0 * B * nhead * Lq * Lk * dh, launch)
    else:
        launch()


class frozen_weights:
    """Context for inference with weights that do not change between calls (the online tracker): the position
    tables pos W^T (+ b), which depend on weights and position encodings only, are computed once and reused.
    Entries are keyed by the operands' addresses, so the context must not span an update of the weights
    (optimizer step, EMA, load_state_dict); leaving it keeps the entries, clear() drops them."""
    _active = 0
    _tables = {}

    def __enter__(self):
        frozen_weights._active += 1
        return self

    def __exit__(self, *exc):
        frozen_weights._active -= 1
        return False

    @staticmethod
    def clear():
        frozen_weights._tables.clear()

    @staticmethod
    def forget(module):
        """Drop the entries derived from this module's parameters (after its weights changed) and leave those of
        other networks, whose captured graphs may still read them."""
        import bisect
        spans = sorted((p.data_ptr(), p.data_ptr() + p.numel() * p.element_size()) for p in module.parameters())
        starts = [a for a, _ in spans]

        def inside(addr):                # keys hold addresses of parameters or of slices of them
            i = bisect.bisect_right(starts, addr) - 1
            return i >= 0 and addr < spans[i][1]
        ids = {id(p) for p in module.parameters()}

        def derived(value):              # entries keep their operands: a parameter itself or a view of one
            return any(torch.is_tensor(t) and (id(t) in ids or (t._base is not None and id(t._base) in ids)) for t in value)
        for key in [k for k, v in frozen_weights._tables.items()
                    if derived(v) or any(isinstance(e, int) and e > 4096 and inside(e) for e in k)]:
            del frozen_weights._tables[key]


def _pos_table(pos2, W, b):
    """pos2[R,C] W[N,C]^T + b -> [R,N]: the additive term of (x + pos) W^T + b that does not depend on x."""
    if frozen_weights._active and not torch.is_grad_enabled():
        key = (pos2.data_ptr(), W.data_ptr(), b.data_ptr() if b is not None else 0, tuple(pos2.shape), tuple(W.shape))
        hit = frozen_weights._tables.get(key)
        if hit is None:
            hit = frozen_weights._tables[key] = (linear_fwd(pos2, W, b), pos2, W, b)   # operands kept alive: addresses stay theirs
        return hit[0]
    return linear_fwd(pos2, W, b)


class MHAFn(_Fn):
    """z = xq + out_proj(MHA(q = xq + pos_q, k = xkv + pos_k, v = xkv))   (residual optional).

    nn.MultiheadAttention as the fusion layers use it (transt_fusion.py:145-149,
    212-214, 223-232) plus the residual add that follows it.  xq [B,Lq,C],
    xkv [B,Lk,C]; pos_q [Rq,C] / pos_k [Rk,C] with R = L (shared by the batch) or
    B*L; kpm [B,Lk] uint8 or None.  Positional terms enter as per-row bias
    tables (pos W^T + b) added in the projection GEMM's epilogue, so x + pos is
    never materialised.  ``add_residual`` adds the query stream xq to the output
    (every call site of the reference does, :149,214,232); its gradient is then
    folded into the query-projection dgrad epilogue instead of a separate add.
    """

    @staticmethod
    def forward(ctx, xq, xkv, pos_q, pos_k, in_w, in_b, out_w, out_b, kpm, add_residual, nhead, self_attn,
                p_attn=0.0, p_out=0.0, bf16=False):
        B, Lq, Cn = xq.shape
        Lk = xkv.shape[1]
        dh = Cn // nhead
        xq2, xkv2 = _view2d(xq.contiguous()), _view2d(xkv.contiguous())
        dev = xq.device
        if self_attn:
            # (x + pos) Wqk^T + b = x Wqk^T + b + pos Wqk^T: the last term is a [R, 2C] table over the q|k columns
            table = _pos_table(pos_q, in_w[:2 * Cn], None)
            qkv = linear_fwd(xq2, in_w, in_b, rowbias=table)
            q, k, v = qkv, qkv[:, Cn:], qkv[:, 2 * Cn:]
            ldq = ldk = ldv = 3 * Cn
            kv = None
        else:
            tq = _pos_table(pos_q, in_w[:Cn], in_b[:Cn])
            tk = _pos_table(pos_k, in_w[Cn:2 * Cn], None)             # [Rk, C] table over the k columns of k|v
            qkv = linear_fwd(xq2, in_w[:Cn], None, rowbias=tq)
            kv = linear_fwd(xkv2, in_w[Cn:], in_b[Cn:], rowbias=tk)
            q, k, v = qkv, kv, kv[:, Cn:]
            ldq, ldk, ldv = Cn, 2 * Cn, 2 * Cn
        need_grad = _wants(ctx)
        sa = next_seed() if p_attn > 0 else 0
        so = next_seed() if p_out > 0 else 0
        # with dropout the forward leaves the mask it drew as bits for the backward
        keep = mha_drop_bits(B, nhead, Lq, Lk, dev) if need_grad and p_attn > 0 else None
        attn, lse = mha_core_fwd(q, ldq, k, ldk, v, ldv, B, Lq, Lk, nhead, dh, kpm, need_grad, drop_p=p_attn,
                                 drop_seed=sa, bf16=bf16, drop_bits=keep)
        z = linear_fwd(attn, out_w, out_b, residual=xq2 if add_residual else None, drop_p=p_out, drop_seed=so)
        if need_grad:
            ctx.save_for_backward(xq2, xkv2, pos_q, pos_k, in_w, out_w, qkv, kv, attn, lse, kpm, keep)
            ctx.dims = (B, Lq, Lk, Cn, nhead, dh, self_attn, bool(add_residual))
            ctx.drop = (p_attn, sa, p_out, so)
            ctx.bwd_entry = "spot_mha_bwd_bf16" if bf16 else "spot_mha_bwd_f32"
            ctx.params = (in_w, in_b, out_w, out_b)
            _sink_expect(*ctx.params)
        _note_out_drop(z, p_out, so, need_grad)
        return z.view(B, Lq, Cn)

    @staticmethod
    def backward(ctx, dz):
        xq2, xkv2, pos_q, pos_k, in_w, out_w, qkv, kv, attn, lse, kpm, keep = ctx.saved_tensors
        B, Lq, Lk, Cn, nhead, dh, self_attn, has_res = ctx.dims
        dev = dz.device
        p_attn, sa, p_out, so = ctx.drop
        dz2 = _view2d(dz.contiguous())
        dzd = _dropped_grad(dz2, p_out, so)                  # gradient behind the output dropout
        dattn = linear_dgrad(dzd, out_w)
        p_in_w, p_in_b, p_out_w, p_out_b = ctx.params
        sinks = [_sink(t) for t in ctx.params]
        d_out_b = _bias_grad_buffer(sinks[3], Cn, dev)
        d_out_w = linear_wgrad(dzd, attn, out=sinks[2], accumulate=sinks[2] is not None, bias_out=d_out_b)
        scale = 1.0 / math.sqrt(dh)
        d_in_w = sinks[0] if sinks[0] is not None else torch.zeros_like(in_w)
        d_in_b = sinks[1] if sinks[1] is not None else torch.zeros((3 * Cn,), dtype=torch.float32, device=dev)
        if self_attn:
            dqkv = torch.empty_like(qkv)
            _timed_mha_bwd(B, Lq, Lk, nhead, dh, lambda: lib.call(
                ctx.bwd_entry, _p(qkv), 3 * Cn, _p(qkv[:, Cn:]), 3 * Cn, _p(qkv[:, 2 * Cn:]), 3 * Cn,
                _p(attn), Cn, _p(dattn), Cn, _p(lse), _p(kpm), None,
                _p(dqkv), 3 * Cn, _p(dqkv[:, Cn:]), 3 * Cn, _p(dqkv[:, 2 * Cn:]), 3 * Cn,
                B, Lq, Lk, nhead, dh, scale, float(p_attn), sa, _p(keep), _st()))
            dxq = linear_dgrad(dqkv, in_w, residual=dz2 if has_res else None)
            dxkv = None
            linear_wgrad(dqkv, xq2, out=d_in_w, accumulate=True, bias_out=d_in_b)
            # (x + pos) W^T: the pos part of dW is (sum over clips of dq|dk)^T pos
            R = pos_q.shape[0]
            linear_wgrad(batch_sum(dqkv[:, :2 * Cn], R), pos_q, out=d_in_w[:2 * Cn], accumulate=True)
        else:
            dq = torch.empty_like(qkv)
            dkv = torch.empty_like(kv)
            _timed_mha_bwd(B, Lq, Lk, nhead, dh, lambda: lib.call(
                ctx.bwd_entry, _p(qkv), Cn, _p(kv), 2 * Cn, _p(kv[:, Cn:]), 2 * Cn,
                _p(attn), Cn, _p(dattn), Cn, _p(lse), _p(kpm), None,
                _p(dq), Cn, _p(dkv), 2 * Cn, _p(dkv[:, Cn:]), 2 * Cn,
                B, Lq, Lk, nhead, dh, scale, float(p_attn), sa, _p(keep), _st()))
            dxq = linear_dgrad(dq, in_w[:Cn], residual=dz2 if has_res else None)
            dxkv = linear_dgrad(dkv, in_w[Cn:])
            linear_wgrad(dq, xq2, out=d_in_w[:Cn], accumulate=True, bias_out=d_in_b[:Cn])
            linear_wgrad(dkv, xkv2, out=d_in_w[Cn:], accumulate=True, bias_out=d_in_b[Cn:])
            Rq, Rk = pos_q.shape[0], pos_k.shape[0]
            linear_wgrad(batch_sum(dq, Rq), pos_q, out=d_in_w[:Cn], accumulate=True)
            linear_wgrad(batch_sum(dkv[:, :Cn], Rk), pos_k, out=d_in_w[Cn:2 * Cn], accumulate=True)
        _sink_done(*ctx.params)
        pg = [None if sk is not None else g for sk, g in zip(sinks, (d_in_w, d_in_b, d_out_w, d_out_b))]
        # self-attention: xq and xkv are the same tensor, the whole gradient goes to xq
        return (dxq.view(B, Lq, Cn), None if self_attn else dxkv.view(B, Lk, Cn), None, None, *pg, None, None, None,
                None, None, None, None)


def mha_block(xq, xkv, pos_q, pos_k, in_w, in_b, out_w, out_b, kpm, add_residual, nhead, self_attn,
              p_attn=0.0, p_out=0.0, bf16=False):
    """bf16=True takes the attention core's matrix products in bf16 (fp32 softmax and accumulation)."""
    return MHAFn.apply(xq, xkv, pos_q, pos_k, in_w, in_b, out_w, out_b, kpm, add_residual, nhead, self_attn,
                       p_attn, p_out, bf16)
