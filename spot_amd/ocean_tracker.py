"""Online tracker of the Siamese variant (pysot_toolkit/trackers/ocean_tracker.py:11-288) on the HIP kernels: the
frame is cut, padded and resized on the device, normalised by the preprocess kernel, run through Ocean.track, and
reduced to the winning cell by one kernel; eight float64 values come back to the host per frame, where the
reference's scalar box update follows."""
import time
import warnings

import numpy as np
import torch

from . import lib, ops


class OceanConfig(object):
    """ocean_tracker.py:278-288."""
    penalty_k = 0.062
    window_influence = 0.38
    lr = 0.765
    windowing = "cosine"
    exemplar_size = 127
    instance_size = 255
    total_stride = 8
    score_size = (instance_size - exemplar_size) // total_stride + 1 + 8
    context_amount = 0.5
    ratio = 0.94


class Ocean_Tracker(object):
    def __init__(self, name, net, use_graph=None):
        """use_graph: replay a frame's device work (normalise, network, decode) as one captured HIP graph from the
        third frame of a sequence on.  Default: on for this package's Ocean model (whose track() is a pure device
        function of the crop and the cached template features); off for any other net."""
        self.name, self.net = name, net
        self.p = OceanConfig()
        self.state = None
        self.mean, self.std = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
        if use_graph is None:
            from .ocean_model import Ocean
            use_graph = isinstance(net, Ocean)
        self.use_graph = bool(use_graph)
        self._graph, self._eager_frames = None, 0
        self._target_dev = None

    def initialize_features(self):
        if not getattr(self, "features_initialized", False) and hasattr(self.net, "initialize"):
            self.net.initialize()
        self.features_initialized = True

    @staticmethod
    def python2round(f):
        """ocean_tracker.py:212-218."""
        if round(f + 1) - round(f) != 1:
            return f + abs(f) / f * 0.5
        return round(f)

    def _device_frame(self, im):
        if torch.is_tensor(im):
            return im if im.is_cuda else im.cuda()
        held = getattr(self, "_frame", None)
        if held is None or held[0] is not im:
            self._frame = held = (im, torch.from_numpy(np.ascontiguousarray(im)).cuda(non_blocking=True))
        return held[1]

    def get_subwindow_tracking(self, im, pos, model_sz, original_sz, avg_chans):
        """SiamFC-type crop (ocean_tracker.py:225-275; note its window origin is round(pos - c), Python's rounding):
        uint8 [1, 3, model_sz, model_sz] on the device."""
        if isinstance(pos, float):
            pos = [pos, pos]
        c = (original_sz + 1) / 2
        x0, y0 = round(pos[0] - c), round(pos[1] - c)
        pad = np.asarray(avg_chans, dtype=np.float64).astype(np.uint8)
        return ops.crop_resize_u8(self._device_frame(im), int(x0), int(y0), int(original_sz), int(model_sz), pad)

    def _normalize(self, crop_u8):
        return ops.preprocess_u8(crop_u8, None, self.mean, self.std)

    def initialize(self, image, info: dict) -> dict:
        tic = time.time()
        p = OceanConfig()
        state = {"im_h": image.shape[0], "im_w": image.shape[1]}
        self.initialize_features()
        bbox = info["init_bbox"]
        target_pos = np.array([bbox[0] + bbox[2] / 2, bbox[1] + bbox[3] / 2])
        target_sz = np.array([bbox[2], bbox[3]])
        wc_z = target_sz[0] + p.context_amount * sum(target_sz)
        hc_z = target_sz[1] + p.context_amount * sum(target_sz)
        s_z = round(np.sqrt(wc_z * hc_z))
        avg_chans = (image.float().mean(dim=(0, 1)).cpu().numpy() if torch.is_tensor(image) else np.mean(image, axis=(0, 1)))
        z_crop = self.get_subwindow_tracking(image, target_pos, p.exemplar_size, s_z, avg_chans)
        self._frame = None              # the uploaded copy serves this call only: the caller may reuse its buffer
        if isinstance(self.net, torch.nn.Module):
            ops.frozen_weights.forget(self.net)     # the network may have been trained since the last sequence
        with torch.no_grad(), ops.frozen_weights():     # transformed convolution weights are kept across frames
            self.net.template(self._normalize(z_crop))
        if p.windowing == "cosine":
            window = np.outer(np.hanning(p.score_size), np.hanning(p.score_size))
        else:
            window = np.ones((int(p.score_size), int(p.score_size)))
        self._window_dev = torch.from_numpy(np.ascontiguousarray(window)).cuda()
        self._graph, self._eager_frames = None, 0        # a captured graph reads the previous template's features
        state.update(p=p, net=self.net, avg_chans=avg_chans, window=window, target_pos=target_pos, target_sz=target_sz)
        self.state = state
        return {"time": time.time() - tic}

    def decode(self, cls_score, bbox_pred, cls_align, p):
        """Device part of update(): the winning cell as eight float64 values (see spot_ocean_tracker_decode); the
        target size is read from self._target_dev."""
        S = int(p.score_size)
        out = torch.empty((8,), dtype=torch.float64, device=cls_score.device)
        lib.call("spot_ocean_tracker_decode", ops._p(cls_score.contiguous()), ops._p(cls_align.contiguous()),
                 ops._p(bbox_pred.contiguous()), ops._p(self._window_dev), ops._p(self._target_dev), float(p.ratio),
                 float(p.penalty_k), float(p.window_influence), S, int(p.total_stride), int(p.instance_size), ops._p(out),
                 ops._st())
        return out

    def _device_pipeline(self, net, x_crops, p):
        cls_score, bbox_pred, cls_align = net.track(x_crops)
        return self.decode(cls_score, bbox_pred, cls_align, p)

    def _run_frame(self, net, x_crops, p):
        if not self.use_graph or x_crops is None:
            return self._device_pipeline(net, x_crops, p)
        if self._graph is None:
            if self._eager_frames < 2:
                self._eager_frames += 1
                return self._device_pipeline(net, x_crops, p)
            try:
                self._x_static = x_crops.clone()
                cap = torch.cuda.Stream()
                cap.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(cap):             # per-stream handles and code objects exist before capture
                    self._device_pipeline(net, self._x_static, p)
                torch.cuda.synchronize()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=cap):
                    self._out_static = self._device_pipeline(net, self._x_static, p)
                self._graph = graph
            except Exception as e:                       # capture unsupported: stay eager, say so once
                warnings.warn("spot_amd Ocean tracker: HIP graph capture failed (%s); running eagerly" % (e,))
                torch.cuda.synchronize()
                self.use_graph = False
                return self._device_pipeline(net, x_crops, p)
        self._x_static.copy_(x_crops)
        self._graph.replay()
        return self._out_static

    def update(self, net, x_crops, target_pos, target_sz, window, scale_z, p):
        """ocean_tracker.py:71-130."""
        with torch.no_grad(), ops.frozen_weights():
            tgt = torch.tensor([float(target_sz[0]), float(target_sz[1])], dtype=torch.float64)
            if self._target_dev is None:
                self._target_dev = tgt.cuda()
            else:
                self._target_dev.copy_(tgt)
            res = self._run_frame(net, x_crops, p).cpu().numpy()                               # the frame's one D2H read
        corners, penalty, score = res[2:6], res[6], np.float32(res[7])          # (x1, y1, x2, y2) of the winning cell
        # crop pixels -> frame pixels: offset of the box centre from the crop centre and the box extent, both / scale_z
        centre, extent = (corners[:2] + corners[2:]) / 2, corners[2:] - corners[:2]
        shift, extent = (centre - p.instance_size // 2) / scale_z, extent / scale_z
        previous = target_sz / scale_z
        lr = penalty * score * p.lr                                             # size learning rate (ocean_tracker.py:118)
        blended = extent * lr + (1 - lr) * previous
        return np.asarray(target_pos, dtype=np.float64) + shift, previous * (1 - lr) + lr * blended, score

    def track(self, image, info: dict = None) -> dict:
        state = self.state
        p, net, avg_chans, window = state["p"], state["net"], state["avg_chans"], state["window"]
        target_pos, target_sz = state["target_pos"], state["target_sz"]
        context = p.context_amount * sum(target_sz)
        s_z = np.sqrt((target_sz[0] + context) * (target_sz[1] + context))     # side of the template-scale window
        scale_z = p.exemplar_size / s_z
        s_x = s_z + 2 * ((p.instance_size - p.exemplar_size) / 2 / scale_z)     # widened to the search crop
        x_crop = self.get_subwindow_tracking(image, target_pos, p.instance_size, self.python2round(s_x), avg_chans)
        self._frame = None
        target_pos, target_sz, cls_score = self.update(net, self._normalize(x_crop), target_pos, target_sz * scale_z,
                                                       window, scale_z, p)
        frame = (state["im_w"], state["im_h"])
        for axis in (0, 1):                                    # centre inside the frame, sides between 10 px and the frame's
            target_pos[axis] = max(0, min(frame[axis], target_pos[axis]))
            target_sz[axis] = max(10, min(frame[axis], target_sz[axis]))
        state["target_pos"], state["target_sz"] = target_pos, target_sz
        self.state = state
        return {"target_bbox": [target_pos[0] - target_sz[0] / 2, target_pos[1] - target_sz[1] / 2, target_sz[0], target_sz[1]],
                "best_score": cls_score}
