"""Online tracker API: host mirror of pysot_toolkit/trackers/transt_tracker.py:12-193 and
pysot_toolkit/trackers/net_wrappers.py:4-56.

``TransT_Tracker(name, net, window_penalty, exemplar_size, instance_size)`` with
``initialize(image, {'init_bbox': [x,y,w,h]}) -> {'time'}`` and
``track(image) -> {'target_bbox': [x,y,w,h], 'best_score': pscore}``.

Per frame: the frame is uploaded once as uint8 HWC; one kernel cuts the window,
pads it with the channel average and resizes it (OpenCV's 8-bit INTER_LINEAR
arithmetic); the crop is normalised by the preprocess kernel, pushed through
backbone + fusion + heads, and the 1024-token score map is reduced on the device;
one small buffer (the chosen box and the pscore map the reference also returns)
comes back per frame.  From the third frame of a sequence the device work
replays as one captured HIP graph.
"""
import math
import time

import numpy as np
import torch

from . import lib, ops


class NetWithBackbone:
    """Lazy network holder (net_wrappers.py:4-56): .initialize() moves to GPU and sets eval()."""

    def __init__(self, net_path=None, use_gpu=True, initialize=False, net=None, **kwargs):
        self.net_path = net_path
        self.use_gpu = use_gpu
        self.net = net
        self.net_kwargs = kwargs
        if initialize:
            self.initialize()

    def __getattr__(self, name):
        if name in ("net", "net_path", "use_gpu", "net_kwargs"):
            raise AttributeError(name)
        return getattr(self.net, name)

    def load_network(self):
        ops.frozen_weights.clear()          # a (re)loaded network: position tables of earlier weights are stale
        if self.net is None:
            from .transt import transt_resnet50
            from .synthetic import Settings
            self.net = transt_resnet50(Settings(device="cuda" if self.use_gpu else "cpu"))
            if self.net_path is not None:
                ckpt = torch.load(self.net_path, map_location="cpu", weights_only=True)
                self.net.load_state_dict(ckpt["net"] if "net" in ckpt else ckpt)
        if self.use_gpu:
            self.net.cuda()
        self.net.eval()

    def initialize(self):
        self.load_network()

    def template(self, z):
        with torch.no_grad(), ops.frozen_weights():
            self.net.template(z)

    def track(self, x):
        with torch.no_grad(), ops.frozen_weights():
            return self.net.track(x)


class TransT_Tracker(object):
    def __init__(self, name, net, window_penalty=0.49, exemplar_size=128, instance_size=256, use_graph=None):
        self.name = name
        self.net = net
        self.window_penalty = window_penalty
        self.exemplar_size = exemplar_size
        self.instance_size = instance_size
        self.mean = [0.485, 0.456, 0.406]
        self.std = [0.229, 0.224, 0.225]
        # A frame is ~250 kernel launches at batch 1 that keep the GPU busy for 3.2 ms, while enqueueing them from
        # Python takes 8 ms: the eager frame is host-bound.  From the third frame of a sequence on, the whole device
        # pipeline (normalise, backbone, fusion, heads, decode) therefore replays as one captured HIP graph (frames
        # 1-2 run eagerly): 8.2 -> 2.3 ms per frame on MI355X.
        # use_graph=False keeps every frame eager.  A replayed graph repeats exactly the
        # device work of the captured frame, so by default only this package's own network holder is trusted to be
        # a pure device function of the crop; any other `net` object runs eagerly unless use_graph=True is passed.
        if use_graph is None:
            use_graph = isinstance(net, NetWithBackbone)
        self.use_graph = bool(use_graph)
        self._graph = None
        self._eager_frames = 0

    @staticmethod
    def _crop_side(size, factor):
        """Side of the square crop around a target of the given (w, h): each side grown by (factor - 1) times the mean
        side, geometric mean of the two, rounded up (transt_tracker.py:121-123,143-145)."""
        margin = (factor - 1) * ((size[0] + size[1]) * 0.5)
        return math.ceil(math.sqrt((size[0] + margin) * (size[1] + margin)))

    def _bbox_clip(self, cx, cy, width, height, boundary):
        """Centre kept inside the frame, sides between 10 px and the frame's (transt_tracker.py:34-39)."""
        frame_h, frame_w = boundary[0], boundary[1]
        keep = lambda v, lo, hi: max(lo, min(v, hi))
        return keep(cx, 0, frame_w), keep(cy, 0, frame_h), keep(width, 10, frame_w), keep(height, 10, frame_h)

    def get_subwindow(self, im, pos, model_sz, original_sz, avg_chans):
        """Square crop of side original_sz around pos, padded with the channel average and resized to model_sz
        (transt_tracker.py:41-101); returned as a uint8 [1,3,model_sz,model_sz] device tensor.  The window is cut,
        padded and resized by one kernel from the frame in device memory (`im` may already be a uint8 HWC CUDA
        tensor; a numpy frame is uploaded once per call of initialize / track)."""
        return self._subwindow(im, pos, model_sz, original_sz, avg_chans, None)

    def _subwindow(self, im, pos, model_sz, original_sz, avg_chans, out):
        if isinstance(pos, float):
            pos = [pos, pos]
        sz = int(original_sz)
        c = (original_sz + 1) / 2
        context_xmin = np.floor(pos[0] - c + 0.5)
        context_ymin = np.floor(pos[1] - c + 0.5)
        pad = np.asarray(avg_chans, dtype=np.float64).astype(np.uint8)        # numpy's cast on assignment
        return ops.crop_resize_u8(self._device_frame(im), int(context_xmin), int(context_ymin), sz, int(model_sz), pad, out=out)

    def _device_frame(self, im):
        """The frame as a uint8 HWC device tensor.  A host frame is uploaded once per initialize() / track() call
        (both drop the copy when they return: a caller may refill the same numpy buffer with the next frame)."""
        if torch.is_tensor(im):
            return im if im.is_cuda else im.cuda()
        held = getattr(self, "_frame", None)
        if held is None or held[0] is not im:
            self._frame = held = (im, torch.from_numpy(np.ascontiguousarray(im)).cuda(non_blocking=True))
        return held[1]

    def _normalize(self, crop_u8):
        return ops.preprocess_u8(crop_u8, None, self.mean, self.std)

    def initialize_features(self):
        if not getattr(self, "features_initialized", False):
            self.net.initialize()
        self.features_initialized = True

    def initialize(self, image, info: dict) -> dict:
        tic = time.time()
        hanning = np.hanning(32)
        self.window = np.outer(hanning, hanning).flatten()
        self._window_dev = torch.from_numpy(self.window).cuda()
        self.initialize_features()
        inner = self.net.__dict__.get("net", self.net) if hasattr(self.net, "__dict__") else self.net
        if isinstance(inner, torch.nn.Module):
            ops.frozen_weights.forget(inner)        # the network may have been trained since the last sequence
        bbox = info["init_bbox"]
        self.center_pos = np.array([bbox[0] + bbox[2] / 2, bbox[1] + bbox[3] / 2])
        self.size = np.array([bbox[2], bbox[3]])
        s_z = self._crop_side(self.size, 2)
        self.channel_average = (image.float().mean(dim=(0, 1)).cpu().numpy() if torch.is_tensor(image)
                                else np.mean(image, axis=(0, 1)))
        z_crop = self.get_subwindow(image, self.center_pos, self.exemplar_size, s_z, self.channel_average)
        self.net.template(self._normalize(z_crop))
        self._frame = None
        self._graph, self._eager_frames = None, 0       # the captured graph reads the previous template's features
        return {"time": time.time() - tic}

    def decode(self, outputs):
        """Device-side score conversion, window penalty and arg-max for one frame.

        Returns one float64 buffer [6 + L]: (pscore, token, cx, cy, w, h) then the pscore map."""
        lg, bx = outputs["pred_logits"].contiguous(), outputs["pred_boxes"].contiguous()
        B, L, _ = lg.shape
        assert B == 1, "the online tracker handles one frame at a time"
        buf = torch.empty((6 + L,), dtype=torch.float64, device=lg.device)
        lib.call("spot_tracker_decode", ops._p(lg), ops._p(bx), ops._p(self._window_dev), float(self.window_penalty),
                 ops._p(buf), ops._p(buf[6:]), B, L, ops._st())
        return buf

    def _device_pipeline(self, x_crop):
        """uint8 search crop -> decode buffer, everything on the device."""
        return self.decode(self.net.track(self._normalize(x_crop)))

    def _run_frame(self, x_crop):
        if not self.use_graph:
            return self._device_pipeline(x_crop)
        if self._graph is None:
            if self._eager_frames < 2:
                self._eager_frames += 1
                return self._device_pipeline(x_crop)
            from . import fusion
            two = fusion.TWO_STREAMS
            try:
                # the template stream of the fusion layers forks from and rejoins the capture stream, so the replayed
                # graph keeps the two branches concurrent
                self._x_static = x_crop.clone()
                # one eager pass on the capture stream first: library handles and code objects that are created
                # per stream on first use must exist before capture begins
                cap = torch.cuda.Stream()
                cap.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(cap):
                    self._device_pipeline(self._x_static)
                torch.cuda.synchronize()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=cap):
                    self._buf_static = self._device_pipeline(self._x_static)
                self._graph = graph
            except Exception as e:                              # capture unsupported: stay eager, say so once
                import warnings
                warnings.warn("spot_amd tracker: HIP graph capture failed (%s); running eagerly" % (e,))
                torch.cuda.synchronize()
                self.use_graph = False
                return self._device_pipeline(x_crop)
            finally:
                fusion.TWO_STREAMS = two
        if x_crop is not self._x_static:                       # track() cuts the window straight into the graph's input
            self._x_static.copy_(x_crop)
        self._graph.replay()
        return self._buf_static

    def track(self, image, info: dict = None) -> dict:
        s_x = self._crop_side(self.size, 4)
        if self._graph is not None and "get_subwindow" not in self.__dict__ and type(self).get_subwindow is TransT_Tracker.get_subwindow:
            # the frame replays as a graph: the window is cut straight into the graph's input buffer (a caller that has
            # replaced get_subwindow keeps its own crop, which is then copied in)
            x_crop = self._subwindow(image, self.center_pos, self.instance_size, round(s_x), self.channel_average, self._x_static)
        else:
            x_crop = self.get_subwindow(image, self.center_pos, self.instance_size, round(s_x), self.channel_average)
        self._frame = None
        res = self._run_frame(x_crop).cpu().numpy()            # the frame's one device->host read
        pscore = res[6:]
        # the winning token's box, crop-relative in [0, 1] -> frame pixels (float32 product as in the reference)
        box = res[2:6].astype(np.float32) * s_x
        half = s_x / 2                                         # crop centre = the previous target centre
        cx, cy, width, height = self._bbox_clip(box[0] + self.center_pos[0] - half, box[1] + self.center_pos[1] - half,
                                                box[2], box[3], image.shape[:2])
        self.center_pos, self.size = np.array([cx, cy]), np.array([width, height])
        return {"target_bbox": [cx - width / 2, cy - height / 2, width, height], "best_score": pscore}
