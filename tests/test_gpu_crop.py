"""Crop + pad + resize of frames on the device against the reference's crop windows (golden g18, calls without
resize) and the CPU restatement of the resize rule.  Needs a real MI355X."""
import numpy as np
import pytest
import torch

from oracle import spot_oracle as O
from spot_amd.detinit import det_input

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def ops():
    from spot_amd import lib, ops as _ops
    lib.load()
    return _ops


def _want(im, pos, sz, S, avg):
    patch, x0, y0 = O.crop_window(im, pos, sz, avg)
    return O.resize_linear_u8(patch, S).transpose(2, 0, 1)[None], x0, y0


def test_crop_golden_windows(ops, golden):
    g = golden("g18_crop")
    im = det_input("frame", (120, 160, 3), int(g["seed"]), 0, 256, np.uint8)
    frame = torch.from_numpy(im).to(DEV)
    pad = g["avg"].astype(np.uint8)
    for i, (pos, sz) in enumerate(zip(g["sub_pos"], g["sub_sz"])):
        sz = int(sz)
        c = (sz + 1) / 2
        x0, y0 = int(np.floor(pos[0] - c + 0.5)), int(np.floor(pos[1] - c + 0.5))
        got = ops.crop_resize_u8(frame, x0, y0, sz, sz, pad)
        assert np.array_equal(got.cpu().numpy(), g["sub%d" % i]), i


@pytest.mark.parametrize("sz,S", [(64, 128), (300, 256), (256, 128), (129, 128), (37, 256), (511, 256), (128, 128)])
def test_crop_resize_matches_restated_rule(ops, sz, S):
    """Bit-exact against the CPU restatement for windows inside the frame and leaving it on each side, up- and
    down-scaling, the 2:1 case and the copy case."""
    im = det_input("frame2", (240, 320, 3), 9, 0, 256, np.uint8)
    frame = torch.from_numpy(im).to(DEV)
    avg = np.mean(im, axis=(0, 1))
    for pos in ((160.0, 120.0), (3.5, 4.25), (318.0, 237.5), (-40.0, 100.0), (160.0, 400.0)):
        want, x0, y0 = _want(im, pos, sz, S, avg)
        got = ops.crop_resize_u8(frame, x0, y0, sz, S, avg.astype(np.uint8))
        assert np.array_equal(got.cpu().numpy(), want), (pos, sz, S)


def test_crop_batch_equals_single_calls(ops):
    frames = [torch.from_numpy(det_input("f%d" % i, (100 + 20 * i, 150 - 10 * i, 3), i, 0, 256, np.uint8)).to(DEV) for i in range(5)]
    windows = [(10, 20, 64), (-15, -8, 90), (100, 60, 128), (0, 0, 33), (70, 90, 256)]
    pads = [(1, 2, 3), (200, 100, 50), (0, 0, 0), (255, 255, 255), (17, 19, 23)]
    out = ops.crop_resize_batch_u8(frames, windows, pads, 128)
    assert out.shape == (5, 3, 128, 128)
    for i in range(5):
        one = ops.crop_resize_u8(frames[i], *windows[i], 128, pads[i])
        assert torch.equal(out[i:i + 1], one), i
    assert ops.crop_resize_batch_u8([], [], [], 128).shape == (0, 3, 128, 128)


def test_tracker_subwindow_on_device(ops):
    """TransT_Tracker.get_subwindow from a host frame and from a frame already in device memory: the same crop as the
    restated reference window + resize."""
    from spot_amd.tracker import TransT_Tracker
    tr = TransT_Tracker.__new__(TransT_Tracker)
    im = det_input("frame3", (180, 240, 3), 21, 0, 256, np.uint8)
    avg = np.mean(im, axis=(0, 1))
    for pos, sz, S in (((120.0, 90.0), 100, 128), ((5.0, 170.0), 301, 256), ((200.5, 30.25), 256, 256)):
        want, _, _ = _want(im, pos, sz, S, avg)
        a = tr.get_subwindow(im, np.array(pos), S, sz, avg)
        b = tr.get_subwindow(torch.from_numpy(im).to(DEV), np.array(pos), S, sz, avg)
        assert np.array_equal(a.cpu().numpy(), want) and torch.equal(a, b)


def test_crop_window_entirely_outside_the_frame_is_all_padding(ops):
    im = det_input("frame4", (60, 80, 3), 2, 0, 256, np.uint8)
    frame = torch.from_numpy(im).to(DEV)
    for x0, y0 in ((-500, 10), (200, 10), (10, -300), (10, 90)):
        for crop, S in ((64, 64), (100, 32), (33, 128)):
            out = ops.crop_resize_u8(frame, x0, y0, crop, S, (7, 99, 201)).cpu().numpy()
            assert (out[0, 0] == 7).all() and (out[0, 1] == 99).all() and (out[0, 2] == 201).all()
