import sys, os, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from spot_amd import lib, ops, fusion
from spot_amd.detinit import det_fill_
from spot_amd.synthetic import Settings
from spot_amd.transt import transt_resnet50
from spot_amd.tracker import NetWithBackbone, TransT_Tracker
lib.load()
def run(two, graph):
    fusion.TWO_STREAMS = two
    net = NetWithBackbone(net=det_fill_(transt_resnet50(Settings(device="cpu")), 1).cuda(), use_gpu=True)
    trk = TransT_Tracker("t", net, 0.49, 128, 256)
    trk.use_graph = graph
    rng = np.random.default_rng(0)
    frames = rng.integers(0, 255, size=(50, 256, 256, 3), dtype=np.uint8)
    for rep in range(2):
        trk.initialize(frames[0], {"init_bbox": [96.0, 96.0, 64.0, 64.0]})
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for f in frames[1:]: trk.track(f)
        torch.cuda.synchronize(); el = time.perf_counter() - t0
    print("two_streams=%s graph=%s: %.3f ms/frame" % (two, graph, el / 49 * 1e3))
for two in (True, False):
    for graph in (True, False):
        run(two, graph)
