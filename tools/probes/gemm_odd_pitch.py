"""Probe: a 1x1 convolution on an odd-sized NCHW map (H*W not a multiple of 4) as the GEMM reading the map in place
(unaligned loader) against copy into a pitch padded to 32 + aligned GEMM + narrowing copy.  Prints microseconds."""
import sys, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spot_amd import lib, ops
lib.load()
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for (B, Ci, Co, H) in [(32, 1024, 256, 31), (32, 256, 1024, 31), (32, 256, 64, 63), (32, 64, 256, 63), (1, 1024, 256, 31), (56, 512, 128, 31)]:
    HW = H * H; Np = (HW + 31) // 32 * 32
    x = torch.randn(B, Ci, HW, device='cuda'); w = torch.randn(Co, Ci, device='cuda')
    y = torch.empty(B, Co, HW, device='cuda'); xp = torch.zeros(B, Ci, Np, device='cuda'); yp = torch.empty(B, Co, Np, device='cuda')
    un = t(lambda: ops.gemm(w, x, y, Co, HW, Ci, transB=True, lda=Ci, ldb=HW, ldc=HW, batch=B, strideB=Ci * HW, strideC=Co * HW))
    al = t(lambda: ops.gemm(w, xp, yp, Co, Np, Ci, transB=True, lda=Ci, ldb=Np, ldc=Np, batch=B, strideB=Ci * Np, strideC=Co * Np))
    def full():
        xp[:, :, :HW] = x
        ops.gemm(w, xp, yp, Co, Np, Ci, transB=True, lda=Ci, ldb=Np, ldc=Np, batch=B, strideB=Ci * Np, strideC=Co * Np)
        return yp[:, :, :HW].contiguous()
    print(B, Ci, Co, H, "unaligned %.0f us  aligned-only %.0f us  pad+aligned+narrow %.0f us" % (un, al, t(full)))
