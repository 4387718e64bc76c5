import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from spot_amd import lib, ops
from tools.microbench import timeit, r
lib.load()
for (M, N, K) in ((256, 256, 40960), (256, 256, 10240), (768, 256, 40960), (768, 256, 10240), (512, 256, 40960), (2048, 256, 40960), (256, 2048, 40960), (256,1024,40960),(1024,256,40960),(2048,256,10240)):
    A, B = r(K, M), r(K, N)
    C = torch.zeros(M, N, device="cuda")
    row = "wgrad M=%5d N=%5d K=%6d auto sk=%3d |" % (M, N, K, ops._splitk_for(M, N, K))
    for sk in (8, 16, 20, 32, 40, 64, 80, 128, 160):
        if (K // 32) % sk: continue
        t = timeit(lambda: ops.gemm(A, B, C, M, N, K, transA=True, transB=True, splitk=sk, accumulate=True), 30)
        row += " sk%3d %6.1f us %5.1f TF |" % (sk, t * 1e6, 2.0 * M * N * K / t / 1e12)
    print(row)
