import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from spot_amd import lib, ops
from tools.microbench import timeit, r
lib.load()
for N in (256, 2048):
    for K in (32, 64, 128, 256, 512, 1024):
        M = 40960
        A, B, C = r(M, K), r(N, K), torch.empty(M, N, device="cuda")
        for tile in (64, 128):
            t = timeit(lambda: ops.gemm(A, B, C, M, N, K, tile=tile), 30)
            print("N=%4d K=%4d tile=%3d  %7.1f us  %6.1f TF   write %.2f TB/s" % (N, K, tile, t * 1e6, 2.0 * M * N * K / t / 1e12, M * N * 4 / t / 1e12))
