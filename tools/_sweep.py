import sys, os
sys.path.insert(0, os.getcwd())
import torch
from spot_amd import lib, ops
from tools.microbench import timeit, r
lib.load()
for M in (32768, 65536):
  for N in (256, 2048):
    for K in (32, 128, 256, 512, 1024, 2048):
        A, B, C = r(M, K), r(N, K), torch.empty(M, N, device="cuda")
        t = timeit(lambda: ops.gemm(A, B, C, M, N, K))
        tiles = (M//128)*(N//128)
        print("M=%6d N=%5d K=%5d tiles=%5d  %8.1f us  %6.1f TF   us/ktile/round=%.2f" % (M, N, K, tiles, t*1e6, 2.0*M*N*K/t/1e12, t*1e6/(K/32)/(tiles/512)))
