"""Error of the exact-fp32 and the bf16x3 GEMM against float64 on the step's shapes, all layouts (rms and max, relative
to the rms of the result)."""
import os
import sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from spot_amd import ops, lib
lib.load()
rng = np.random.default_rng(0)
G = lambda a: torch.from_numpy(np.ascontiguousarray(a.astype(np.float32))).cuda()
for name, ta, tb, M, N, K, sk in (("fwd ffn1", 0, 0, 8192, 2048, 256, 1), ("fwd ffn2", 0, 0, 8192, 256, 2048, 1),
                                  ("dgrad ffn1", 0, 1, 8192, 256, 2048, 1), ("dgrad ffn2", 0, 1, 8192, 2048, 256, 1),
                                  ("wgrad ffn1", 1, 1, 2048, 256, 40960, 16), ("wgrad proj", 1, 1, 256, 256, 40960, 128),
                                  ("conv1x1", 0, 1, 256, 4096, 1024, 1)):
    a64 = rng.standard_normal((M, K))
    b64 = rng.standard_normal((K, N))
    A = G(a64.T if ta else a64)
    B = G(b64 if tb else b64.T)
    ref = (A.T if ta else A).double() @ (B if tb else B.T).double()
    row = "%-11s M=%5d N=%5d K=%6d sk=%3d" % (name, M, N, K, sk)
    for prec in (0, 1):
        out = torch.zeros(M, N, device="cuda")
        ops.gemm(A, B, out, M, N, K, transA=bool(ta), transB=bool(tb), splitk=sk, accumulate=sk > 1, precision=prec)
        d = out.double() - ref
        row += " | %s rms %.2e max %.2e" % ("bf16x3" if prec else "fp32  ", float(d.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()),
                                            float(d.abs().max() / ref.pow(2).mean().sqrt()))
    print(row)
