#!/usr/bin/env python3
"""Kernel micro-benchmarks on one MI355X (HIP-event timed, random data).

    python tools/microbench.py gemm | mha | xcorr | norm | all
"""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from spot_amd import lib, ops  # noqa: E402


def timeit(fn, iters=20, warm=3, rounds=5):
    """Median over `rounds` timed runs of `iters` launches each (seconds per launch).  The first timed run of a process
    is preceded by ~30 ms of the same launches: the clocks take about that long to come up from idle, and a kernel
    timed before then reads 10-15 % slow."""
    global _WARM
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    if not _WARM:
        import time
        t0 = time.time()
        while time.time() - t0 < 0.03:
            fn()
            torch.cuda.synchronize()
        _WARM = True
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(rounds):
        s.record()
        for _ in range(iters):
            fn()
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / iters * 1e-3)
    ts.sort()
    return ts[len(ts) // 2]


_WARM = False


def r(*shape):
    return torch.rand(*shape, device="cuda") * 2 - 1


def bench_gemm():
    print("== gemm (fp32 MFMA, peak 157.3 TF)")
    B = 40
    Ms, Mt = B * 1024, B * 256
    cases = []
    for M in (Ms, Mt):
        cases += [("fwd  in-proj   ", M, 768, 256, False, False, dict(rowbias=True)),
                  ("fwd  out-proj  ", M, 256, 256, False, False, dict(residual=True, bias=True)),
                  ("fwd  ffn1      ", M, 2048, 256, False, False, dict(relu=True, bias=True)),
                  ("fwd  ffn2      ", M, 256, 2048, False, False, dict(residual=True, bias=True)),
                  ("dgrad in-proj  ", M, 256, 768, False, True, {}),
                  ("dgrad ffn2+mask", M, 2048, 256, False, True, dict(mask=True)),
                  ("dgrad ffn1     ", M, 256, 2048, False, True, {}),
                  ("wgrad ffn1     ", 2048, 256, M, True, True, dict(splitk=True)),
                  ("wgrad ffn2     ", 256, 2048, M, True, True, dict(splitk=True)),
                  ("wgrad proj     ", 256, 256, M, True, True, dict(splitk=True)),
                  ("wgrad in-proj  ", 768, 256, M, True, True, dict(splitk=True))]
    cases.append(("fwd  input_proj", 1024, 256, 1024, True, False, dict(batch=B)))
    # the square problem GEMM kernels are usually quoted on, in the two common layouts
    cases.append(("square nt      ", 4096, 4096, 4096, False, False, {}))
    cases.append(("square nn      ", 4096, 4096, 4096, False, True, {}))
    for name, M, N, K, ta, tb, opt in cases:
        batch = opt.get("batch", 1)
        A = r(batch, K, M) if ta else r(batch, M, K)
        Bm = r(K, N) if tb else r(N, K)
        Cc = torch.zeros(batch, M, N, device="cuda")
        kw = {}
        if opt.get("bias"):
            kw["bias"] = r(N)
        if opt.get("rowbias"):
            kw.update(rowbias=r(1024, N), rb_rows=1024, ldrb=N)
        if opt.get("relu"):
            kw["relu"] = True
        if opt.get("residual"):
            kw.update(residual=r(M, N), ldres=N)
        if opt.get("mask"):
            kw.update(mask=r(M, N), ldmask=N)
        if opt.get("splitk"):
            kw.update(splitk=ops._splitk_for(M, N, K), accumulate=True)
        if batch > 1:
            kw.update(batch=batch, strideA=M * K, strideC=M * N)
        t = timeit(lambda: ops.gemm(A, Bm, Cc, M, N, K, transA=ta, transB=tb, **kw))
        fl = 2.0 * M * N * K * batch
        extra = ""
        if os.environ.get("SPOT_MB_TILES"):
            extra = "  | tile 64: %7.1f us  tile 128: %7.1f us" % tuple(
                timeit(lambda: ops.gemm(A, Bm, Cc, M, N, K, transA=ta, transB=tb, tile=tl, **kw)) * 1e6 for tl in (64, 128))
        print("%s M=%6d N=%5d K=%6d  %8.1f us  %6.1f TF  %s%s" % (name, M, N, K, t * 1e6, fl / t / 1e12,
                                                                 "sk=%d" % kw["splitk"] if "splitk" in kw else "", extra))


def bench_gemm_bf16x3():
    """Forward nn.Linear shapes: exact fp32 products against the bf16x3 experiment, time and error against float64."""
    print("== gemm forward (A[m][k] W[n][k]): exact fp32 vs bf16x3 pieces; error = max |C - C64| / max |C64|")
    for name, M, N, K in (("in-proj ", 40960, 768, 256), ("out-proj", 40960, 256, 256), ("ffn1    ", 40960, 2048, 256),
                          ("ffn2    ", 40960, 256, 2048), ("in-proj ", 10240, 768, 256), ("out-proj", 10240, 256, 256),
                          ("ffn1    ", 10240, 2048, 256), ("ffn2    ", 10240, 256, 2048)):
        A, W, bias = r(M, K), r(N, K), r(N)
        ref = (A[:2048].double() @ W.double().T + bias.double())
        row = "%s M=%6d N=%5d K=%5d " % (name, M, N, K)
        for prec in (0, 1):
            Cc = torch.empty(M, N, device="cuda")
            t = timeit(lambda: ops.gemm(A, W, Cc, M, N, K, bias=bias, precision=prec))
            err = float((Cc[:2048].double() - ref).abs().max() / ref.abs().max())
            row += " | %s %7.1f us %6.1f TF err %.2e" % ("bf16x3" if prec else "fp32  ", t * 1e6, 2.0 * M * N * K / t / 1e12, err)
        for tile in (64, 128):
            t = timeit(lambda: ops.gemm(A, W, Cc, M, N, K, bias=bias, precision=1, tile=tile))
            row += " | tile %d: %7.1f us" % (tile, t * 1e6)
        print(row)


def bench_gemm_tiles():
    """The step's small / narrow problems under both workgroup tiles and the library's own choice."""
    print("== gemm, 64x64 vs 128x128 workgroup tile (us)")
    cases = [  # transA, transB, M, N, K, batch, splitk
        (0, 0, 256, 256, 256, 1, 1), (0, 0, 1024, 256, 256, 1, 1), (0, 0, 8192, 256, 256, 1, 1),
        (0, 0, 10240, 256, 256, 1, 1), (0, 1, 10240, 256, 256, 1, 1), (0, 0, 40960, 256, 256, 1, 1),
        (0, 0, 256, 512, 256, 1, 1), (0, 0, 1024, 512, 256, 1, 1), (0, 0, 10240, 512, 256, 1, 1),
        (0, 0, 10240, 256, 2048, 1, 1), (0, 1, 10240, 256, 2048, 1, 1), (0, 1, 10240, 2048, 256, 1, 1),
        (0, 1, 10240, 256, 768, 1, 1), (0, 0, 10240, 768, 256, 1, 1),
        (1, 1, 256, 256, 1024, 1, 4), (1, 1, 256, 256, 256, 1, 1), (1, 1, 256, 256, 10240, 1, 40),
        (1, 1, 256, 256, 40960, 1, 128), (1, 1, 512, 256, 10240, 1, 40), (1, 1, 768, 256, 10240, 1, 40),
        (1, 1, 2048, 256, 10240, 1, 16),
        (0, 0, 256, 64, 163840, 1, 256), (0, 0, 64, 256, 163840, 1, 256), (0, 0, 512, 128, 40960, 1, 128),
        (0, 0, 128, 512, 40960, 1, 128), (0, 0, 256, 64, 40960, 1, 160),
        (1, 1, 64, 4096, 256, 40, 1), (0, 1, 256, 4096, 64, 40, 1), (0, 1, 64, 4096, 256, 40, 1),
        (0, 1, 512, 1024, 128, 40, 1), (1, 1, 128, 1024, 512, 40, 1), (0, 1, 128, 1024, 512, 40, 1),
        (1, 1, 256, 256, 1024, 40, 1), (0, 1, 256, 256, 1024, 40, 1), (0, 1, 1024, 256, 256, 40, 1),
        (0, 0, 1024, 2048, 256, 1, 1), (0, 0, 1024, 256, 2048, 1, 1), (0, 0, 1024, 768, 256, 1, 1),
        (0, 0, 256, 2048, 256, 1, 1), (0, 0, 256, 256, 2048, 1, 1),
    ]
    for ta, tb, M, N, K, batch, sk in cases:
        A = r(K, M) if ta else r(M, K)
        Bm = r(batch, K, N) if tb else r(batch, N, K)
        Cc = torch.zeros(batch, M, N, device="cuda")
        kw = dict(transA=bool(ta), transB=bool(tb), batch=batch, strideB=K * N if batch > 1 else 0,
                  strideC=M * N if batch > 1 else 0, splitk=sk, accumulate=sk > 1)
        t = {tile: timeit(lambda: ops.gemm(A, Bm, Cc, M, N, K, tile=tile, **kw)) * 1e6 for tile in (64, 128, 0)}
        fl = 2.0 * M * N * K * batch
        print("%s%s M=%6d N=%5d K=%6d b=%2d sk=%3d   64: %7.1f  128: %7.1f  auto: %7.1f (%5.1f TF)%s" % (
            "t" if ta else "n", "n" if tb else "t", M, N, K, batch, sk, t[64], t[128], t[0], fl / t[0] / 1e6,
            "   <-- auto picks the slower" if t[0] > 1.08 * min(t[64], t[128]) else ""))


def bench_mha(shapes=((1024, 1024), (256, 256), (256, 1024), (1024, 256)), iters=20):
    pd = float(os.environ.get("SPOT_MB_DROP", "0"))
    bf16 = os.environ.get("SPOT_MB_BF16", "0") == "1"
    print("== mha core (%s MFMA), attention dropout %g" % ("bf16" if bf16 else "fp32", pd))
    B, H, dh = 40, 8, 32
    C = H * dh
    for Lq, Lk in shapes:
        q, k, v = r(B * Lq, C), r(B * Lk, C), r(B * Lk, C)
        out, lse = ops.mha_core_fwd(q, C, k, C, v, C, B, Lq, Lk, H, dh, None, True)
        # as in training: with dropout the forward writes the keep bits its backward reads
        keep = ops.mha_drop_bits(B, H, Lq, Lk, "cuda") if pd > 0 and os.environ.get("SPOT_MB_KEEP_BITS", "1") == "1" else None
        t = timeit(lambda: ops.mha_core_fwd(q, C, k, C, v, C, B, Lq, Lk, H, dh, None, True, out=out, drop_p=pd,
                                            drop_seed=7, bf16=bf16, drop_bits=keep), iters)
        fl = 4.0 * B * H * Lq * Lk * dh
        do = r(B * Lq, C)
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        delta = torch.empty(B, H, Lq, device="cuda")
        P = ops._p

        def bwd():
            lib.call("spot_mha_bwd_bf16" if bf16 else "spot_mha_bwd_f32", P(q), C, P(k), C, P(v), C, P(out), C, P(do), C, P(lse), None, P(delta),
                     P(dq), C, P(dk), C, P(dv), C, B, Lq, Lk, H, dh, 1.0 / math.sqrt(dh), pd, 7, P(keep), ops._st())
        tb = timeit(bwd, iters)
        print("Lq=%4d Lk=%4d  fwd %8.1f us %6.1f TF | bwd %8.1f us %6.1f TF" %
              (Lq, Lk, t * 1e6, fl / t / 1e12, tb * 1e6, 2.5 * fl / tb / 1e12))


def bench_xcorr():
    print("== groupdw (HBM-bound, 8 TB/s peak)")
    for B, hx, hz, tag in ((128, 45, 5, "cfg5 384^2"), (128, 29, 5, "ocean 255^2")):
        Cn = 256
        xs = [r(B, Cn, hx, hx), r(B, Cn, hx - 2, hx), r(B, Cn, hx, hx - 2)]
        zs = [r(B, Cn, hz, hz), r(B, Cn, hz - 2, hz), r(B, Cn, hz, hz - 2)]
        w = r(3)
        t = timeit(lambda: ops.groupdw(zs, xs, w))
        ho = hx - hz + 1
        by = 4.0 * (sum(x.numel() for x in xs) + sum(z.numel() for z in zs) + B * Cn * ho * ho)
        print("%-12s B=%d  %8.1f us  %7.1f GB/s (%.3f GB)" % (tag, B, t * 1e6, by / t / 1e9, by / 1e9))
        g = r(B, Cn, ho, ho)
        dxs, dzs = [torch.empty_like(x) for x in xs], [torch.empty_like(z) for z in zs]
        part = torch.empty(B * Cn, 3, device="cuda")
        P = ops._p
        tb = timeit(lambda: lib.call("spot_groupdw_bwd_f32", P(xs[0]), P(xs[1]), P(xs[2]), P(zs[0]), P(zs[1]), P(zs[2]),
                                     P(w), P(g), P(dxs[0]), P(dxs[1]), P(dxs[2]), P(dzs[0]), P(dzs[1]), P(dzs[2]),
                                     P(part), B * Cn, hx, hx, hz, hz, ops._st()))
        bb = 4.0 * (2 * sum(x.numel() for x in xs) + 2 * sum(z.numel() for z in zs) + B * Cn * ho * ho)
        print("%-12s backward     %8.1f us  %7.1f GB/s (%.3f GB: g, x, z in; dx, dz out)" % (tag, tb * 1e6, bb / tb / 1e9,
                                                                                          bb / 1e9))
        t = timeit(lambda: ops.xcorr_depthwise(xs[0], zs[0]))
        by = 4.0 * (xs[0].numel() + zs[0].numel() + B * Cn * ho * ho)
        print("%-12s single xcorr %8.1f us  %7.1f GB/s" % (tag, t * 1e6, by / t / 1e9))


def bench_norm():
    print("== layernorm / colsum / ema / preprocess (HBM-bound)")
    rows, C = 40960, 256
    x, g, b = r(rows, C), r(C), r(C)
    y, mean, rstd = torch.empty_like(x), torch.empty(rows, device="cuda"), torch.empty(rows, device="cuda")
    P = ops._p
    t = timeit(lambda: lib.call("spot_layernorm_fwd_f32", P(x), P(g), P(b), P(y), P(mean), P(rstd), rows, C, 1e-5, ops._st()))
    print("ln fwd  %7.1f us %7.1f GB/s" % (t * 1e6, 2 * x.numel() * 4 / t / 1e9))
    dx, dg, db = torch.empty_like(x), torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    ws = torch.empty(lib.load().spot_layernorm_bwd_ws_floats(rows, C), device="cuda")
    t = timeit(lambda: lib.call("spot_layernorm_bwd_f32", P(y), P(x), P(mean), P(rstd), P(g), P(dx), P(dg), P(db), P(ws), rows, C, 0, None, 0.0, 0, ops._st()))
    print("ln bwd  %7.1f us %7.1f GB/s" % (t * 1e6, 3 * x.numel() * 4 / t / 1e9))
    t = timeit(lambda: ops.colsum(x, out=dg))
    print("colsum  %7.1f us %7.1f GB/s" % (t * 1e6, x.numel() * 4 / t / 1e9))
    n = 23_200_000
    a, s = r(n), r(n)
    t = timeit(lambda: ops.ema_lerp_(a, s, 0.998))
    print("ema     %7.1f us %7.1f GB/s" % (t * 1e6, 12.0 * n / t / 1e9))
    img = torch.randint(0, 255, (40, 3, 256, 256), dtype=torch.uint8, device="cuda")
    t = timeit(lambda: ops.preprocess_u8(img, None, [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]))
    print("preproc %7.1f us %7.1f GB/s" % (t * 1e6, 5.0 * img.numel() / t / 1e9))


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    lib.load()
    for name, fn in (("gemm", bench_gemm), ("gemm_tiles", bench_gemm_tiles), ("mha", bench_mha), ("xcorr", bench_xcorr), ("norm", bench_norm)):
        if what in (name, "all"):
            fn()
    if what == "gemm_bf16x3":
        bench_gemm_bf16x3()
    if what == "mha1":      # one shape, few launches: for counter passes under rocprofv3 --pmc
        bench_mha(((1024, 1024),), iters=4)
