"""Kernel-level timing probe (run on the GPU box): GEMM TFLOP/s at the c4 shapes, CTC at c2/c4, LSTM
fwd/bwd recurrence per step.  Prints one line per measurement; used to steer optimisation."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import os  # noqa: E402
from lstm_ctc_amd import ops, _lib  # noqa: E402
if os.environ.get("LC_LIB"):          # development only: time an experimental build of the library
    _lib.LIB_PATH = os.path.abspath(os.environ["LC_LIB"])


def timeit(fn, warmup=2, iters=5):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True)
    e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def gemm_probe(bf16=False):
    for (name, ta, tb, M, N, K) in [
        ("NN zx  ", 0, 0, 64000, 4096, 2048),
        ("NT dX  ", 0, 1, 64000, 2048, 4096),
        ("TN dKx ", 1, 0, 2048, 4096, 64000),
        ("NN proj", 0, 0, 64000, 1024, 1024),
        ("TN dR  ", 1, 0, 1024, 4096, 63936),
        ("NN head", 0, 0, 64000, 44, 2048),
        ("NN c2  ", 0, 0, 32000, 1280, 640),
    ]:
        A = torch.randn((K, M) if ta else (M, K), device="cuda")
        B = torch.randn((N, K) if tb else (K, N), device="cuda")
        C = torch.empty((M, N), device="cuda")
        t = timeit(lambda: ops.gemm(A, B, ta=bool(ta), tb=bool(tb), out=C))
        At, Bt = (A.t() if ta else A), (B.t() if tb else B)
        t2 = timeit(lambda: torch.mm(At, Bt, out=C))
        fl = 2.0 * M * N * K
        print("gemm %s M=%d N=%d K=%d: mine %.3f ms %.1f TF | torch.mm %.3f ms %.1f TF" %
              (name, M, N, K, t * 1e3, fl / t / 1e12, t2 * 1e3, fl / t2 / 1e12), flush=True)
        if bf16:
            t3 = timeit(lambda: ops.gemm(A, B, ta=bool(ta), tb=bool(tb), out=C, bf16=True))
            Ah, Bh = At.to(torch.bfloat16), Bt.to(torch.bfloat16)
            Ch = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
            t4 = timeit(lambda: torch.mm(Ah, Bh, out=Ch))
            print("     bf16 operands: mine (fp32 in HBM) %.3f ms %.1f TF | torch.mm bf16 %.3f ms %.1f TF" %
                  (t3 * 1e3, fl / t3 / 1e12, t4 * 1e3, fl / t4 / 1e12), flush=True)
            # shadow path: casts (natural or transposed as the NT form needs) timed separately from the product
            tc = timeit(lambda: (ops.cast_bf16(A, nat=not ta, tr=bool(ta)), ops.cast_bf16(B, nat=bool(tb), tr=not tb)))
            an, at_ = ops.cast_bf16(A, nat=not ta, tr=bool(ta))
            bn, bt_ = ops.cast_bf16(B, nat=bool(tb), tr=not tb)
            As, Bs = (at_ if ta else an), (bn if tb else bt_)
            if K % 8 == 0:
                t5 = timeit(lambda: ops.gemm_bf16_nt(As, Bs, out=C, K=K))
                print("     bf16 shadows:  casts %.3f ms, lc_gemm_bf16_nt %.3f ms %.1f TF" % (tc * 1e3, t5 * 1e3, fl / t5 / 1e12),
                      flush=True)
            if ta and not tb and M % 256 == 0 and N % 256 == 0:      # K-major kernel on the natural shadows: no casts at all
                an2, _ = ops.cast_bf16(A, nat=True, tr=False)
                bn2, _ = ops.cast_bf16(B, nat=True, tr=False)
                t6 = timeit(lambda: ops.gemm_bf16_tn(an2, bn2, out=C))
                print("     bf16 natural shadows: lc_gemm_bf16_tn %.3f ms %.1f TF" % (t6 * 1e3, fl / t6 / 1e12), flush=True)


def ctc_probe():
    for (T, B, V, L) in [(1000, 32, 72, 100), (1000, 64, 44, 100), (1000, 64, 44, 63), (1000, 64, 44, 31), (1000, 512, 44, 100),
                         (1000, 2048, 44, 100)]:
        logits = torch.randn(T, B, V, device="cuda")
        flat = torch.randint(0, V - 1, (B * L,), device="cuda", dtype=torch.int32)
        offs = (torch.arange(B + 1, device="cuda") * L).to(torch.int32)
        sl = torch.full((B,), T, device="cuda", dtype=torch.int32)
        t = timeit(lambda: ops.ctc_loss(logits, flat, offs, sl, L), iters=10)
        S = 2 * L + 1
        byts = T * B * (8 * V + 8 * S)
        print("ctc T=%d B=%d V=%d L=%d: %.1f us, algorithmic %.1f MB -> %.2f TB/s (%.1f%% of 8 TB/s)" %
              (T, B, V, L, t * 1e6, byts / 1e6, byts / t / 1e12, byts / t / 8e12 * 100), flush=True)


def lstm_probe():
    for (T, B, N) in [(400, 64, 1024), (400, 64, 512), (400, 32, 320), (400, 32, 512)]:
        rows = T * B
        dirs = []
        for d in range(2):
            dirs.append(dict(zx=torch.randn(rows, 4 * N, device="cuda") * 0.1, R=torch.randn(N, 4 * N, device="cuda") * 0.02,
                             w_f=torch.zeros(N, device="cuda"), w_i=torch.zeros(N, device="cuda"),
                             w_o=torch.zeros(N, device="cuda"), cs=torch.empty(rows, N, device="cuda"),
                             hs=torch.empty(rows, N, device="cuda"), reverse=d))
        sl = torch.full((B,), T, device="cuda", dtype=torch.int32)
        t = timeit(lambda: ops.lstm_fwd(dirs, sl, T, B, N, 5.0), warmup=1, iters=3)
        fl = 2 * 2.0 * B * N * 4 * N * T
        print("lstm_fwd bidir T=%d B=%d N=%d: %.2f ms = %.2f us/step, %.1f TF" % (T, B, N, t * 1e3, t / T * 1e6, fl / t / 1e12), flush=True)
        if N % 32 == 0:
            t = timeit(lambda: ops.lstm_fwd(dirs, sl, T, B, N, 5.0, bf16=True), warmup=1, iters=3)
            print("lstm_fwd bf16  T=%d B=%d N=%d: %.2f ms = %.2f us/step" % (T, B, N, t * 1e3, t / T * 1e6), flush=True)
        bd = [dict(gates=dirs[d]["zx"], RT=torch.randn(4 * N, N, device="cuda") * 0.02, w_f=dirs[d]["w_f"], w_i=dirs[d]["w_i"],
                   w_o=dirs[d]["w_o"], cs=dirs[d]["cs"], dh=torch.randn(rows, N, device="cuda") * 0.01,
                   dpeep=torch.zeros(3, N, device="cuda"), reverse=d) for d in range(2)]
        t = timeit(lambda: ops.lstm_bwd(bd, sl, T, B, N), warmup=1, iters=3)
        print("lstm_bwd bidir T=%d B=%d N=%d: %.2f ms = %.2f us/step, %.1f TF" % (T, B, N, t * 1e3, t / T * 1e6, fl / t / 1e12), flush=True)


if __name__ == "__main__":
    which = sys.argv[1:] or ["gemm", "ctc", "lstm"]
    print(torch.cuda.get_device_name(0), flush=True)
    if "stream" in which:          # run everything on a non-default stream (graph capture needs one)
        torch.cuda.set_stream(torch.cuda.Stream())
    if "gemm" in which:
        gemm_probe()
    if "gemm_bf16" in which:
        gemm_probe(True)
    if "ctc" in which:
        ctc_probe()
    if "lstm" in which:
        lstm_probe()
