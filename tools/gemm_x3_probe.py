"""lc_gemm_bf16x3_nt against lc_gemm_f32 on the c4 product shapes: time (TFLOP/s of the fp32 product it replaces), the
split passes, and the error of both against float64 (relative to the largest |C| entry).  X3_SHAPES="M,N,K;..." overrides."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from lstm_ctc_amd import _lib as _l
if os.environ.get("LC_DEV_LIB"):
    _l.LIB_PATH = _l.LIB_PATH + "." + os.environ["LC_DEV_LIB"]
from lstm_ctc_amd import ops

shapes = [(64000, 4096, 1024), (64000, 1024, 4096), (64000, 512, 1024), (64000, 4096, 40), (16000, 2048, 512), (1000, 300, 100)]
if os.environ.get("X3_SHAPES"):
    shapes = [tuple(int(v) for v in s.split(",")) for s in os.environ["X3_SHAPES"].split(";")]


def timeit(fn, n=10):
    fn(); fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e-3


g = torch.Generator(device="cuda").manual_seed(1)
for M, N, K in shapes:
    A = torch.randn(M, K, device="cuda", generator=g)
    B = torch.randn(N, K, device="cuda", generator=g) * 0.05
    if os.environ.get("X3_DATA") == "zeros":          # (a power probe: no toggling operand bits)
        A.zero_(); B.zero_()
    elif os.environ.get("X3_DATA") == "ones":
        A.fill_(1.0); B.fill_(1.0)
    out32 = torch.empty(M, N, device="cuda")
    outx3 = torch.empty(M, N, device="cuda")
    t32 = timeit(lambda: ops.gemm(A, B, tb=True, out=out32))
    A3, B3 = ops.split_bf16x3(A), ops.split_bf16x3(B)
    tx3 = timeit(lambda: ops.gemm_bf16x3_nt(A3, B3, K, out=outx3))
    tsa = timeit(lambda: ops.split_bf16x3(A))
    Bt = B.T.contiguous()
    outbl = torch.empty(M, N, device="cuda")
    tbl = timeit(lambda: torch.mm(A, Bt, out=outbl))             # hipBLASLt / rocBLAS fp32 (torch's choice)
    fl = 2.0 * M * N * K
    rows = min(M, 512)
    ref = (A[:rows].double() @ B.double().T)
    scale = float(ref.abs().max()) or 1.0
    e32 = float((out32[:rows].double() - ref).abs().max()) / scale
    ex3 = float((outx3[:rows].double() - ref).abs().max()) / scale
    r32 = float((out32[:rows].double() - ref).pow(2).mean().sqrt()) / scale
    rx3 = float((outx3[:rows].double() - ref).pow(2).mean().sqrt()) / scale
    full = float((outx3 - out32).abs().max()) / scale          # every row against the fp32 kernel
    print("M=%6d N=%5d K=%5d  torch.mm f32 %7.1f us %6.1f TF | f32 %7.1f us %6.1f TF | x3 %7.1f us %6.1f TF-eq (x%.2f) | split A %6.1f us (%.2f TB/s) | "
          "max err / max|C|: f32 %.2e  x3 %.2e   rms: f32 %.2e  x3 %.2e  | x3 - f32 over all rows %.2e"
          % (M, N, K, tbl * 1e6, fl / tbl * 1e-12, t32 * 1e6, fl / t32 * 1e-12, tx3 * 1e6, fl / tx3 * 1e-12, t32 / tx3, tsa * 1e6,
             M * K * 10 / tsa * 1e-12, e32, ex3, r32, rx3, full), flush=True)

# ---- TN form (weight gradients): A [K, M], B [K, N] K-major; X3_TN_SHAPES="M,N,K;..."
tn = [(2048, 4096, 64000), (1024, 4096, 63936), (1024, 1024, 64000), (40, 4096, 64000), (2048, 44, 64000), (300, 520, 1000)]
if os.environ.get("X3_TN_SHAPES"):
    tn = [tuple(int(v) for v in s.split(",")) for s in os.environ["X3_TN_SHAPES"].split(";")]
for M, N, K in tn:
    A = torch.randn(K, M, device="cuda", generator=g)
    B = torch.randn(K, N, device="cuda", generator=g) * 0.05
    out32 = torch.empty(M, N, device="cuda")
    outx3 = torch.empty(M, N, device="cuda")
    t32 = timeit(lambda: ops.gemm(A, B, ta=True, out=out32))
    A3, B3 = ops.split_bf16x3(A), ops.split_bf16x3(B)
    tx3 = timeit(lambda: ops.gemm_bf16x3_tn(A3, B3, M, N, out=outx3))
    fl = 2.0 * M * N * K
    cols = min(M, 256)
    ref = A[:, :cols].double().T @ B.double()
    scale = float(ref.abs().max()) or 1.0
    e32 = float((out32[:cols].double() - ref).abs().max()) / scale
    ex3 = float((outx3[:cols].double() - ref).abs().max()) / scale
    full = float((outx3 - out32).abs().max()) / scale
    print("TN M=%5d N=%5d K=%6d  f32 %7.1f us %6.1f TF | x3 %7.1f us %6.1f TF-eq (x%.2f) | max err / max|C|: f32 %.2e  x3 %.2e"
          " | x3 - f32 over all rows %.2e" % (M, N, K, t32 * 1e6, fl / t32 * 1e-12, tx3 * 1e6, fl / tx3 * 1e-12, t32 / tx3,
                                             e32, ex3, full), flush=True)
