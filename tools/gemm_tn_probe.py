"""lc_gemm_bf16_tn (K-major bf16 operands, transposing LDS reads) on the c5 weight-gradient shapes: correctness against
float64 on a small ragged case, then ms / TFLOP/s next to lc_gemm_bf16_nt on transposed shadows.  LC_DEV_LIB=<tag> loads a
development build (liblstm_ctc_hip.so.<tag>)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lstm_ctc_amd import _lib as _l
if os.environ.get("LC_DEV_LIB"):
    _l.LIB_PATH = _l.LIB_PATH + "." + os.environ["LC_DEV_LIB"]
from lstm_ctc_amd import ops


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


g = torch.Generator().manual_seed(1)
A = torch.randn((200, 512), generator=g).cuda(); B = torch.randn((200, 256), generator=g).cuda()
an, _ = ops.cast_bf16(A, nat=True, tr=False); bn, _ = ops.cast_bf16(B, nat=True, tr=False)
ref = an.double().t() @ bn.double()
err = (ops.gemm_bf16_tn(an, bn).double() - ref).abs().max().item()
print("check 512 x 256 x 200: max err %.2e %s" % (err, "OK" if err < 1e-3 else "WRONG"))
for name, M, N, K in [("dKx", 2048, 4096, 64000), ("dR", 1024, 4096, 63936), ("dproj", 1024, 1024, 64000)]:
    A = torch.randn((K, M), device="cuda"); B = torch.randn((K, N), device="cuda")
    an, at = ops.cast_bf16(A, nat=True, tr=True); bn, bt = ops.cast_bf16(B, nat=True, tr=True)
    C = torch.empty((M, N), device="cuda")
    t1 = timeit(lambda: ops.gemm_bf16_tn(an, bn, out=C))
    t2 = timeit(lambda: ops.gemm_bf16_nt(at, bt, out=C, K=K))
    fl = 2.0 * M * N * K
    print("%-5s M=%d N=%d K=%d: tn %.3f ms %.0f TF | nt (transposed shadows) %.3f ms %.0f TF" % (name, M, N, K, t1 * 1e3, fl / t1 / 1e12, t2 * 1e3, fl / t2 / 1e12))
