import os, sys
sys.path.insert(0, "/root/repo")
import torch
from lstm_ctc_amd import _lib as _l
if os.environ.get("LC_DEV_LIB"): _l.LIB_PATH = _l.LIB_PATH + "." + os.environ["LC_DEV_LIB"]
from lstm_ctc_amd import ops
def timeit(fn, iters=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3
for name, M, N, K in [("zx", 64000, 4096, 2048), ("dX", 64000, 2048, 4096), ("proj", 64000, 1024, 1024)]:
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16); B = torch.randn(N, K, device="cuda").to(torch.bfloat16)
    C = torch.empty(M, N, device="cuda")
    t = timeit(lambda: ops.gemm_bf16_nt(A, B, out=C, K=K))
    Bk = B.t().contiguous()
    t2 = timeit(lambda: ops.gemm_bf16_nn(A, Bk, out=C)) if hasattr(ops, "gemm_bf16_nn") and M % 256 == 0 and N % 256 == 0 else float("nan")
    print("%s nt %.3f ms %.0f TF | nn (B K-major) %.3f ms %.0f TF" % (name, t * 1e3, 2.0 * M * N * K / t / 1e12, t2 * 1e3, 2.0 * M * N * K / t2 / 1e12))
