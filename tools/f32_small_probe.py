import sys; sys.path.insert(0, "/root/repo")
import torch
from lstm_ctc_amd import ops
def timeit(fn, iters=10):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
M, N, K = 64000, 2048, 4096
A = torch.randn(M, K, device="cuda"); B = torch.randn(N, K, device="cuda"); C = torch.zeros(M, N, device="cuda")
for beta in (0.0, 1.0, 0.0, 1.0):
    print("f32 NT dX beta=%g: %.3f ms" % (beta, timeit(lambda: ops.gemm(A, B, tb=True, out=C, beta=beta))))
A2 = torch.randn(M, 2 * K, device="cuda"); B2 = torch.randn(N, 2 * K, device="cuda")
print("f32 NT K=8192 (what a fused dX would cost): %.3f ms" % timeit(lambda: ops.gemm(A2, B2, tb=True, out=C)))
del A2, B2
for Kp in (40, 48, 64):
    X = torch.randn(M, Kp, device="cuda"); W = torch.randn(Kp, 4096, device="cuda"); Z = torch.zeros(M, 4096, device="cuda")
    print("zx0 K=%d: %.3f ms" % (Kp, timeit(lambda: ops.gemm(X, W, out=Z))))
    dZ = torch.randn(M, 4096, device="cuda"); G = torch.zeros(Kp, 4096, device="cuda")
    print("dKx0 K=%d: %.3f ms" % (Kp, timeit(lambda: ops.gemm(X, dZ, ta=True, out=G))))
