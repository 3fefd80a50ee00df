"""Round 6: gemm_bf16g_kernel's persistent tile walk (LC_GEMM_BF16_PERSIST) against the one-tile-per-workgroup launch on
the c5 forward / dX shapes (NT, NN) and an unsplit TN shape: ms, TFLOP/s, and the two results compared bit for bit
(same per-tile arithmetic: the walk only changes which workgroup computes a tile).  `PROBE_BETA=1`: with beta = 1."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lstm_ctc_amd import ops


def timeit(fn, iters=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


beta = 1.0 if os.environ.get("PROBE_BETA") == "1" else 0.0
shapes = [("zx", 64000, 4096, 2048), ("dX", 64000, 2048, 4096), ("proj", 64000, 1024, 1024), ("zx_l0", 64000, 4096, 128),
          ("c3 zx", 32000, 2048, 1024)]
for name, M, N, K in shapes:
    M = M // 256 * 256
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    B = torch.randn(N, K, device="cuda").to(torch.bfloat16)
    Bk = B.t().contiguous()
    res = {}
    for form, fn in (("nt", lambda C: ops.gemm_bf16_nt(A, B, out=C, K=K, beta=beta)),
                     ("nn", lambda C: ops.gemm_bf16_nn(A, Bk, out=C, beta=beta))):
        for mode in (0, 1, 0, 1):
            ops.set_option("gemm_bf16_persist", mode)
            C = torch.zeros(M, N, device="cuda")
            fn(C)
            keep = C.clone()
            t = timeit(lambda: fn(C))
            res.setdefault((form, mode), []).append(t)
            if mode == 1:
                ops.set_option("gemm_bf16_persist", 0)
                C0 = torch.zeros(M, N, device="cuda"); fn(C0)
                assert torch.equal(C0, keep), (name, form, float((C0 - keep).abs().max()))
        ops.set_option("gemm_bf16_persist", None)
    fl = 2.0 * M * N * K
    print("%-6s M=%d N=%d K=%d: " % (name, M, N, K) + " | ".join(
        "%s %s %.3f ms %.0f TF" % (form, "persist" if mode else "one-tile", min(ts) * 1e3, fl / min(ts) / 1e12)
        for (form, mode), ts in sorted(res.items())), flush=True)
# an unsplit K-major (TN) product with many tiles: dKx-like but short K
M, N, K = 4096, 8192, 2048
A = torch.randn(K, M, device="cuda").to(torch.bfloat16); B = torch.randn(K, N, device="cuda").to(torch.bfloat16)
out = {}
for mode in (0, 1):
    ops.set_option("gemm_bf16_persist", mode)
    C = torch.zeros(M, N, device="cuda")
    ops.gemm_bf16_tn(A, B, out=C)
    out[mode] = C.clone()
    t = timeit(lambda: ops.gemm_bf16_tn(A, B, out=C))
    print("tn M=%d N=%d K=%d %s %.3f ms %.0f TF" % (M, N, K, "persist" if mode else "one-tile", t * 1e3, 2.0 * M * N * K / t / 1e12))
ops.set_option("gemm_bf16_persist", None)
assert torch.equal(out[0], out[1])
ref = (A.float().t() @ B.float())
print("tn max rel err vs torch fp32:", float((out[1] - ref).abs().max() / ref.abs().max()))
