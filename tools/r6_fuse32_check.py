"""Round 6: lc_gemm_f32_nt2 at c4's full dX shape against the two products it replaces and against float64 on a row sample."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lstm_ctc_amd import ops
M, N, K = 64000, 2048, 4096
g = torch.Generator(device="cuda").manual_seed(0)
A1 = torch.randn(M, K, device="cuda", generator=g); A2 = torch.randn(M, K, device="cuda", generator=g)
B1 = torch.randn(N, K, device="cuda", generator=g) * 0.02; B2 = torch.randn(N, K, device="cuda", generator=g) * 0.02
one = ops.gemm_nt2(A1, B1, A2, B2)
two = ops.gemm(A1, B1, tb=True)
ops.gemm(A2, B2, tb=True, out=two, beta=1.0)
rows = torch.randint(0, M, (512,), device="cuda", generator=g)
ref = A1[rows].double() @ B1.double().t() + A2[rows].double() @ B2.double().t()
s = float(ref.abs().max())
print("nt2 vs two products: max |diff| %.3e (scale %.3f)" % (float((one - two).abs().max()), s))
print("vs float64 on 512 rows: nt2 %.3e, two products %.3e" % (float((one[rows].double() - ref).abs().max()), float((two[rows].double() - ref).abs().max())))
print("first / last row blocks equal-ish:", float((one[:256] - two[:256]).abs().max()), float((one[-256:] - two[-256:]).abs().max()))
