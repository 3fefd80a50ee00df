"""Fixed workload for rocprofv3 --pmc passes on the fp32-as-bf16x3 product: 4 launches each of lc_gemm_bf16x3_nt,
lc_gemm_f32 and lc_gemm_bf16_nt on the c4 zx shape (64000 x 4096 x 1024).  LC_DEV_LIB=<tag> loads a dev build."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lstm_ctc_amd import _lib as _l
if os.environ.get("LC_DEV_LIB"):
    _l.LIB_PATH = _l.LIB_PATH + "." + os.environ["LC_DEV_LIB"]
from lstm_ctc_amd import ops
M, N, K = 64000, 4096, 1024
A = torch.randn(M, K, device="cuda"); B = torch.randn(N, K, device="cuda") * 0.05
C = torch.empty(M, N, device="cuda")
A3, B3 = ops.split_bf16x3(A), ops.split_bf16x3(B)
for _ in range(4):
    ops.gemm_bf16x3_nt(A3, B3, K, out=C)
for _ in range(4):
    ops.gemm(A, B, tb=True, out=C)
an, _ = ops.cast_bf16(A, nat=True, tr=False); bn, _ = ops.cast_bf16(B, nat=True, tr=False)
for _ in range(4):
    ops.gemm_bf16_nt(an, bn, out=C, K=K)
torch.cuda.synchronize()
print("done")
