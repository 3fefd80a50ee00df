"""Fixed GEMM workload for rocprofv3 --pmc passes: the c4 input-projection shape through lc_gemm_f32,
lc_gemm_bf16 and (for comparison of clocks / MFMA occupancy) torch.mm, 4 launches each."""
import sys
import torch
sys.path.insert(0, ".")
from lstm_ctc_amd import ops

import os
if os.environ.get("SHAPE", "NN") == "TN":          # the weight-gradient shape dKx = X^T dZ
    A = torch.randn(64000, 2048, device="cuda")
    B = torch.randn(64000, 4096, device="cuda")
    C = torch.empty(2048, 4096, device="cuda")
    for _ in range(4):
        ops.gemm(A, B, ta=True, out=C)
    for _ in range(4):
        torch.mm(A.t(), B, out=C)
else:
    A = torch.randn(64000, 2048, device="cuda")
    B = torch.randn(2048, 4096, device="cuda")
    C = torch.empty(64000, 4096, device="cuda")
    for _ in range(4):
        ops.gemm(A, B, out=C)
    for _ in range(4):
        torch.mm(A, B, out=C)
    for _ in range(4):
        ops.gemm(A, B, out=C, bf16=True)
torch.cuda.synchronize()
print("done")
