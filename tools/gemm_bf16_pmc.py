"""Fixed bf16 GEMM workload for rocprofv3 --pmc passes: lc_gemm_bf16_nt on the zx shape, lc_gemm_bf16_tn on the dKx shape
and torch.mm (hipBLASLt) in bf16 on both, 4 launches each."""
import sys
import torch
sys.path.insert(0, ".")
from lstm_ctc_amd import ops
A = torch.randn(64000, 2048, device="cuda"); B = torch.randn(2048, 4096, device="cuda")
an, _ = ops.cast_bf16(A, nat=True, tr=False); _, bt = ops.cast_bf16(B, nat=False, tr=True)
C = torch.empty(64000, 4096, device="cuda")
for _ in range(4):
    ops.gemm_bf16_nt(an, bt, out=C, K=2048)
Ah, Bh = A.to(torch.bfloat16), B.to(torch.bfloat16)
Ch = torch.empty(64000, 4096, device="cuda", dtype=torch.bfloat16)
for _ in range(4):
    torch.mm(Ah, Bh, out=Ch)
X = torch.randn(64000, 2048, device="cuda"); Z = torch.randn(64000, 4096, device="cuda")
xn, _ = ops.cast_bf16(X, nat=True, tr=False); zn, _ = ops.cast_bf16(Z, nat=True, tr=False)
G = torch.empty(2048, 4096, device="cuda")
for _ in range(4):
    ops.gemm_bf16_tn(xn, zn, out=G)
torch.cuda.synchronize()
print("done")
