"""Fixed workload for rocprofv3 --pmc passes over the 256 x 256 f32 LDS-DMA kernel: the same 64000 x 4096 x 2048 product in
its NN, NT and TN operand forms (4 launches each), to compare LDS conflicts / wait classes of the row-staged and the
k-major-staged operand paths.  Development only."""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lstm_ctc_amd import ops

M, N, K = 64000, 4096, 2048
A = torch.randn(M, K, device="cuda")
At = A.t().contiguous()
B = torch.randn(K, N, device="cuda")
Bt = B.t().contiguous()
C = torch.empty(M, N, device="cuda")
for _ in range(4):
    ops.gemm(A, B, out=C)                       # NN: A row-staged, B k-major
for _ in range(4):
    ops.gemm(A, Bt, tb=True, out=C)             # NT: both row-staged
for _ in range(4):
    ops.gemm(At, B, ta=True, out=C)             # TN: both k-major
torch.cuda.synchronize()
print("done")
