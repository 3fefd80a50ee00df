"""Small fixed workload for rocprofv3 --pmc passes (HBM traffic per launch): the c4 input-projection GEMM and
one c4 CTC call, 3 launches each."""
import sys
import torch
sys.path.insert(0, ".")
from lstm_ctc_amd import ops

A = torch.randn(64000, 2048, device="cuda")
B = torch.randn(2048, 4096, device="cuda")
C = torch.empty(64000, 4096, device="cuda")
T, Bt, V, L = 1000, 64, 44, 100
logits = torch.randn(T, Bt, V, device="cuda")
flat = torch.randint(0, V - 1, (Bt * L,), device="cuda", dtype=torch.int32)
offs = (torch.arange(Bt + 1, device="cuda") * L).to(torch.int32)
sl = torch.full((Bt,), T, device="cuda", dtype=torch.int32)
for _ in range(3):
    ops.gemm(A, B, out=C)
    ops.ctc_loss(logits, flat, offs, sl, L)
torch.cuda.synchronize()
print("done")
