"""One c4-shaped CTC call repeated (for rocprofv3 --kernel-trace --stats: per-kernel split of the CTC op)."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lstm_ctc_amd import ops
T, B, V, L = 1000, int(os.environ.get("CTC_B", "64")), 44, 100
logits = torch.randn(T, B, V, device="cuda")
flat = torch.randint(0, V - 1, (B * L,), device="cuda", dtype=torch.int32)
offs = (torch.arange(B + 1, device="cuda") * L).to(torch.int32)
sl = torch.full((B,), T, device="cuda", dtype=torch.int32)
for _ in range(20):
    ops.ctc_loss(logits, flat, offs, sl, L)
torch.cuda.synchronize()
