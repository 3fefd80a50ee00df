"""Round 6: fixed workload for `rocprofv3 --pmc` passes over the c5 BPTT kernel (lstm_bwd_persist_bf16_kernel, B = 64,
N = 1024, T = 1000, both directions) WITHOUT and WITH the bf16 shadow of dz that a c5 step makes it write (VERDICT
round 5 item 3: name the cause of the 0.45 us per step the shadow costs).  Dispatch order: 1 warm-up + 4 launches without
the shadow, then 1 warm-up + 4 with it (tools/c5_bptt_pmc_summary.py splits the per-dispatch counters on that order).
`FWD=1`: the forward kernel with / without the hs shadow instead."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lstm_ctc_amd import ops

T, B, N = 1000, 64, 1024
FWD = os.environ.get("FWD") == "1"
torch.manual_seed(0)
rows = T * B
sl = torch.full((B,), T, device="cuda", dtype=torch.int32)
dirs = [dict(zx=torch.randn(rows, 4 * N, device="cuda") * 0.1, R=torch.randn(N, 4 * N, device="cuda") * 0.02,
             w_f=torch.zeros(N, device="cuda"), w_i=torch.zeros(N, device="cuda"), w_o=torch.zeros(N, device="cuda"),
             cs=torch.empty(rows, N, device="cuda"), hs=torch.empty(rows, N, device="cuda"), reverse=d) for d in range(2)]
ops.lstm_fwd(dirs, sl, T, B, N, 5.0, bf16=True)
for shadow in (False, True):
    if FWD:
        for dd in dirs:
            dd["hs_bf16"] = torch.empty(rows, N, dtype=torch.bfloat16, device="cuda") if shadow else None
        for _ in range(5):
            ops.lstm_fwd(dirs, sl, T, B, N, 5.0, bf16=True)
    else:
        bd = [dict(gates=dirs[d]["zx"].clone(), RT=torch.randn(4 * N, N, device="cuda") * 0.02, w_f=dirs[d]["w_f"], w_i=dirs[d]["w_i"],
                   w_o=dirs[d]["w_o"], cs=dirs[d]["cs"], dh=torch.randn(rows, N, device="cuda") * 0.01,
                   dpeep=torch.zeros(3, N, device="cuda"), dbias=torch.zeros(4 * N, device="cuda"), reverse=d) for d in range(2)]
        if shadow:
            for dd in bd:
                dd["dz_bf16"] = torch.empty(rows, 4 * N, dtype=torch.bfloat16, device="cuda")
        for _ in range(5):
            ops.lstm_bwd(bd, sl, T, B, N, bf16=True)
    torch.cuda.synchronize()
assert int(ops.lstm_status("cuda").item()) == 0
print("schedule", ops.last_lstm_schedule())
