"""XCD-pair persistent recurrence (c4: fp32, N = 1024, both directions) against the launch train: results and us per step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lstm_ctc_amd import ops
N = 1024
B = int(os.environ.get("PB", "64"))


def run(T, persistent, bwd=False):
    os.environ["LC_LSTM_PERSISTENT"] = "1" if persistent else "0"
    g = torch.Generator().manual_seed(5)
    rows = T * B
    seq = torch.full((B,), T, dtype=torch.int32)
    seq[-5:] = T - 7
    seq = seq.cuda()
    fd = []
    for d in range(2):
        fd.append(dict(zx=(torch.randn(rows, 4 * N, generator=g) * 0.5).cuda(),
                       R=(torch.randn(N, 4 * N, generator=g) * (0.5 / N ** 0.5)).cuda(),
                       w_f=(torch.randn(N, generator=g) * 0.2).cuda(), w_i=(torch.randn(N, generator=g) * 0.2).cuda(),
                       w_o=(torch.randn(N, generator=g) * 0.2).cuda(),
                       cs=torch.zeros(rows, N, device="cuda"), hs=torch.zeros(rows, N, device="cuda"), reverse=d))
    ops.lstm_status("cuda").zero_()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ops.lstm_fwd(fd, seq, T, B, N, 1.0)
    torch.cuda.synchronize()
    tf = time.perf_counter() - t0
    sched = ops.last_lstm_schedule()
    tb = 0.0
    bd = None
    if bwd:
        bd = [dict(gates=fd[d]["zx"].clone(), RT=fd[d]["R"].t().contiguous(), w_f=fd[d]["w_f"], w_i=fd[d]["w_i"],
                   w_o=fd[d]["w_o"], cs=fd[d]["cs"], dh=(torch.randn(rows, N, generator=g) * 0.1).cuda(),
                   dpeep=torch.zeros(3, N, device="cuda"), dbias=torch.zeros(4 * N, device="cuda"), reverse=d)
              for d in range(2)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ops.lstm_bwd(bd, seq, T, B, N)
        torch.cuda.synchronize()
        tb = time.perf_counter() - t0
        sched = (sched, ops.last_lstm_schedule())
    return fd, bd, tf, tb, sched, int(ops.lstm_status("cuda").item())


bwd = os.environ.get("BWD", "0") == "1"
for T in (12, 200, 1000):
    pf, pb, tpf, tpb, sp, stp = run(T, True, bwd)
    lf, lb, tlf, tlb, sl, stl = run(T, False, bwd)
    # second timing (first call includes lazy module load)
    pf, pb, tpf, tpb, sp, stp = run(T, True, bwd)
    lf, lb, tlf, tlb, sl, stl = run(T, False, bwd)
    errs = [float((pf[d][k] - lf[d][k]).abs().max()) for d in range(2) for k in ("hs", "cs", "zx")]
    print("T=%d  status %d/%d  schedules %s | %s" % (T, stp, stl, sp, sl))
    print("   fwd: pair %.2f us/step, launch train %.2f us/step, max |diff| hs/cs/gates %s" %
          (tpf / T * 1e6, tlf / T * 1e6, ["%.2e" % e for e in errs]))
    if bwd:
        e2 = [float((pb[d]["gates"] - lb[d]["gates"]).abs().max()) for d in range(2)]
        sc = float(lb[0]["gates"].abs().max())
        print("   bwd: pair %.2f us/step, launch train %.2f us/step, max |diff| dz %s (scale %.2e)" %
              (tpb / T * 1e6, tlb / T * 1e6, ["%.2e" % e for e in e2], sc))

# ---- s_memtime anatomy of one workgroup (forward)
import ctypes
import numpy as np
from lstm_ctc_amd import _lib
T = 400
lib = _lib.load()
names = ["A: wait for operand", "A: MFMAs + B's post-processing", "A: tile store + barrier", "(gap)", "B: wait for operand",
         "B: MFMAs + A's post-processing", "B: tile store + barrier"]
idx = [(0, 1), (1, 2), (2, 3), (3, 8), (8, 9), (9, 10), (10, 11)]
if bwd:
    # the forward call's stamps are overwritten by the backward call's (same hook): anatomy of the BPTT kernel
    buf = torch.zeros(T * 16, dtype=torch.int64, device="cuda")
    fd, _, _, _, _, _ = run(T, True)
    lib.lc_debug_set_lstm_stamps(ctypes.c_void_p(buf.data_ptr()))
    g = torch.Generator().manual_seed(1)
    rows = T * B
    seqd = torch.full((B,), T, dtype=torch.int32).cuda()
    bd = [dict(gates=fd[d]["zx"].clone(), RT=fd[d]["R"].t().contiguous(), w_f=fd[d]["w_f"], w_i=fd[d]["w_i"],
               w_o=fd[d]["w_o"], cs=fd[d]["cs"], dh=(torch.randn(rows, N, generator=g) * 0.1).cuda(),
               dpeep=torch.zeros(3, N, device="cuda"), dbias=torch.zeros(4 * N, device="cuda"), reverse=d) for d in range(2)]
    ops.lstm_bwd(bd, seqd, T, B, N)
    torch.cuda.synchronize()
    lib.lc_debug_set_lstm_stamps(None)
    st = buf.cpu().numpy().reshape(T, 16)[20:-5].astype(np.float64)
    print("backward step anatomy (cycles, mean over %d steps): period %.0f" % (len(st), np.diff(st[:, 0]).mean()))
    for nm, (a, b) in zip(names, idx):
        print("   %-34s %7.0f" % (nm, (st[:, b] - st[:, a]).mean()))
buf = torch.zeros(T * 16, dtype=torch.int64, device="cuda")
lib.lc_debug_set_lstm_stamps(ctypes.c_void_p(buf.data_ptr()))
run(T, True)
lib.lc_debug_set_lstm_stamps(None)
st = buf.cpu().numpy().reshape(T, 16)[20:-5].astype(np.float64)
names = ["A: wait for state", "A: MFMAs + B's post-processing", "A: tile store + barrier", "(gap)", "B: wait for state",
         "B: MFMAs + A's post-processing", "B: tile store + barrier"]
idx = [(0, 1), (1, 2), (2, 3), (3, 8), (8, 9), (9, 10), (10, 11)]
print("forward step anatomy (cycles, mean over %d steps): period %.0f" % (len(st), np.diff(st[:, 0]).mean()))
for nm, (a, b) in zip(names, idx):
    print("   %-34s %7.0f" % (nm, (st[:, b] - st[:, a]).mean()))
