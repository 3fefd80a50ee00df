"""Split-operand (bf16x3) XCD-pair recurrence next to the fp32 XCD-pair kernels: errors of both against a float64 restatement
of the step recursion (teacher-forced per step AND free-running), us per step, and the s_memtime anatomy of a workgroup.
    PN=1024 PB=64 BWD=1 python tools/x3_pair_probe.py"""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from lstm_ctc_amd import _lib
if os.environ.get('LC_DEV_LIB'):      # a tools/lstm_dev_build.sh variant of the library
    _lib.LIB_PATH = _lib.LIB_PATH + '.' + os.environ['LC_DEV_LIB']
from lstm_ctc_amd import ops

N = int(os.environ.get("PN", "1024"))
B = int(os.environ.get("PB", "64"))
BWD = os.environ.get("BWD", "0") == "1"
n_ = np.arange(N)
COLS = [torch.from_numpy((n_ // 8) * 32 + g * 8 + (n_ % 8)).cuda() for g in range(4)]      # gate-interleaved columns of gate g


def make(T, seed=5):
    g = torch.Generator().manual_seed(seed)
    rows = T * B
    seq = torch.full((B,), T, dtype=torch.int32)
    seq[-5:] = max(1, T - 7)
    fd = []
    for d in range(2):
        fd.append(dict(zx=(torch.randn(rows, 4 * N, generator=g) * 0.5).cuda(),
                       R=(torch.randn(N, 4 * N, generator=g) * (0.5 / N ** 0.5)).cuda(),
                       w_f=(torch.randn(N, generator=g) * 0.2).cuda(), w_i=(torch.randn(N, generator=g) * 0.2).cuda(),
                       w_o=(torch.randn(N, generator=g) * 0.2).cuda(),
                       cs=torch.zeros(rows, N, device="cuda"), hs=torch.zeros(rows, N, device="cuda"), reverse=d))
    dh = [(torch.randn(rows, N, generator=g) * 0.1).cuda() for _ in range(2)]
    return fd, dh, seq.cuda()


def run_fwd(T, x3, seed=5):
    fd, dh, seq = make(T, seed)
    zx0 = [d["zx"].clone() for d in fd]
    ops.lstm_status("cuda").zero_()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ops.lstm_fwd(fd, seq, T, B, N, 1.0, x3=x3)
    torch.cuda.synchronize()
    return fd, zx0, dh, seq, time.perf_counter() - t0, ops.last_lstm_schedule()["kind"], int(ops.lstm_status("cuda").item())


def run_bwd(T, x3, fd, dh, seq):
    bd = [dict(gates=fd[d]["zx"].clone(), RT=fd[d]["R"].t().contiguous(), w_f=fd[d]["w_f"], w_i=fd[d]["w_i"],
               w_o=fd[d]["w_o"], cs=fd[d]["cs"], dh=dh[d], dpeep=torch.zeros(3, N, device="cuda"),
               dbias=torch.zeros(4 * N, device="cuda"), reverse=d) for d in range(2)]
    ops.lstm_status("cuda").zero_()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ops.lstm_bwd(bd, seq, T, B, N, x3=x3)
    torch.cuda.synchronize()
    return bd, time.perf_counter() - t0, ops.last_lstm_schedule()["kind"], int(ops.lstm_status("cuda").item())


def fwd_teacher_forced_error(fd, zx0, seq, T):
    """max over steps of |h_t - f64 step from the kernel's OWN h_{t-1}, c_{t-1}| (nothing cascades)."""
    worst = 0.0
    sig = torch.sigmoid
    for d in range(2):
        R = fd[d]["R"].double()
        wf, wi, wo = (fd[d][k].double() for k in ("w_f", "w_i", "w_o"))
        hs, cs = fd[d]["hs"].view(T, B, N).double(), fd[d]["cs"].view(T, B, N).double()
        zx = zx0[d].view(T, B, 4 * N).double()
        for t in range(T):
            tp = t + 1 if d else t - 1
            hp = hs[tp] if 0 <= tp < T else torch.zeros(B, N, dtype=torch.float64, device="cuda")
            cp = cs[tp] if 0 <= tp < T else torch.zeros(B, N, dtype=torch.float64, device="cuda")
            z = zx[t] + hp @ R
            ia = sig(z[:, COLS[0]] + wi * cp)
            fa = sig(z[:, COLS[2]] + 1.0 + wf * cp)
            ja = torch.tanh(z[:, COLS[1]])
            cn = fa * cp + ia * ja
            oa = sig(z[:, COLS[3]] + wo * cn)
            act = (t < seq)[:, None]
            want = torch.where(act, oa * torch.tanh(cn), torch.zeros_like(cn))
            worst = max(worst, float((hs[t] - want).abs().max()))
    return worst


def bwd_teacher_forced_error(bd, fd, dh, seq, T):
    """max over steps of |dz_t - f64 derivative step from the kernel's own dz_{t'}| relative to max |dz|; the carried cell
    gradient is the float64 one (exact in both)."""
    worst, scale = 0.0, 0.0
    for d in range(2):
        RT = bd[d]["RT"].double()
        wf, wi, wo = (fd[d][k].double() for k in ("w_f", "w_i", "w_o"))
        g = fd[d]["zx"].view(T, B, 4 * N).double()          # activated gates (the forward's output)
        cs = fd[d]["cs"].view(T, B, N).double()
        dz = bd[d]["gates"].view(T, B, 4 * N).double()
        dc = torch.zeros(B, N, dtype=torch.float64, device="cuda")
        order = list(range(T)) if d else list(range(T - 1, -1, -1))
        for s_, t in enumerate(order):
            tprev = t + 1 if d else t - 1
            cp = cs[tprev] if 0 <= tprev < T else torch.zeros_like(dc)
            drec = dz[order[s_ - 1]] @ RT if s_ else torch.zeros_like(dc)
            dhh = dh[d].view(T, B, N)[t].double() + drec
            ia, ja, fa, oa = (g[t][:, c] for c in COLS)
            cn = cs[t]
            tc = torch.tanh(cn)
            do_pre = dhh * tc * oa * (1 - oa)
            dcn = dc + dhh * oa * (1 - tc * tc) + do_pre * wo
            di_pre = dcn * ja * ia * (1 - ia)
            dj_pre = dcn * ia * (1 - ja * ja)
            df_pre = dcn * cp * fa * (1 - fa)
            act = (t < seq)[:, None]
            dc = torch.where(act, dcn * fa + di_pre * wi + df_pre * wf, dc)
            want = torch.zeros(B, 4 * N, dtype=torch.float64, device="cuda")
            for c, v in zip(COLS, (di_pre, dj_pre, df_pre, do_pre)):
                want[:, c] = torch.where(act, v, torch.zeros_like(v))
            worst = max(worst, float((dz[t] - want).abs().max()))
            scale = max(scale, float(want.abs().max()))
    return worst, scale


for T in (12, 200, 1000):
    res = {}
    for x3 in (False, True):
        run_fwd(T, x3)                                        # lazy module load / first touch
        fd, zx0, dh, seq, tf, kind, status = run_fwd(T, x3)
        res[x3] = (fd, zx0, dh, seq)
        line = "T=%4d %-26s status %d  fwd %.2f us/step" % (T, kind, status, tf / T * 1e6)
        if T <= 200:
            line += "  teacher-forced max |h - f64 step| %.2e" % fwd_teacher_forced_error(fd, zx0, seq, T)
        if BWD:
            run_bwd(T, x3, fd, dh, seq)
            bd, tb, kb, sb = run_bwd(T, x3, fd, dh, seq)
            line += " | %-26s status %d  bwd %.2f us/step" % (kb, sb, tb / T * 1e6)
            if T <= 200:
                e, sc = bwd_teacher_forced_error(bd, fd, dh, seq, T)
                line += "  teacher-forced max |dz - f64 step| %.2e of %.2e" % (e, sc)
        print(line, flush=True)
    errs = [float((res[True][0][d][k] - res[False][0][d][k]).abs().max()) for d in range(2) for k in ("hs", "cs", "zx")]
    print("        x3 vs fp32 (free-running) max |diff| hs/cs/gates per direction: %s" % ["%.1e" % e for e in errs], flush=True)

# ---- s_memtime anatomy
lib = _lib.load()
T = 400
names = ["A: wait for operand", "A: MFMAs + B's post-processing", "A: tile store + barrier", "(gap)", "B: wait for operand",
         "B: MFMAs + A's post-processing", "B: tile store + barrier"]
idx = [(0, 1), (1, 2), (2, 3), (3, 8), (8, 9), (9, 10), (10, 11)]
REGIONS = ["XCD 0 slot 0", "XCD 0 last slot", "XCD 1 slot 0 (partner)", "XCD 3 slot 17"]


def anatomy(tag, kind, buf, x3):
    allst = buf.cpu().numpy().reshape(4, T, 16).astype(np.float64)
    st = allst[0][20:-5]
    print("%s anatomy, %s (cycles, mean over %d steps): period %.0f" % (tag, kind, len(st), np.diff(st[:, 0]).mean()))
    for nm, (a, b) in zip(names, idx):
        print("   %-34s %7.0f" % (nm, (st[:, b] - st[:, a]).mean()))
    if not x3:
        return
    # the split-operand kernels stamp four workgroups: 4 first block split, 5 / 6 around the receipt check, 7 publish
    for r, rn in enumerate(REGIONS):
        s4 = allst[r][20:-5]
        if not s4[:, 1].any():
            continue
        for base, nm in ((0, "A"), (8, "B")):
            print("   %-24s %s: wait operand %5.0f | split0 %4.0f | -> receipt check %5.0f | waited for partner %5.0f (max %5.0f) | "
                  "-> published %4.0f | -> end of stream %5.0f | store + barrier + send %5.0f" % (
                      rn, nm, (s4[:, base + 1] - s4[:, base]).mean(), (s4[:, base + 4] - s4[:, base + 1]).mean(),
                      (s4[:, base + 5] - s4[:, base + 4]).mean(), (s4[:, base + 6] - s4[:, base + 5]).mean(),
                      (s4[:, base + 6] - s4[:, base + 5]).max(), (s4[:, base + 7] - s4[:, base + 6]).mean(),
                      (s4[:, base + 2] - s4[:, base + 7]).mean(), (s4[:, base + 3] - s4[:, base + 2]).mean()))
    q = lambda v: "p10 %5.0f p50 %5.0f p90 %5.0f max %5.0f" % tuple(np.percentile(v, [10, 50, 90, 100]))
    for r, rn in enumerate(REGIONS[:2]):
        s4 = allst[r][20:-5]
        if s4[:, 1].any():
            print("   %-18s wait for operand (A): %s | waited for partner (A): %s" % (
                rn, q(s4[:, 1] - s4[:, 0]), q(s4[:, 6] - s4[:, 5])))
    # skew inside XCD 0 (one clock): when slot 0 and the last slot reach the same points
    d0, d1 = allst[0][20:-5], allst[1][20:-5]
    if d1[:, 1].any():
        for k, nm in ((1, "operand fresh"), (7, "published"), (2, "end of stream"), (3, "after barrier + send")):
            dd = d1[:, k] - d0[:, k]
            print("   last slot - slot 0 at '%s': mean %6.0f  min %6.0f  max %6.0f" % (nm, dd.mean(), dd.min(), dd.max()))


for x3 in (False, True):
    buf = torch.zeros(4 * T * 16, dtype=torch.int64, device="cuda")
    lib.lc_debug_set_lstm_stamps(ctypes.c_void_p(buf.data_ptr()))
    fd, zx0, dh, seq, _, kind, _ = run_fwd(T, x3)
    lib.lc_debug_set_lstm_stamps(None)
    anatomy("forward", kind, buf, x3)
    if BWD:
        buf.zero_()
        lib.lc_debug_set_lstm_stamps(ctypes.c_void_p(buf.data_ptr()))
        _, _, kind, _ = run_bwd(T, x3, fd, dh, seq)
        lib.lc_debug_set_lstm_stamps(None)
        anatomy("backward", kind, buf, x3)
