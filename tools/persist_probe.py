"""Anatomy of the persistent (small-model) forward recurrence: fixed cost vs per-step cost, and s_memtime phase
stamps of one workgroup (development tool)."""
import ctypes, os, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from lstm_ctc_amd import _lib
if os.environ.get('LC_DEV_LIB'):      # a tools/lstm_dev_build.sh variant of the library
    _lib.LIB_PATH = _lib.LIB_PATH + '.' + os.environ['LC_DEV_LIB']
from lstm_ctc_amd import ops
lib = _lib.load()
lib.lc_debug_set_lstm_stamps.argtypes = [ctypes.c_void_p]


def mk(T, B, N):
    rows = T * B
    dirs = [dict(zx=torch.randn(rows, 4 * N, device="cuda") * 0.1, R=torch.randn(N, 4 * N, device="cuda") * 0.02,
                 w_f=torch.zeros(N, device="cuda"), w_i=torch.zeros(N, device="cuda"), w_o=torch.zeros(N, device="cuda"),
                 cs=torch.empty(rows, N, device="cuda"), hs=torch.empty(rows, N, device="cuda"), reverse=d) for d in range(2)]
    sl = torch.full((B,), T, device="cuda", dtype=torch.int32)
    return dirs, sl


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


SHADOW = bool(int(os.environ.get("SHADOW", "0")))  # BF16=1: the kernels also write the bf16 shadows of hs / dz, as in a c5 step
BF = bool(int(os.environ.get("BF16", "0")))
X3 = bool(int(os.environ.get("X3", "0")))          # the split-operand (bf16x3) kernels instead of the fp32 ones
KW = dict(x3=True) if X3 else {}
SHAPES = ([tuple(int(v) for v in sh.split(",")) for sh in os.environ["PROBE_SHAPES"].split(";")] if os.environ.get("PROBE_SHAPES")
          else [(64, 1024), (64, 512), (32, 320)] if BF else [(32, 320), (32, 512), (64, 512), (32, 256)])     # PROBE_SHAPES="B,N;B,N"
for (B, N) in SHAPES:
    ts = {}
    for T in (200, 1000):
        dirs, sl = mk(T, B, N)
        if BF and SHADOW:
            for dd in dirs:
                dd["hs_bf16"] = torch.empty(T * B, N, dtype=torch.bfloat16, device="cuda")
        ts[T] = timeit(lambda: ops.lstm_fwd(dirs, sl, T, B, N, 5.0, bf16=BF, **KW))
        bd = [dict(gates=dirs[d]["zx"], RT=torch.randn(4 * N, N, device="cuda") * 0.02, w_f=dirs[d]["w_f"], w_i=dirs[d]["w_i"],
                   w_o=dirs[d]["w_o"], cs=dirs[d]["cs"], dh=torch.randn(T * B, N, device="cuda") * 0.01,
                   dpeep=torch.zeros(3, N, device="cuda"), reverse=d) for d in range(2)]
        if BF and SHADOW:
            for dd in bd:
                dd["dz_bf16"] = torch.empty(T * B, 4 * N, dtype=torch.bfloat16, device="cuda")
        ts[("b", T)] = timeit(lambda: ops.lstm_bwd(bd, sl, T, B, N, bf16=BF, **KW))
    per = (ts[1000] - ts[200]) / 800
    perb = (ts[("b", 1000)] - ts[("b", 200)]) / 800
    print("B=%d N=%d fwd: %.2f us/step + %.0f us fixed | bwd: %.2f us/step + %.0f us fixed" %
          (B, N, per * 1e6, (ts[200] - 200 * per) * 1e6, perb * 1e6, (ts[("b", 200)] - 200 * perb) * 1e6), flush=True)
    if BF:
        T = 200
        dirs, sl = mk(T, B, N)
        ops.lstm_fwd(dirs, sl, T, B, N, 5.0, bf16=True)
        bd = [dict(gates=dirs[d]["zx"], RT=torch.randn(4 * N, N, device="cuda") * 0.02, w_f=dirs[d]["w_f"], w_i=dirs[d]["w_i"],
                   w_o=dirs[d]["w_o"], cs=dirs[d]["cs"], dh=torch.randn(T * B, N, device="cuda") * 0.01,
                   dpeep=torch.zeros(3, N, device="cuda"), reverse=d) for d in range(2)]
        buf = torch.zeros(T * 8, dtype=torch.int64, device="cuda")
        lib.lc_debug_set_lstm_stamps(ctypes.c_void_p(buf.data_ptr()))
        ops.lstm_bwd(bd, sl, T, B, N, bf16=True)
        torch.cuda.synchronize()
        lib.lc_debug_set_lstm_stamps(None)
        r = buf.cpu().numpy().reshape(T, 8)[20:].astype(np.float64)
        print("   bf16 bwd cycles: loads + chunk-0 poll %.0f | rest of the chunks + MFMA %.0f | reduce+epilogue+publish %.0f | saved stores+sync %.0f | step %.0f" %
              ((r[:, 1] - r[:, 0]).mean(), (r[:, 2] - r[:, 1]).mean(), (r[:, 3] - r[:, 2]).mean(), (r[:, 4] - r[:, 3]).mean(), np.diff(r[:, 0]).mean()), flush=True)
        if r[:, 5].max() > 0:
            print("   looks at chunk 0 per step: mean %.2f, distribution %s" % (r[:, 5].mean(), np.bincount(r[:, 5].astype(int)).tolist()), flush=True)
        continue
    T = 200
    dirs, sl = mk(T, B, N)
    buf = torch.zeros(T * 8, dtype=torch.int64, device="cuda")
    lib.lc_debug_set_lstm_stamps(ctypes.c_void_p(buf.data_ptr()))
    ops.lstm_fwd(dirs, sl, T, B, N, 5.0, **KW)
    kind_f = ops.last_lstm_schedule()["kind"]
    torch.cuda.synchronize()
    lib.lc_debug_set_lstm_stamps(None)
    r = buf.cpu().numpy().reshape(T, 8)[20:].astype(np.float64)
    print("   %s" % kind_f)
    print("   fwd cycles: loads+wait %.0f | MFMA %.0f | reduce+epilogue+publish %.0f | saved stores+sync %.0f | step %.0f" %
          ((r[:, 1] - r[:, 0]).mean(), (r[:, 2] - r[:, 1]).mean(), (r[:, 3] - r[:, 2]).mean(), (r[:, 4] - r[:, 3]).mean(),
           np.diff(r[:, 0]).mean()), flush=True)
    bd = [dict(gates=dirs[d]["zx"], RT=torch.randn(4 * N, N, device="cuda") * 0.02, w_f=dirs[d]["w_f"], w_i=dirs[d]["w_i"],
               w_o=dirs[d]["w_o"], cs=dirs[d]["cs"], dh=torch.randn(T * B, N, device="cuda") * 0.01,
               dpeep=torch.zeros(3, N, device="cuda"), reverse=d) for d in range(2)]
    buf.zero_()
    lib.lc_debug_set_lstm_stamps(ctypes.c_void_p(buf.data_ptr()))
    ops.lstm_bwd(bd, sl, T, B, N, **KW)
    kind_b = ops.last_lstm_schedule()["kind"]
    torch.cuda.synchronize()
    lib.lc_debug_set_lstm_stamps(None)
    r = buf.cpu().numpy().reshape(T, 8)[20:].astype(np.float64)
    print("   %s" % kind_b)
    print("   bwd cycles: loads+wait %.0f | A loads+MFMA %.0f | reduce+epilogue+publish %.0f | arrive %.0f | step %.0f" %
          ((r[:, 1] - r[:, 0]).mean(), (r[:, 2] - r[:, 1]).mean(), (r[:, 3] - r[:, 2]).mean(), (r[:, 4] - r[:, 3]).mean(),
           np.diff(r[:, 0]).mean()), flush=True)
