"""Phase timing of the LSTM forward step kernel via s_memtime stamps (development tool)."""
import ctypes, sys
import numpy as np, torch
sys.path.insert(0, ".")
from lstm_ctc_amd import ops, _lib
lib = _lib.load()
import os
NDIR = int(os.environ.get("NDIR", "2"))
T, B, N = 100, 64, int(os.environ.get("NN", "1024"))
BF = bool(int(os.environ.get("BF16", "0")))
rows = T * B
dirs = [dict(zx=torch.randn(rows, 4 * N, device="cuda") * 0.1, R=torch.randn(N, 4 * N, device="cuda") * 0.02,
             w_f=torch.zeros(N, device="cuda"), w_i=torch.zeros(N, device="cuda"), w_o=torch.zeros(N, device="cuda"),
             cs=torch.empty(rows, N, device="cuda"), hs=torch.empty(rows, N, device="cuda"), reverse=d) for d in range(NDIR)]
sl = torch.full((B,), T, device="cuda", dtype=torch.int32)
ops.lstm_fwd(dirs, sl, T, B, N, 5.0, bf16=BF)
buf = torch.zeros(T * 4 * 8, dtype=torch.int64, device="cuda")
lib.lc_debug_set_lstm_stamps.argtypes = [ctypes.c_void_p]
lib.lc_debug_set_lstm_stamps(ctypes.c_void_p(buf.data_ptr()))
ops.lstm_fwd(dirs, sl, T, B, N, 5.0, bf16=BF)
torch.cuda.synchronize()
lib.lc_debug_set_lstm_stamps(None)
raw = buf.cpu().numpy().reshape(T, 4, 8)[5:].astype(np.float64)
print("wave0: start->loopbegin %.0f  loop %.0f  loopend->kend %.0f" % ((raw[:,0,4]-raw[:,0,0]).mean(), (raw[:,0,5]-raw[:,0,4]).mean(), (raw[:,0,1]-raw[:,0,5]).mean()))
s = raw[:, :, :4]
d = np.diff(s, axis=2)
print("per-wave mean ticks: kloop %s  spill+barrier %s  epilogue %s" % (d[:, :, 0].mean(0), d[:, :, 1].mean(0), d[:, :, 2].mean(0)))
print("kernel start->end (wave0) mean ticks", (s[:, 0, 3] - s[:, 0, 0]).mean())
print("step-to-step start delta mean ticks", np.diff(s[:, 0, 0]).mean(), "-> if tick=10ns: us =", np.diff(s[:, 0, 0]).mean() / 100)
