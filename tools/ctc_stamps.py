"""Phase anatomy of the meet-in-the-middle CTC from s_memtime stamps of workgroup 0 (lc_debug_set_ctc_stamps):
per pipeline iteration (16 steps), cycles of a scan wave's two ring passes, of its barrier wait, and of the frame wave's
load issue / prefix sums (part A) / class sums + softmax + stores (part B) / barrier wait.  CTC_B, CTC_L select the shape;
the number of scan waves follows from the library's geometry (frame-wave stamps live in slot NW)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes
import numpy as np
import torch
from lstm_ctc_amd import _lib as _l
if os.environ.get('LC_DEV_LIB'):      # a tools/ctc_dev_build.sh variant of the library
    _l.LIB_PATH = _l.LIB_PATH + '.' + os.environ['LC_DEV_LIB']
from lstm_ctc_amd import ops, _lib
T, B, V, L = 1000, int(os.environ.get("CTC_B", "64")), 44, int(os.environ.get("CTC_L", "100"))
logits = torch.randn(T, B, V, device="cuda")
flat = torch.randint(0, V - 1, (B * L,), device="cuda", dtype=torch.int32)
offs = (torch.arange(B + 1, device="cuda") * L).to(torch.int32)
sl = torch.full((B,), T, device="cuda", dtype=torch.int32)
for _ in range(3):
    ops.ctc_loss(logits, flat, offs, sl, L)
buf = torch.zeros(2 * 5 * 4096, dtype=torch.int64, device="cuda")
lib = _lib.load()
lib.lc_debug_set_ctc_stamps(ctypes.c_void_p(buf.data_ptr()))
ops.ctc_loss(logits, flat, offs, sl, L)
torch.cuda.synchronize()
lib.lc_debug_set_ctc_stamps(None)
st = buf.cpu().numpy().reshape(2, 5, 512, 8)
for ph in (0, 1):
    for w in range(5):
        s = st[ph, w]
        if ph == 1 and (s[:, 3] > 0).sum() > 4:          # the frame wave's slot: five stamps per iteration
            ok = (s[:, 0] > 0) & (s[:, 3] > 0) & (s[:, 4] > 0)
            idx = np.where(ok)[0][2:-1]
            print("phase 2 frame wave (slot %d): period %.0f cycles = issue loads %.0f + part A %.0f + part B %.0f + barrier wait %.0f"
                  % (w, np.diff(s[idx, 0]).mean(), (s[idx, 1] - s[idx, 0]).mean(), (s[idx, 2] - s[idx, 1]).mean(),
                     (s[idx, 3] - s[idx, 2]).mean(), (s[idx, 4] - s[idx, 3]).mean()))
            continue
        ok = (s[:, 0] > 0) & (s[:, 2] > 0)
        if ok.sum() < 4:
            continue
        idx = np.where(ok)[0][2:-1]
        body = (s[idx, 1] - s[idx, 0]).mean()
        wait = (s[idx, 2] - s[idx, 1]).mean()
        period = np.diff(s[idx, 0]).mean()
        print("phase %d scan wave %d: iteration period %.0f cycles = 2 ring passes %.0f + barrier wait %.0f  (%d iterations)"
              % (ph + 1, w, period, body, wait, len(idx)))
