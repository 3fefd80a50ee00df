"""CTC op alone at several batch sizes: HIP-event time per call, achieved fraction of the 8 TB/s HBM roofline on the
algorithmic bytes T*B*(8V + 8S).  CTC_SHAPES="B,L;B,L;..." overrides the list."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lstm_ctc_amd import _lib as _l
if os.environ.get('LC_DEV_LIB'):      # a tools/ctc_dev_build.sh variant of the library
    _l.LIB_PATH = _l.LIB_PATH + '.' + os.environ['LC_DEV_LIB']
from lstm_ctc_amd import ops
T, V = 1000, int(os.environ.get("CTC_V", "44"))
shapes = [tuple(int(v) for v in s.split(",")) for s in os.environ.get("CTC_SHAPES", "64,100;128,100;256,100;512,100;2048,100").split(";")]
for B, L in shapes:
    g = torch.Generator().manual_seed(5)
    logits = torch.randn((T, B, V), generator=g).cuda()
    labels = torch.randint(0, V - 1, (B * L,), generator=g, dtype=torch.int32).cuda()
    offs = (torch.arange(B + 1, dtype=torch.int64) * L).to(torch.int32).cuda()
    seq = torch.full((B,), T, dtype=torch.int32).cuda()
    for _ in range(3):
        ops.ctc_loss(logits, labels, offs, seq, L)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.ctc_loss(logits, labels, offs, seq, L)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    nbytes = T * B * (8 * V + 8 * (2 * L + 1))
    print("ctc T=%d B=%d V=%d L=%d: %.1f us, algorithmic %.1f MB -> %.2f TB/s (%.1f%% of 8 TB/s)"
          % (T, B, V, L, us, nbytes / 1e6, nbytes / us / 1e6, nbytes / us / 1e6 / 8 * 100))
