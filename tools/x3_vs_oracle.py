"""Long-sequence accuracy of the three compute modes against the float64 oracle: a T = 1000 BiLSTM amplifies rounding-level
differences of the products feeding the recurrence, so fp32 and bf16x3 are each compared with float64 TRUTH on the same
parameters and batch (not with each other)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import torch
import oracle
from lstm_ctc_amd.nnet import model as model_mod
from lstm_ctc_amd.nnet.model import Model
model_mod.X3_FORCE = True          # sizes below the mode's own threshold: every eligible product on the bf16x3 kernels

T, B = int(os.environ.get("X3_T", 1000)), int(os.environ.get("X3_B", 32))
cfg = dict(nnet_type="blstm", input_dim=40, left_context=0, right_context=0, num_layers=int(os.environ.get("X3_LAYERS", 2)),
           num_neurons=int(os.environ.get("X3_N", 256)), num_projects=int(os.environ.get("X3_P", 128)), num_targets=44,
           use_peepholes=True, dropout_rate=0.9)
rng = np.random.default_rng(3)
x = rng.normal(size=(B, T, 40)).astype(np.float32)
seq = np.full((B,), T, np.int32)
xt = torch.from_numpy(np.ascontiguousarray(x.transpose(1, 0, 2))).cuda()
sl = torch.from_numpy(seq).cuda()
ref = None
for mode in ("fp32", "bf16x3", "bf16"):
    m = Model(dict(cfg, compute_dtype=mode), "cuda", seed=9)
    if ref is None:
        p64 = {k: v.astype(np.float64) for k, v in m.ps.export_tf().items()}
        ref, _ = oracle.forward(p64, cfg, x.astype(np.float64), seq, drop_seed=7)
    got = m.forward(xt, sl, drop_seed=7).cpu().numpy().transpose(1, 0, 2)
    e = np.abs(got - ref)
    print("%-7s logits vs float64: max %.3g  rms %.3g   (max |logit| %.3g)" % (mode, e.max(), np.sqrt((e ** 2).mean()),
                                                                             np.abs(ref).max()), flush=True)
