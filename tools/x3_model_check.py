"""compute_dtype = bf16x3 against fp32 at full c4 size: same parameters and batch, logits (and with X3_GRADS=1 every gradient)
compared - next to fp32 against ITSELF with another accumulation order (the 128 x 128 f32 kernel, LC_GEMM_F32_BIG=0): a
T = 1000 recurrence amplifies rounding-level differences, and this is the yardstick for how much."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from lstm_ctc_amd.nnet.model import Model
import bench

w = bench.WORKLOADS[os.environ.get("X3_WORKLOAD", "c4")]
T = int(os.environ.get("X3_T", w["T"]))
cfg = dict(w["cfg"])
if os.environ.get("X3_LAYERS"):
    cfg["num_layers"] = int(os.environ["X3_LAYERS"])
if os.environ.get("X3_KEEP"):
    cfg["dropout_rate"] = float(os.environ["X3_KEEP"])
B, D, V = w["B"], cfg["input_dim"], cfg["num_targets"]
g = torch.Generator().manual_seed(5)
x = torch.randn((T, B, D), generator=g).cuda()
sl = torch.full((B,), T, dtype=torch.int32).cuda()
dl = (torch.randn((T, B, V), generator=g) * 0.01).cuda()
res = {}
from lstm_ctc_amd import ops
for mode in ("fp32", "bf16x3", "fp32_small_tiles"):
    ops.set_option("gemm_f32_big", 0 if mode == "fp32_small_tiles" else None)
    m = Model(dict(cfg, compute_dtype=mode.split("_")[0]), "cuda", seed=9)
    logits = m.forward(x, sl, drop_seed=3).clone()
    m.backward(dl)
    res[mode] = (logits, m.ps.export_tf(grads=True))
    del m
a, b = res["fp32"], res["bf16x3"]
c = res["fp32_small_tiles"]
print("logits: max |fp32| %.4g  | bf16x3 - fp32: max %.3g rms %.3g | fp32 (128 x 128 tiles) - fp32: max %.3g rms %.3g"
      % (float(a[0].abs().max()), float((a[0] - b[0]).abs().max()), float((a[0] - b[0]).pow(2).mean().sqrt()),
         float((a[0] - c[0]).abs().max()), float((a[0] - c[0]).pow(2).mean().sqrt())))
for k in sorted(a[1]) if os.environ.get("X3_GRADS") else []:
    ga, gb = a[1][k], b[1][k]
    print("%-40s max|g| %.4g   diff %.3g  (rel %.2e)" % (k, np.abs(ga).max(), np.abs(ga - gb).max(),
                                                        np.abs(ga - gb).max() / max(np.abs(ga).max(), 1e-30)))
