"""float64 TRUTH for the compute modes at the sizes that are benched (verdict of round 4, item 1).

For each experiment (width, layers, batch, T, keep probability) the BiLSTM stack runs in float64 on the GPU (`oracle/torch_f64.py`,
itself pinned to the C oracle) on the product's own parameters, inputs and dropout masks, and every mode's logits are compared with
THAT - never with each other:

  fp32           the fp32 MFMA kernels (256 x 256 GEMM tiles)
  fp32/128       the same arithmetic in another summation order (128 x 128 GEMM tiles): the yardstick for "two fp32 orders"
  x3/none        bf16x3 products around fp32 recurrences (round 3's mode)
  x3/fwd         ... + the split-operand forward recurrence (what a forward pass of the bf16x3 mode runs)
  bf16           plain bf16 operands (c5's arithmetic): what "reduced precision" looks like on the same scale
  torch32        an INDEPENDENT fp32 implementation (the float64 restatement's own code run in torch.float32: rocBLAS products,
                 torch's sigmoid / tanh): how far ANY fp32 arithmetic lands from float64 on this workload

plus the error profile over time and over batch rows for fp32 and x3 (a defect - a stale or torn exchange, a term error - shows as a
jump at a step or in a row group; amplification grows smoothly), and a run-to-run determinism check of the x3 mode.

    python tools/x3_truth.py                      # the default experiment list (c4 full size first)
    X3_EXPS="1024,1024,5,64,1000,0.9;320,320,3,32,1000,0.9" python tools/x3_truth.py      # N,P,layers,B,T,keep
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import torch
import torch_f64
from lstm_ctc_amd import ops
from lstm_ctc_amd.nnet import model as model_mod
from lstm_ctc_amd.nnet.model import Model

model_mod.X3_FORCE = os.environ.get("X3_FORCE", "0") == "1"

DEFAULT = ("1024,1024,5,64,1000,0.9;1024,1024,5,64,1000,1.0;1024,1024,1,64,1000,0.9;1024,1024,2,64,1000,0.9;"
           "1024,1024,5,32,1000,0.9;1024,1024,5,64,300,0.9;768,768,5,64,1000,0.9;512,512,5,32,1000,0.9;320,320,3,32,1000,0.9")


def run_mode(cfg, mode, x, sl, drop_seed):
    name, _, rec = mode.partition("/")
    ops.set_option("gemm_f32_big", 0 if rec == "128" else None)
    os.environ["LC_X3_REC"] = rec if name == "x3" else "both"
    cd = {"fp32": "fp32", "x3": "bf16x3", "bf16": "bf16"}[name]
    m = Model(dict(cfg, compute_dtype=cd), "cuda", seed=9)
    out = m.forward(x, sl, drop_seed=drop_seed).clone()
    sched = ops.last_lstm_schedule()["kind"]
    del m
    ops.set_option("gemm_f32_big", None)
    return out, sched


def stats(e):
    return "max %.3g rms %.3g" % (float(e.abs().max()), float(e.pow(2).mean().sqrt()))


for spec in os.environ.get("X3_EXPS", DEFAULT).split(";"):
    N, P, L, B, T, keep = spec.split(",")
    N, P, L, B, T, keep = int(N), int(P), int(L), int(B), int(T), float(keep)
    cfg = dict(nnet_type="blstm", input_dim=40, left_context=0, right_context=0, num_layers=L, num_neurons=N,
               num_projects=P, num_targets=44, use_peepholes=True, dropout_rate=keep)
    g = torch.Generator().manual_seed(5)
    x = torch.randn((T, B, 40), generator=g).cuda()
    sl = torch.full((B,), T, dtype=torch.int32).cuda()
    m = Model(dict(cfg), "cuda", seed=9)
    params = m.ps.export_tf()
    del m
    truth = torch_f64.blstm_forward(params, cfg, x, sl, drop_seed=3)
    print("== N %d P %d layers %d B %d T %d keep %.2f: max |logit| %.3g rms %.3g" %
          (N, P, L, B, T, keep, float(truth.abs().max()), float(truth.pow(2).mean().sqrt())), flush=True)
    outs = {}
    for mode in ("fp32", "fp32/128", "x3/none", "x3/fwd", "bf16"):
        outs[mode], sched = run_mode(cfg, mode, x, sl, 3)
        print("   %-9s vs float64: %s   (schedule %s)" % (mode, stats(outs[mode].double() - truth), sched), flush=True)
    outs["torch32"] = torch_f64.blstm_forward(params, cfg, x, sl, drop_seed=3, dtype=torch.float32)
    print("   %-9s vs float64: %s   (independent fp32: torch eager);  torch32 - fp32: %s" %
          ("torch32", stats(outs["torch32"].double() - truth), stats(outs["torch32"] - outs["fp32"])), flush=True)
    again, _ = run_mode(cfg, "x3/fwd", x, sl, 3)
    print("   x3/fwd run twice: %s;  x3/fwd - fp32: %s;  fp32/128 - fp32: %s" %
          ("bit-identical" if torch.equal(again, outs["x3/fwd"]) else "DIFFERENT " + stats(again - outs["x3/fwd"]),
           stats(outs["x3/fwd"] - outs["fp32"]), stats(outs["fp32/128"] - outs["fp32"])), flush=True)
    nb = 10
    for mode in ("fp32", "x3/none", "x3/fwd"):
        e = (outs[mode].double() - truth)
        over_t = [float(e[i * T // nb:(i + 1) * T // nb].pow(2).mean().sqrt()) for i in range(nb)]
        over_b = e.pow(2).mean(dim=(0, 2)).sqrt()
        print("   %-9s rms by tenth of T: %s" % (mode, " ".join("%.2g" % v for v in over_t)))
        print("   %-9s rms by row: min %.2g median %.2g max %.2g (row %d); rows 0-15 %.2g 16-31 %.2g 32-47 %.2g 48-63 %.2g" %
              (mode, float(over_b.min()), float(over_b.median()), float(over_b.max()), int(over_b.argmax()),
               *[float(e[:, lo:lo + 16].pow(2).mean().sqrt()) if lo < B else float("nan") for lo in (0, 16, 32, 48)]), flush=True)
    del truth, outs
    torch.cuda.empty_cache()
