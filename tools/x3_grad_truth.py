"""Gradients of the compute modes against float64 TRUTH (torch autograd through oracle/torch_f64.py) on full-length,
non-contractive sequences at the reference's initialisation: per configuration the relative error of ALL gradients taken together
(sqrt(sum ||g - g64||^2 / sum ||g64||^2)) and the worst single tensor, for the fp32 kernels, bf16x3 (split-operand products,
forward and BPTT recurrences) and plain bf16; the ratio bf16x3 / fp32 per tensor (min, geometric mean, max).

    X3_EXPS="N,layers,B,T;..." python tools/x3_grad_truth.py"""
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
model_mod.X3_FORCE = True

for spec in os.environ.get("X3_EXPS", "1024,2,16,300;768,1,40,200;512,2,16,300;320,3,32,300;1024,1,64,400;320,3,32,100").split(";"):
    N, layers, B, T = (int(v) for v in spec.split(","))
    cfg = dict(nnet_type="blstm", input_dim=40, left_context=0, right_context=0, num_layers=layers, num_neurons=N,
               num_projects=N, num_targets=44, use_peepholes=True, dropout_rate=0.9)
    g = torch.Generator().manual_seed(23)
    x = torch.randn((T, B, 40), generator=g)
    seq = torch.randint(T * 3 // 5, T + 1, (B,), generator=g, dtype=torch.int32)
    seq[0] = T
    dl = torch.randn((T, B, 44), generator=g) * 0.01
    for b in range(B):
        x[int(seq[b]):, b] = 0
        dl[int(seq[b]):, b] = 0
    xd, sd, dld = x.cuda(), seq.cuda(), dl.cuda()
    grads, truth, kinds = {}, None, {}
    for mode in ("fp32", "bf16x3", "bf16"):
        m = Model(dict(cfg, compute_dtype=mode), "cuda", seed=9)
        if truth is None:
            _, truth = torch_f64.blstm_gradients(m.ps.export_tf(), cfg, xd, sd, dld, drop_seed=7)
        m.forward(xd, sd, drop_seed=7)
        m.backward(dld)
        kinds[mode] = ops.last_lstm_schedule()["kind"]
        grads[mode] = m.ps.export_tf(grads=True)
        del m
        torch.cuda.empty_cache()
    den = np.sqrt(sum(np.linalg.norm(truth[k]) ** 2 for k in truth))
    line = "N %4d layers %d B %2d T %3d (%s):" % (N, layers, B, T, kinds["bf16x3"])
    rel = {}
    for mode in grads:
        rel[mode] = {k: float(np.linalg.norm(grads[mode][k] - truth[k]) / max(np.linalg.norm(truth[k]), 1e-30)) for k in truth}
        tot = np.sqrt(sum(np.linalg.norm(grads[mode][k] - truth[k]) ** 2 for k in truth)) / den
        line += "  %s all %.2e worst %.2e" % (mode, tot, max(rel[mode].values()))
    ratios = np.array([(rel["bf16x3"][k] + 1e-12) / (rel["fp32"][k] + 1e-12) for k in truth])
    print(line + "  | bf16x3 / fp32 per tensor: min %.2f geo-mean %.2f max %.2f" % (ratios.min(), np.exp(np.log(ratios).mean()), ratios.max()), flush=True)
