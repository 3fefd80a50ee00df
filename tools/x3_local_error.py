"""Per-step (teacher-forced) error of the fp32 and the split-operand forward recurrences over a FULL-LENGTH trajectory at the
reference's initialisation (forget bias 5, Glorot weights, peepholes): every step's (c_t, h_t) is recomputed in float64 from the
kernel's OWN (c_{t-1}, h_{t-1}) and the same zx / R / peepholes, so nothing cascades - a defect in the step product (a stale
piece, a wrong term pair) would show as a local error far above the fp32 kernel's; equal local errors + diverging trajectories
= amplification by the dynamics.  Also prints how far the two kernels' trajectories drift apart over time, and |c| statistics
(the cell state is a 150-step integrator at forget bias 5: 1 / (1 - sigmoid(5)) = 149).

    X3_N=1024 X3_B=64 X3_T=1000 python tools/x3_local_error.py
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from lstm_ctc_amd import ops
from lstm_ctc_amd.nnet.model import Model

N = int(os.environ.get("X3_N", 1024))
B = int(os.environ.get("X3_B", 64))
T = int(os.environ.get("X3_T", 1000))
cfg = dict(nnet_type="blstm", input_dim=40, left_context=0, right_context=0, num_layers=1, num_neurons=N, num_projects=N,
           num_targets=44, use_peepholes=True, dropout_rate=1.0)
m = Model(cfg, "cuda", seed=9)
g = torch.Generator().manual_seed(5)
x = torch.randn((T * B, 40), generator=g).cuda()
sl = torch.full((B,), T, dtype=torch.int32).cuda()
cells = [m._cell(p) for p in m._prefixes(0)]
n_ = np.arange(N)
cols = [torch.from_numpy((n_ // 8) * 32 + k * 8 + (n_ % 8)).cuda() for k in range(4)]          # gate k of unit n

base = []
for d, c in enumerate(cells):
    zx = ops.gemm(x, c["Kx"], bias=c["bias"])
    R = ops.gemm(c["proj"], c["Kh"])
    base.append(dict(zx=zx, R=R, w_f=c["w_f"], w_i=c["w_i"], w_o=c["w_o"], reverse=(d == 1)))

runs = {}
for x3 in (False, True):
    dirs = [dict(zx=b_["zx"].clone(), R=b_["R"], w_f=b_["w_f"], w_i=b_["w_i"], w_o=b_["w_o"],
                 cs=torch.zeros(T * B, N, device="cuda"), hs=torch.zeros(T * B, N, device="cuda"), reverse=b_["reverse"])
            for b_ in base]
    ops.lstm_fwd(dirs, sl, T, B, N, 5.0, x3=x3)
    kind = ops.last_lstm_schedule()["kind"]
    torch.cuda.synchronize()
    runs[x3] = dirs
    print("== %s (%s), N %d B %d T %d" % ("split-operand" if x3 else "fp32", kind, N, B, T))
    for d, dd in enumerate(dirs):
        R = dd["R"].double()
        wf, wi, wo = (dd[k].double() for k in ("w_f", "w_i", "w_o"))
        hs, cs = dd["hs"].view(T, B, N), dd["cs"].view(T, B, N)
        zx = base[d]["zx"].view(T, B, 4 * N)
        ec, eh, ez_rel = [], [], []
        cmax = 0.0
        for t in range(T):
            tp = t + 1 if dd["reverse"] else t - 1
            if 0 <= tp < T:
                hp, cp = hs[tp].double(), cs[tp].double()
            else:
                hp = cp = torch.zeros(B, N, dtype=torch.float64, device="cuda")
            z = zx[t].double() + hp @ R
            ia, fa = torch.sigmoid(z[:, cols[0]] + wi * cp), torch.sigmoid(z[:, cols[2]] + 5.0 + wf * cp)
            cn = fa * cp + ia * torch.tanh(z[:, cols[1]])
            hn = torch.sigmoid(z[:, cols[3]] + wo * cn) * torch.tanh(cn)
            dc, dh = (cs[t].double() - cn), (hs[t].double() - hn)
            ec.append((float(dc.abs().max()), float(dc.pow(2).mean().sqrt()), float((dc.abs() / cn.abs().clamp_min(1.0)).max())))
            eh.append((float(dh.abs().max()), float(dh.pow(2).mean().sqrt())))
            cmax = max(cmax, float(cn.abs().max()))
        ec, eh = np.array(ec), np.array(eh)
        print("  dir %d: local |dc| max %.3g rms %.3g (rel to max(|c|,1): %.3g)   local |dh| max %.3g rms %.3g   max |c| %.3g  rms |c| %.3g"
              % (d, ec[:, 0].max(), np.sqrt((ec[:, 1] ** 2).mean()), ec[:, 2].max(), eh[:, 0].max(),
                 np.sqrt((eh[:, 1] ** 2).mean()), cmax, float(cs.pow(2).mean().sqrt())), flush=True)

for d in range(2):
    a, b = runs[False][d], runs[True][d]
    dh = (a["hs"] - b["hs"]).view(T, B, N)
    dc = (a["cs"] - b["cs"]).view(T, B, N)
    order = range(T) if not a["reverse"] else range(T - 1, -1, -1)
    steps = [s for s in order]
    nb = 10
    print("  dir %d trajectories, split-operand - fp32, by tenth of the walk: rms dh %s | rms dc %s" % (
        d, " ".join("%.2g" % float(dh[steps[i * T // nb:(i + 1) * T // nb]].pow(2).mean().sqrt()) for i in range(nb)),
        " ".join("%.2g" % float(dc[steps[i * T // nb:(i + 1) * T // nb]].pow(2).mean().sqrt()) for i in range(nb))))
