"""float64 emulation of the product's `compute_dtype = bf16` path (BASELINE.json configs[4], "c5").

TEST INFRASTRUCTURE, like everything under oracle/: only tests/ may import it.  PARITY UNPINNED BY THE REFERENCE:
mobvoi/lstm_ctc has no bf16 arithmetic at all, so the semantics are the product's own and
this file restates them independently in numpy:

  * the algebra is the folded form the product evaluates of the reference's BiLSTM-P stack
    (nnet/bilstm.py:125-250; cell semantics SURVEY.md App. A.1/A.2): x.Kx hoisted, R = proj.Kh recurs on the
    pre-projection output m', m = m'.proj batched afterwards;
  * every product with an ACTIVATION operand rounds BOTH operands to bf16 (round-to-nearest-even, oracle.bf16_round)
    and accumulates exactly (float64 here, float32 on the MFMA): x.Kx, m'.R, m'.proj, the head, and in the backward
    dY.proj^T, dz.R^T, X^T.dz, M'^T.dz, M'^T.dY, dz.Kx^T, dl.W^T, H^T.dl;
  * weight-only products (R = proj.Kh, dKh = proj^T.dR, dproj += dR.Kh^T), gate math, cell state, bias / peephole
    gradients stay unrounded.

Supported: nnet_type blstm, num_projects set, peepholes on or off, affine head, no dropout, no first-layer residual
(what config c5 needs).  Everything is in TF layouts ([i|j|f|o] gate blocks, TF variable names).
"""
import numpy as np

from . import oracle as orc


def _bf(a):
    return orc.bf16_round(np.ascontiguousarray(a, dtype=np.float32)).astype(np.float64)


def _mm(a, b):
    """bf16(a) @ bf16(b), exact accumulation."""
    return _bf(a) @ _bf(b)


def _sig(x):
    return 1.0 / (1.0 + np.exp(-x))


def _cell(params, prefix, I):
    k = params[prefix + "/kernel"].astype(np.float64)
    return dict(Kx=k[:I], Kh=k[I:], bias=params[prefix + "/bias"].astype(np.float64),
                w_f=params.get(prefix + "/w_f_diag"), w_i=params.get(prefix + "/w_i_diag"),
                w_o=params.get(prefix + "/w_o_diag"), proj=params[prefix + "/projection/kernel"].astype(np.float64),
                prefix=prefix)


def _prefixes(i):
    return ["fd%d/frnn%d" % (i, i), "bd%d/brnn%d" % (i, i)]


def forward(params, cfg, x, seq_len, forget_bias=5.0):
    """x [B,T,D] -> (logits [B,T,V], saved)."""
    assert cfg.get("nnet_type", "blstm") == "blstm" and cfg.get("num_projects") and not cfg.get("num_experts")
    B, T, D = x.shape
    N, P = cfg["num_neurons"], cfg["num_projects"]
    assert D != 2 * P, "first-layer residual is not emulated"
    seq_len = np.asarray(seq_len)
    inp = np.ascontiguousarray(np.asarray(x, np.float64).transpose(1, 0, 2)).reshape(T * B, D)    # time-major rows
    layers = []
    for i in range(cfg["num_layers"]):
        I = inp.shape[1]
        Y = np.zeros((T * B, 2 * P))
        dirs = []
        for d, pre in enumerate(_prefixes(i)):
            c = _cell(params, pre, I)
            zx = (_mm(inp, c["Kx"]) + c["bias"]).reshape(T, B, 4 * N)
            # the fold itself is a float32 GEMM in the product (not a bf16 product): keep its float32 value
            R = (c["proj"] @ c["Kh"]).astype(np.float32).astype(np.float64)
            Rb = _bf(R)
            wf, wi, wo = (np.zeros(N) if w is None else w.astype(np.float64) for w in (c["w_f"], c["w_i"], c["w_o"]))
            gates = np.zeros((T, B, 4 * N)); cs = np.zeros((T, B, N)); hs = np.zeros((T, B, N))
            hq = np.zeros((B, N)); cp = np.zeros((B, N))
            for t in (range(T - 1, -1, -1) if d == 1 else range(T)):
                z = zx[t] + hq @ Rb
                zi, zj, zf, zo = z[:, :N], z[:, N:2 * N], z[:, 2 * N:3 * N], z[:, 3 * N:]
                ia = _sig(zi + wi * cp); fa = _sig(zf + forget_bias + wf * cp); ja = np.tanh(zj)
                cn = fa * cp + ia * ja
                oa = _sig(zo + wo * cn)
                h = oa * np.tanh(cn)
                act = (t < seq_len)[:, None]
                gates[t] = np.where(act, np.concatenate([ia, ja, fa, oa], axis=1), 0.0)
                cs[t] = np.where(act, cn, 0.0); hs[t] = np.where(act, h, 0.0)
                cp = cs[t]
                hq = _bf(hs[t])
            hs2 = hs.reshape(T * B, N)
            Y[:, d * P:(d + 1) * P] = _mm(hs2, c["proj"])
            dirs.append(dict(cell=c, gates=gates, cs=cs, hs=hs2, R=R, reverse=(d == 1)))
        layers.append(dict(inp=inp, dirs=dirs))
        inp = Y
    W, b = params["Variable"].astype(np.float64), params["Variable_1"].astype(np.float64)
    logits = _mm(inp, W) + b
    V = W.shape[1]
    saved = dict(layers=layers, top=inp, T=T, B=B, seq_len=seq_len)
    return logits.reshape(T, B, V).transpose(1, 0, 2), saved


def backward(params, cfg, saved, dlogits):
    """dlogits [B,T,V] -> dict of gradients under the TF variable names, TF layouts."""
    T, B, seq_len = saved["T"], saved["B"], saved["seq_len"]
    N, P = cfg["num_neurons"], cfg["num_projects"]
    rows = T * B
    dl = np.ascontiguousarray(np.asarray(dlogits, np.float64).transpose(1, 0, 2)).reshape(rows, -1)
    W = params["Variable"].astype(np.float64)
    g = {"Variable": _mm(saved["top"].T, dl), "Variable_1": dl.sum(0)}
    dY = _mm(dl, W.T)
    for i in reversed(range(cfg["num_layers"])):
        L = saved["layers"][i]
        inp = L["inp"]
        dinp = np.zeros_like(inp) if i > 0 else None
        for d, dd in enumerate(L["dirs"]):
            c = dd["cell"]
            pre = c["prefix"]
            half = dY[:, d * P:(d + 1) * P]
            dh_all = _mm(half, c["proj"].T).reshape(T, B, N)
            RTb = _bf(dd["R"].T)
            wf, wi, wo = (np.zeros(N) if w is None else w.astype(np.float64) for w in (c["w_f"], c["w_i"], c["w_o"]))
            gates, cs = dd["gates"], dd["cs"]
            dz = np.zeros((T, B, 4 * N)); dzq = np.zeros((B, 4 * N)); dc = np.zeros((B, N))
            for t in (range(T) if dd["reverse"] else range(T - 1, -1, -1)):
                tprev = t + 1 if dd["reverse"] else t - 1
                cp = cs[tprev] if 0 <= tprev < T else np.zeros((B, N))
                dh = dh_all[t] + dzq @ RTb
                ia, ja, fa, oa = gates[t][:, :N], gates[t][:, N:2 * N], gates[t][:, 2 * N:3 * N], gates[t][:, 3 * N:]
                cn = cs[t]; tc = np.tanh(cn)
                do_pre = dh * tc * oa * (1 - oa)
                dcn = dc + dh * oa * (1 - tc * tc) + do_pre * wo
                di_pre = dcn * ja * ia * (1 - ia); dj_pre = dcn * ia * (1 - ja * ja); df_pre = dcn * cp * fa * (1 - fa)
                act = (t < seq_len)[:, None]
                dc = np.where(act, dcn * fa + di_pre * wi + df_pre * wf, dc)
                dz[t] = np.where(act, np.concatenate([di_pre, dj_pre, df_pre, do_pre], axis=1), 0.0)
                dzq = _bf(dz[t])
            dz2 = dz.reshape(rows, 4 * N)
            g[pre + "/bias"] = dz2.sum(0)
            if c["w_f"] is not None:
                cprev = np.zeros((T, B, N))
                if dd["reverse"]:
                    cprev[:-1] = cs[1:]
                else:
                    cprev[1:] = cs[:-1]
                g[pre + "/w_i_diag"] = (dz[:, :, :N] * cprev).sum((0, 1))
                g[pre + "/w_f_diag"] = (dz[:, :, 2 * N:3 * N] * cprev).sum((0, 1))
                g[pre + "/w_o_diag"] = (dz[:, :, 3 * N:] * cs).sum((0, 1))
            dKx = _mm(inp.T, dz2)
            hs = dd["hs"]
            if dd["reverse"]:
                hprev, dzs = hs[B:], dz2[:rows - B]
            else:
                hprev, dzs = hs[:rows - B], dz2[B:]
            dR = _mm(hprev.T, dzs) if T > 1 else np.zeros((N, 4 * N))
            dR = dR.astype(np.float32).astype(np.float64)                  # a float32 tensor in the product
            g[pre + "/kernel"] = np.concatenate([dKx, c["proj"].T @ dR], axis=0)
            g[pre + "/projection/kernel"] = _mm(hs.T, half) + dR @ c["Kh"].T
            if dinp is not None:
                dinp += _mm(dz2, c["Kx"].T)
        dY = dinp
    return g
