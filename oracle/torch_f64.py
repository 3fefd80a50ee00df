"""float64 forward pass of the BiLSTM stack on the GPU (torch) — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Why it exists: the C oracle (`oracle.py` / `oracle_core.inc`) needs minutes of host time for a 5 x BiLSTM-1024, T = 1000
forward pass, and the question "how far is an fp32 / split-operand / bf16 implementation from float64 TRUTH at the size that
is benched" has to be answered at that size (a 1000-step recurrence at the reference's initialisation amplifies rounding noise,
so small-model evidence does not carry over).  This file restates the same TF-1.8 semantics (SURVEY.md App. A.1 / A.2,
reference call sites `nnet/bilstm.py:107-250`) with plain `torch.float64` tensor operations — no call into the product
library, no fp32 anywhere — and is itself pinned to the C oracle by `tests/test_gpu_truth.py` (agreement to 1e-12 on small
models, ragged lengths and dropout included).  Only `tests/` and `tools/` import it.

`dtype = torch.float32` runs the very same code in fp32 (rocBLAS products, torch's sigmoid / tanh): an INDEPENDENT fp32
implementation of the stack, used as the yardstick for "how far from float64 is fp32 arithmetic on this workload at all".
"""
import numpy as np
import torch


def _fmix32(h):
    m = 0xFFFFFFFF
    h = h ^ (h >> 16)
    h = (h * 0x85EBCA6B) & m
    h = h ^ (h >> 13)
    h = (h * 0xC2B2AE35) & m
    h = h ^ (h >> 16)
    return h


def dropout_mask(seed, stream, T, B, P, keep, device, dtype=torch.float64):
    """The product's counter-based Bernoulli(keep) / keep mask (`csrc/common.h: lc_dropout_scale`, restated in
    `oracle.dropout_mask`), time-major [T, B, P], as float64."""
    if keep >= 1.0:
        return None
    n = T * B * P
    idx = torch.arange(n, dtype=torch.int64, device=device)
    h = ((seed * 0x9E3779B1) & 0xFFFFFFFF) ^ ((stream * 0x85EBCA77 + 0x165667B1) & 0xFFFFFFFF)
    lo, hi = idx & 0xFFFFFFFF, idx >> 32
    v = _fmix32(lo ^ h)
    v = _fmix32(v ^ ((hi + 0x27D4EB2F) & 0xFFFFFFFF))
    u = (v >> 8).to(torch.float32) * np.float32(1.0 / 16777216.0)
    scale = float(np.float32(1.0) / np.float32(keep))
    m = torch.where(u < float(np.float32(keep)), scale, 0.0).to(dtype)
    return m.reshape(T, B, P)


def reverse_sequence(x, seq_len):
    """tf.reverse_sequence on a time-major [T, B, *] tensor: y[t, b] = x[len_b - 1 - t, b] for t < len_b, else x[t, b]."""
    T, B = x.shape[:2]
    t = torch.arange(T, device=x.device)[:, None]
    L = seq_len.to(torch.int64)[None, :]
    src = torch.where(t < L, L - 1 - t, t)                                  # [T, B]
    return torch.gather(x, 0, src[:, :, None].expand_as(x))


def lstmp_forward(x, seq_len, kernel, bias, w_f, w_i, w_o, proj, forget_bias):
    """One dynamic_rnn(LSTMCell(N, num_proj = P, use_peepholes)) over time-major x [T, B, I] -> out [T, B, P]
    (`nnet/bilstm.py:125-188`; gate order i, j, f, o; peepholes on c_{t-1} for i / f and on c_t for o; masked steps emit
    zeros and carry the state)."""
    T, B, I = x.shape
    N = bias.shape[0] // 4
    Pout = proj.shape[1] if proj is not None else N
    Kx, Kh = kernel[:I], kernel[I:]
    zx = (x.reshape(T * B, I) @ Kx).reshape(T, B, 4 * N)
    c = torch.zeros((B, N), dtype=x.dtype, device=x.device)
    m = torch.zeros((B, Pout), dtype=x.dtype, device=x.device)
    outs = []                                  # (a list + stack, not in-place writes: the pass stays differentiable)
    live = torch.arange(T, device=x.device)[:, None] < seq_len.to(torch.int64)[None, :]      # [T, B]
    for t in range(T):
        z = zx[t] + m @ Kh + bias
        zi, zj, zf, zo = z[:, :N], z[:, N:2 * N], z[:, 2 * N:3 * N], z[:, 3 * N:]
        ia = torch.sigmoid(zi + (w_i * c if w_i is not None else 0))
        fa = torch.sigmoid(zf + forget_bias + (w_f * c if w_f is not None else 0))
        cn = fa * c + ia * torch.tanh(zj)
        oa = torch.sigmoid(zo + (w_o * cn if w_o is not None else 0))
        mn = oa * torch.tanh(cn)
        if proj is not None:
            mn = mn @ proj
        a = live[t][:, None]
        outs.append(torch.where(a, mn, torch.zeros_like(mn)))
        c = torch.where(a, cn, c)
        m = torch.where(a, mn, m)
    return torch.stack(outs, 0)


def blstm_forward(params, cfg, x_tbd, seq_len, drop_seed=0, device="cuda", return_layers=False, dtype=torch.float64):
    """create_logits_blstm (`nnet/bilstm.py:25-273`, plain affine head) in float64.  `params`: TF-layout numpy arrays as
    `ParamStore.export_tf()` / the C oracle take them; x_tbd [T, B, D] (time-major, any float dtype); returns logits
    [T, B, V] float64 on `device` (and the list of layer outputs with `return_layers`)."""
    assert not cfg.get("num_experts"), "plain head only"

    def g(k):          # numpy (TF layout) -> tensor; a torch tensor is taken as it is (a leaf that requires grad: blstm_gradients)
        v = params.get(k)
        if v is None or torch.is_tensor(v):
            return v
        return torch.as_tensor(np.asarray(v, np.float64), device=device).to(dtype)

    x = torch.as_tensor(x_tbd).to(device=device, dtype=dtype)
    sl = torch.as_tensor(seq_len).to(device=device, dtype=torch.int64)
    keep = 1.0 if not cfg.get("is_training", True) else float(cfg.get("dropout_rate", 1.0))
    T, B, D = x.shape
    finput = x
    outs = []
    for i in range(cfg["num_layers"]):
        halves = []
        for d, prefix in enumerate(("fd%d/frnn%d" % (i, i), "bd%d/brnn%d" % (i, i))):
            inp = finput if d == 0 else reverse_sequence(finput, sl)
            o = lstmp_forward(inp, sl, g(prefix + "/kernel"), g(prefix + "/bias"), g(prefix + "/w_f_diag"),
                              g(prefix + "/w_i_diag"), g(prefix + "/w_o_diag"), g(prefix + "/projection/kernel"), 5.0)
            if d == 1:
                o = reverse_sequence(o, sl)
            mask = dropout_mask(drop_seed, 2 * i + d, T, B, o.shape[2], keep, device, dtype)
            halves.append(o if mask is None else o * mask)
        cat = torch.cat(halves, dim=2)
        finput = finput + cat if (i == 0 and D == cat.shape[2]) else cat
        outs.append(finput)
    logits = (finput.reshape(T * B, -1) @ g("Variable") + g("Variable_1")).reshape(T, B, -1)
    return (logits, outs) if return_layers else logits


def blstm_gradients(params, cfg, x_tbd, seq_len, dlogits_tbv, drop_seed=0, device="cuda"):
    """float64 gradients of sum(logits * dlogits) with respect to every parameter (what `tf.gradients` hands the optimizer for
    a loss whose gradient with respect to the logits is `dlogits`, `nnet/graph.py:190`), by torch autograd through
    `blstm_forward` - TRUTH for the BPTT kernels on long, non-contractive sequences at the widths that are benched.  Returns
    (logits float64 [T, B, V], {TF variable name: numpy float64 gradient})."""
    leaves = {k: torch.as_tensor(np.asarray(v, np.float64), device=device).requires_grad_(True) for k, v in params.items()
              if v is not None}
    logits = blstm_forward(leaves, cfg, x_tbd, seq_len, drop_seed=drop_seed, device=device)
    dl = torch.as_tensor(dlogits_tbv).to(device=device, dtype=torch.float64)
    (logits * dl).sum().backward()
    return logits.detach(), {k: v.grad.detach().cpu().numpy() for k, v in leaves.items() if v.grad is not None}
