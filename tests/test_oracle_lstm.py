"""Pins the oracle's LSTMP / BiLSTM / MoE restatement (CPU only).

Anchors: torch.nn.LSTM(proj_size) for the peephole-free subset (gate order / bias remap),
a torch-autograd restatement of SURVEY.md Appendix A.1/A.2 for peepholes + masking (fwd and
bwd), and fp64 finite differences through the whole BiLSTM(+MoE) + CTC graph.
"""
import numpy as np
import pytest
import torch


def _torch_lstmp(x, seq_len, kernel, bias, wf, wi, wo, proj, fb):
    """Literal per-step restatement of LSTMCell + dynamic_rnn masking (App. A.1/A.2)."""
    B, T, I = x.shape
    N = bias.shape[0] // 4
    Pout = proj.shape[1] if proj is not None else N
    c = torch.zeros(B, N, dtype=x.dtype)
    m = torch.zeros(B, Pout, dtype=x.dtype)
    outs = []
    for t in range(T):
        z = torch.cat([x[:, t], m], 1) @ kernel + bias
        i, j, f, o = z.split(N, dim=1)
        cn = torch.sigmoid(f + fb + (wf * c if wf is not None else 0)) * c + \
            torch.sigmoid(i + (wi * c if wi is not None else 0)) * torch.tanh(j)
        mn = torch.sigmoid(o + (wo * cn if wo is not None else 0)) * torch.tanh(cn)
        if proj is not None:
            mn = mn @ proj
        act = (t < seq_len).to(x.dtype)[:, None]
        outs.append(mn * act)
        c = act * cn + (1 - act) * c
        m = act * mn + (1 - act) * m
    return torch.stack(outs, 1), c, m


@pytest.mark.parametrize("peep,use_proj", [(True, True), (False, True), (True, False)])
def test_lstmp_fwd_bwd_vs_torch_autograd(oracle, peep, use_proj):
    rng = np.random.default_rng(0)
    B, T, I, N, P = 3, 7, 5, 6, 4
    Pout = P if use_proj else N
    x = rng.normal(size=(B, T, I))
    seq_len = np.array([7, 4, 1], np.int32)
    kernel = rng.normal(0, 0.4, size=(I + Pout, 4 * N))
    bias = rng.normal(0, 0.1, size=4 * N)
    wf, wi, wo = (rng.normal(0, 0.5, size=N) for _ in range(3)) if peep else (None, None, None)
    proj = rng.normal(0, 0.4, size=(N, P)) if use_proj else None
    d_out = rng.normal(size=(B, T, Pout))
    out, sv = oracle.lstmp_fwd(x, seq_len, kernel, bias, wf, wi, wo, proj, 5.0)
    dx, g = oracle.lstmp_bwd(sv, kernel, wf, wi, wo, proj, d_out)

    tt = lambda a: None if a is None else torch.tensor(a, dtype=torch.float64, requires_grad=True)
    tx, tk, tb, twf, twi, two, tp = map(tt, (x, kernel, bias, wf, wi, wo, proj))
    tout, tc, tm = _torch_lstmp(tx, torch.tensor(seq_len), tk, tb, twf, twi, two, tp, 5.0)
    (tout * torch.tensor(d_out)).sum().backward()
    np.testing.assert_allclose(out, tout.detach().numpy(), atol=1e-12)
    np.testing.assert_allclose(sv["final_c"], tc.detach().numpy(), atol=1e-12)
    np.testing.assert_allclose(sv["final_m"], tm.detach().numpy(), atol=1e-12)
    np.testing.assert_allclose(dx, tx.grad.numpy(), atol=1e-11)
    np.testing.assert_allclose(g["kernel"], tk.grad.numpy(), atol=1e-11)
    np.testing.assert_allclose(g["bias"], tb.grad.numpy(), atol=1e-11)
    if peep:
        np.testing.assert_allclose(g["w_f_diag"], twf.grad.numpy(), atol=1e-11)
        np.testing.assert_allclose(g["w_i_diag"], twi.grad.numpy(), atol=1e-11)
        np.testing.assert_allclose(g["w_o_diag"], two.grad.numpy(), atol=1e-11)
    if use_proj:
        np.testing.assert_allclose(g["proj"], tp.grad.numpy(), atol=1e-11)


def test_lstmp_vs_torch_nn_lstm(oracle):
    """Independent implementation: torch.nn.LSTM(proj_size) (gate order i,f,g,o; two biases)."""
    rng = np.random.default_rng(1)
    B, T, I, N, P = 2, 6, 4, 8, 3
    x = rng.normal(size=(B, T, I)).astype(np.float32)
    kernel = rng.normal(0, 0.3, size=(I + P, 4 * N)).astype(np.float32)
    bias = rng.normal(0, 0.1, size=4 * N).astype(np.float32)
    proj = rng.normal(0, 0.3, size=(N, P)).astype(np.float32)
    out, _ = oracle.lstmp_fwd(x, [T, T], kernel, bias, None, None, None, proj, 1.0)
    lstm = torch.nn.LSTM(I, N, proj_size=P, batch_first=True)
    Ki, Kj, Kf, Ko = np.split(kernel, 4, axis=1)
    bi, bj, bf, bo = np.split(bias, 4)
    Kt = np.concatenate([Ki, Kf, Kj, Ko], axis=1)          # torch order i,f,g,o
    bt = np.concatenate([bi, bf + 1.0, bj, bo])            # forget_bias folded into the bias
    with torch.no_grad():
        lstm.weight_ih_l0.copy_(torch.tensor(Kt[:I].T))
        lstm.weight_hh_l0.copy_(torch.tensor(Kt[I:].T))
        lstm.bias_ih_l0.copy_(torch.tensor(bt))
        lstm.bias_hh_l0.zero_()
        lstm.weight_hr_l0.copy_(torch.tensor(proj.T))
        tout, _ = lstm(torch.tensor(x))
    np.testing.assert_allclose(out, tout.numpy(), atol=2e-6)


def test_cudnnlstm_vs_torch_nn_lstm_stack(oracle):
    """nnet/lstm.py:26-122 by intent: num_layers plain LSTM cells (i, j, f, o; forget bias 0; no peepholes, projection or
    dropout) under dynamic_rnn's length masking + affine head, against torch.nn.LSTM(num_layers) on packed sequences."""
    rng = np.random.default_rng(2)
    cfg = dict(nnet_type="cudnnlstm", input_dim=5, left_context=0, right_context=0, num_layers=3, num_neurons=8,
               num_projects=8, num_targets=6, use_peepholes=True, dropout_rate=0.5)
    params = oracle.init_params(cfg, seed=3)
    assert sorted(params) == sorted(["rnn/multi_rnn_cell/cell_%d/cudnn_compatible_lstm_cell/%s" % (i, k)
                                     for i in range(3) for k in ("kernel", "bias")] + ["Variable", "Variable_1"])
    assert params["rnn/multi_rnn_cell/cell_0/cudnn_compatible_lstm_cell/kernel"].shape == (5 + 8, 32)
    assert params["rnn/multi_rnn_cell/cell_2/cudnn_compatible_lstm_cell/kernel"].shape == (8 + 8, 32)
    for k in params:
        if "bias" in k or k == "Variable_1":
            params[k] = rng.normal(0, 0.2, size=params[k].shape).astype(np.float32)
    B, T, N = 3, 7, 8
    seq_len = np.array([7, 5, 2], np.int32)
    x = rng.normal(size=(B, T, 5)).astype(np.float32)
    for b in range(B):
        x[b, seq_len[b]:] = 0
    logits, _ = oracle.forward(params, cfg, x, seq_len)
    lstm = torch.nn.LSTM(5, N, num_layers=3, batch_first=True)
    with torch.no_grad():
        for i in range(3):
            kernel = params["rnn/multi_rnn_cell/cell_%d/cudnn_compatible_lstm_cell/kernel" % i]
            bias = params["rnn/multi_rnn_cell/cell_%d/cudnn_compatible_lstm_cell/bias" % i]
            I = kernel.shape[0] - N
            Ki, Kj, Kf, Ko = np.split(kernel, 4, axis=1)
            bi, bj, bf, bo = np.split(bias, 4)
            Kt = np.concatenate([Ki, Kf, Kj, Ko], axis=1)      # torch order i,f,g,o; forget bias 0: nothing folded in
            getattr(lstm, "weight_ih_l%d" % i).copy_(torch.tensor(Kt[:I].T))
            getattr(lstm, "weight_hh_l%d" % i).copy_(torch.tensor(Kt[I:].T))
            getattr(lstm, "bias_ih_l%d" % i).copy_(torch.tensor(np.concatenate([bi, bf, bj, bo])))
            getattr(lstm, "bias_hh_l%d" % i).zero_()
        packed = torch.nn.utils.rnn.pack_padded_sequence(torch.tensor(x), torch.tensor(seq_len).long(), batch_first=True)
        out, _ = torch.nn.utils.rnn.pad_packed_sequence(lstm(packed)[0], batch_first=True, total_length=T)
        ref = out.numpy().reshape(B * T, N) @ params["Variable"] + params["Variable_1"]
    np.testing.assert_allclose(logits, ref.reshape(B, T, -1), atol=3e-6)


def test_reverse_sequence(oracle):
    x = np.arange(2 * 5 * 1, dtype=np.float64).reshape(2, 5, 1)
    y = oracle.reverse_sequence(x, [3, 5])
    assert list(y[0, :, 0]) == [2, 1, 0, 3, 4]
    assert list(y[1, :, 0]) == [9, 8, 7, 6, 5]


def _tiny_cfg(**kw):
    cfg = dict(nnet_type="blstm", input_dim=3, left_context=0, right_context=0, num_layers=2,
               num_neurons=4, num_projects=3, num_targets=5, use_peepholes=True, dropout_rate=1.0)
    cfg.update(kw)
    return cfg


@pytest.mark.parametrize("variant", ["plain", "moe", "residual", "noproj", "lstm", "dropout", "labelsm", "lstm_bn",
                                     "cudnnlstm"])
def test_model_grad_finite_difference(oracle, variant):
    """d(sum CTC loss [+reg]) / d(param) via the oracle's backward vs central differences (fp64)."""
    cfg = _tiny_cfg()
    if variant == "moe":
        cfg.update(num_experts=3, moe_temp=2.0)
    if variant == "residual":
        cfg.update(input_dim=6)                              # D == 2P -> first-layer residual (bilstm.py:199)
    if variant == "noproj":
        cfg.pop("num_projects")
    if variant == "lstm":
        cfg.update(nnet_type="lstm", input_dim=3, num_projects=3)   # D == P -> residual on layer 0 too
    if variant == "lstm_bn":
        cfg.update(nnet_type="lstm", input_dim=3, num_projects=3, use_bn=True)   # lstm.py:271-294
    if variant == "cudnnlstm":
        cfg.update(nnet_type="cudnnlstm", num_projects=4, dropout_rate=0.5)   # P == N and a keep-prob: both ignored
    if variant == "dropout":
        cfg.update(dropout_rate=0.7, num_experts=2)
    if variant == "labelsm":
        cfg.update(uniform_label_sm=0.3)
    rng = np.random.default_rng(4)
    params = {k: v.astype(np.float64) for k, v in oracle.init_params(cfg, seed=1, dtype=np.float64).items()}
    for k in params:
        if "bias" in k or k in ("Variable_1", "Variable_3") or k.endswith("/beta"):
            params[k] = rng.normal(0, 0.1, size=params[k].shape)
        if k.endswith("/gamma"):
            params[k] = rng.uniform(0.7, 1.3, size=params[k].shape)
    B, T = 3, 6
    x = rng.normal(size=(B, T, cfg["input_dim"]))
    seq_len = np.array([6, 4, 5], np.int32)
    for b in range(B):
        x[b, seq_len[b]:] = 0
    labels = np.array([[0, 1, -1], [2, 2, -1], [3, -1, -1]], np.int64)

    def f(p):
        return oracle.validation_graph(p, cfg, x, seq_len, labels, drop_seed=5)["loss"]

    out = oracle.validation_graph(params, cfg, x, seq_len, labels, drop_seed=5, want_grad=True)
    grads, _ = oracle.backward(params, cfg, out["saved"], np.ascontiguousarray(out["dlogits"]))
    trainable = {k for k in params if "/moving_" not in k}       # tf.trainable_variables()
    assert set(grads) == trainable
    eps = 1e-6
    for name in sorted(trainable):
        flat = params[name].reshape(-1)
        for idx in rng.choice(flat.size, size=min(3, flat.size), replace=False):
            old = flat[idx]
            flat[idx] = old + eps
            fp = f(params)
            flat[idx] = old - eps
            fm = f(params)
            flat[idx] = old
            fd = (fp - fm) / (2 * eps)
            an = grads[name].reshape(-1)[idx]
            assert abs(fd - an) < 1e-6 * max(1.0, abs(fd)), (variant, name, idx, fd, an)


def test_optimizers_and_clip(oracle):
    """Adam/SGD/momentum + L2 + global-norm clip vs a direct numpy restatement of App. A.6."""
    rng = np.random.default_rng(6)
    p = {"a/kernel": rng.normal(size=(3, 4)), "a/bias": rng.normal(size=4), "Variable_1": rng.normal(size=2)}
    g = {k: rng.normal(size=v.shape) * 10 for k, v in p.items()}
    cl, norm = oracle.l2_and_clip(p, g, clip_norm=5.0, l2=1e-5)
    g2 = {k: g[k] + (0 if "bias" in k else 1e-5 * p[k]) for k in p}
    n = np.sqrt(sum((v ** 2).sum() for v in g2.values()))
    assert abs(n - norm) < 1e-9 and n > 5
    for k in p:
        np.testing.assert_allclose(cl[k], g2[k] * 5.0 / n, rtol=1e-12)
    st = {}
    p0 = {k: v.copy() for k, v in p.items()}
    oracle.apply_optimizer("adam", p, cl, st, 1e-3)
    # first Adam step: m = .1 g, v = .001 g^2, lr_t = lr*sqrt(.001)/.1 -> step = lr * g/(|g| + eps/sqrt(.001)...)
    for k in p:
        gk = cl[k]
        lr_t = 1e-3 * np.sqrt(1 - 0.999) / (1 - 0.9)
        exp = p0[k] - lr_t * (0.1 * gk) / (np.sqrt(0.001 * gk ** 2) + 1e-8)
        np.testing.assert_allclose(p[k], exp, rtol=1e-12)
    st = {}
    q = {k: v.copy() for k, v in p0.items()}
    oracle.apply_optimizer("momentum", q, cl, st, 0.1)
    oracle.apply_optimizer("momentum", q, cl, st, 0.1)
    for k in q:
        np.testing.assert_allclose(q[k], p0[k] - 0.1 * cl[k] - 0.1 * 1.9 * cl[k], rtol=1e-12)


def test_running_stats(oracle):
    """nnet/funcs.py:48-54 label-weighted running means on a synthetic sequence of triples."""
    rs = oracle.RunningStats()
    triples = [(10, 25.0, 4.0), (0, 0.0, 0.0), (30, 45.0, 3.0)]
    for s, l, e in triples:
        assert not rs.update(s, l, e)
    assert abs(rs.loss - 70.0 / 40) < 1e-12 and abs(rs.acc - 7.0 / 40) < 1e-12 and rs.step == 3
    assert rs.update(5, float("nan"))


def test_basic_lstm_cell_tf_known_answers(oracle):
    """The dependency's own cell vector (TF r1.8 core_rnn_cell_test.py::testBasicLSTMCell, committed as
    tests/golden/tf_basic_lstm_known_answers.json): two stacked BasicLSTMCell(2), every kernel entry 0.5, zero biases, forget
    bias 1, x = [[1, 1]], every state entry 0.1.  Pins the oracle's gate order (i, j, f, o) and the forget-bias placement to
    numbers TensorFlow itself asserts - for the cell both `nnet/lstm.py:73-76` and `nnet/bilstm.py:129-136` instantiate."""
    import json
    import os
    ka = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "tf_basic_lstm_known_answers.json")))
    N, exp, tol = ka["num_units"], ka["expected"], ka["tolerance"]
    state = np.full((1, N), ka["initial_state_value"])
    x = np.asarray(ka["x"], np.float64)[:, None, :]                      # [B = 1, T = 1, I = 2]
    got = {}
    for layer, (ck, hk) in enumerate((("c1", "h1"), ("c2", "h2"))):
        I = x.shape[2]
        kernel = np.full((I + N, 4 * N), ka["kernel_value"])
        bias = np.full(4 * N, ka["bias_value"])
        for dt in (np.float64, np.float32):
            out, sv = oracle.lstmp_fwd(x.astype(dt), np.array([1], np.int32), kernel.astype(dt), bias.astype(dt), None, None,
                                       None, None, ka["forget_bias"], init_c=state.astype(dt), init_m=state.astype(dt))
            t = tol
            np.testing.assert_allclose(sv["final_c"], exp[ck], rtol=0, atol=t)
            np.testing.assert_allclose(sv["final_m"], exp[hk], rtol=0, atol=t)
            np.testing.assert_allclose(out[:, 0], exp[hk], rtol=0, atol=t)
            if dt == np.float64:
                got[ck], got[hk], nxt = float(sv["final_c"][0, 0]), float(sv["final_m"][0, 0]), out
        x = nxt                                                            # MultiRNNCell: layer 2's input is layer 1's h
    # a permuted gate order or a forget bias on another gate cannot reproduce these: e.g. i, f, j, o (Keras / cuDNN order)
    # gives c1' = 0.1 sigmoid(1.1) + sigmoid(1.1) tanh(2.1) - not equal only if the forget bias matters; check it does
    assert abs(0.1 / (1 + np.exp(-1.1)) + np.tanh(1.1) / (1 + np.exp(-1.1)) - exp["c1"]) > 1e-2


def test_torch_f64_restatement_equals_the_c_oracle(oracle):
    """`oracle/torch_f64.py` (the float64 restatement that gives TRUTH at the benched sizes on the GPU, tests/test_gpu_truth.py)
    run on the CPU against the C oracle: forward logits and - through torch autograd - the gradient of sum(logits * dlogits)
    with respect to every parameter, ragged lengths (a one-frame utterance), dropout masks, peepholes, projection, two layers."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import torch_f64
    rng = np.random.default_rng(0)
    B, T, D, N, P, V, L = 4, 15, 6, 8, 4, 5, 2
    cfg = dict(nnet_type="blstm", input_dim=D, num_layers=L, num_neurons=N, num_projects=P, num_targets=V, use_peepholes=True,
               dropout_rate=0.8)
    params, I = {}, D
    for i in range(L):
        for pre in ("fd%d/frnn%d" % (i, i), "bd%d/brnn%d" % (i, i)):
            params[pre + "/kernel"] = rng.normal(size=(I + P, 4 * N)) * 0.3
            params[pre + "/bias"] = rng.normal(size=4 * N) * 0.1
            for w in ("w_f_diag", "w_i_diag", "w_o_diag"):
                params[pre + "/" + w] = rng.normal(size=N) * 0.3
            params[pre + "/projection/kernel"] = rng.normal(size=(N, P)) * 0.3
        I = 2 * P
    params["Variable"], params["Variable_1"] = rng.normal(size=(2 * P, V)), rng.normal(size=V)
    x = rng.normal(size=(B, T, D))
    seq = np.array([15, 12, 7, 1], np.int32)
    dl = rng.normal(size=(B, T, V))
    for b in range(B):
        dl[b, seq[b]:] = 0
    ref_logits, saved = oracle.forward(params, cfg, x, seq, drop_seed=7)
    ref_grads, _ = oracle.backward(params, cfg, saved, dl)
    logits, grads = torch_f64.blstm_gradients(params, cfg, np.ascontiguousarray(x.transpose(1, 0, 2)), seq,
                                              np.ascontiguousarray(dl.transpose(1, 0, 2)), drop_seed=7, device="cpu")
    np.testing.assert_allclose(logits.numpy().transpose(1, 0, 2), ref_logits, rtol=0, atol=1e-12)
    assert set(grads) == set(ref_grads)
    for k in ref_grads:
        np.testing.assert_allclose(grads[k], ref_grads[k], rtol=0, atol=1e-11, err_msg=k)
