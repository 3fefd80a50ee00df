"""GPU parity tests at the widths BASELINE.json names, on the kernel instantiations bench.py actually runs.

Each case builds the model at its REAL width / head (c2: 3 x BiLSTM-320, V = 72; c3: 5 x BiLSTM-512 + high-rank head
E = V = 72; c4: BiLSTM-1024, V = 44; c5: the same with bf16 operands) on a few frames, asserts which recurrence
schedule the C ABI took (lc_debug_last_lstm_schedule) - so a test cannot silently land on another kernel than the one
the benchmark runs - and compares logits, CTC loss / gradient, greedy tokens and every parameter gradient with the
fp64 oracle (c5: with oracle/bf16_emulation.py, the float64 emulation with the product's operand roundings).

Tolerances as in test_gpu_model.py: logits and loss 1e-4 relative (north star), gradients 2e-3 of each tensor's largest
entry, tokens bit-exact.  The long-chain cases run T = 1000 in a CONTRACTIVE regime (recurrent weights scaled down,
forget gate < 1; the decay of a perturbation is asserted first), where - unlike with the reference's random
initialisation, DESIGN.md section 6 - a full-length recurrence can be compared with the oracle element by element.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _base(**kw):
    cfg = dict(nnet_type="blstm", input_dim=40, left_context=0, right_context=0, use_peepholes=True, dropout_rate=1.0)
    cfg.update(kw)
    return cfg


C2 = _base(num_layers=3, num_neurons=320, num_projects=320, num_targets=72)
C3 = _base(num_layers=5, num_neurons=512, num_projects=512, num_targets=72, num_experts=72, moe_temp=10.0)
C4_1 = _base(num_layers=1, num_neurons=1024, num_projects=1024, num_targets=44)
C4 = _base(num_layers=5, num_neurons=1024, num_projects=1024, num_targets=44)
C5 = dict(C4, compute_dtype="bf16")


def _batch(rng, cfg, B, T, ragged=True):
    D, V = cfg["input_dim"], cfg["num_targets"]
    seq = np.full(B, T, np.int32)
    if ragged:
        seq = np.sort(rng.integers(max(2, T // 2), T + 1, size=B))[::-1].astype(np.int32).copy()
        seq[0] = T
    x = rng.normal(size=(B, T, D)).astype(np.float32)
    Lmax = max(1, T // 2)
    labels = np.full((B, Lmax), -1, np.int64)
    for b in range(B):
        x[b, seq[b]:] = 0
        n = int(rng.integers(1, max(2, seq[b] // 2 + 1)))
        labels[b, :n] = rng.integers(0, V - 1, size=n)
        if n >= 2 and b % 3 == 0:
            labels[b, 1] = labels[b, 0]                      # adjacent repeat: the mandatory-blank path
    return x, seq, labels


def _randomise_biases(model, rng):
    params = model.ps.export_tf()
    for k in params:
        if "bias" in k or k in ("Variable_1", "Variable_3"):
            params[k] = rng.normal(0, 0.2, size=params[k].shape).astype(np.float32)
    model.ps.load_tf(params)
    return params


def _run_model(model, cfg, x, seq, labels):
    """forward + CTC + greedy + backward on the HIP path; returns host arrays and the schedules taken."""
    from lstm_ctc_amd import ops
    from lstm_ctc_amd.nnet.graph import flatten_labels
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    flat, offs, maxlen = flatten_labels(labels)
    xt, sl = dev(x.transpose(1, 0, 2)), dev(seq)
    logits = model.forward(xt, sl)
    sched_f = ops.last_lstm_schedule()
    loss, grad = ops.ctc_loss(logits, dev(flat), dev(offs), sl, maxlen)
    tok, n = ops.ctc_greedy(logits, sl)
    model.backward(grad)
    sched_b = ops.last_lstm_schedule()
    torch.cuda.synchronize()
    return dict(logits=logits.cpu().numpy().transpose(1, 0, 2), loss=loss.cpu().numpy(),
                dlogits=grad.cpu().numpy().transpose(1, 0, 2), tokens=tok.cpu().numpy(), token_len=n.cpu().numpy(),
                grads=model.ps.export_tf(grads=True), sched_f=sched_f, sched_b=sched_b, flat=flat, offs=offs)


def _check(got, ref_logits, ref_loss, ref_dlogits, ref_tokens, ref_len, ref_grads, logit_tol=1e-4, grad_tol=2e-3,
           loss_tol=1e-4, tag=""):
    scale = max(np.abs(ref_logits).max(), 1.0)
    err = np.abs(got["logits"] - ref_logits).max()
    assert err < logit_tol * scale, (tag, "logits", err, scale)
    fin = np.isfinite(ref_loss)
    assert np.array_equal(np.isfinite(got["loss"]), fin)
    assert np.all(np.abs(got["loss"][fin] - ref_loss[fin]) <= loss_tol * np.maximum(np.abs(ref_loss[fin]), 1.0)), (
        tag, "loss", got["loss"], ref_loss)
    assert np.abs(got["dlogits"] - ref_dlogits).max() < max(10 * logit_tol, 2e-4), (
        tag, "ctc grad", np.abs(got["dlogits"] - ref_dlogits).max())
    assert np.array_equal(got["token_len"], ref_len), (tag, "token counts")
    for b in range(len(ref_len)):
        assert np.array_equal(got["tokens"][b, :ref_len[b]], ref_tokens[b, :ref_len[b]]), (tag, "tokens", b)
    assert set(got["grads"]) == set(ref_grads)
    for k in sorted(ref_grads):
        tol = grad_tol * max(np.abs(ref_grads[k]).max(), 1e-3)
        e = np.abs(got["grads"][k] - ref_grads[k]).max()
        assert e < tol, (tag, k, e, tol)


def _oracle_reference(oracle, params, cfg, x, seq, labels):
    p64 = {k: v.astype(np.float64) for k, v in params.items()}
    ref = oracle.validation_graph(p64, cfg, x.astype(np.float64), seq, labels, want_grad=True)
    grads, _ = oracle.backward(p64, cfg, ref["saved"], ref["dlogits"])
    return ref, grads


FP32_CASES = {
    # name: (cfg, B, T, expected forward schedule (kind, mt), expected backward schedule (kind, mt), env)
    "c2_3x320_persistent": (C2, 32, 12, ("persistent_f32", 0), ("persistent_f32", 0), {}),
    "c3_5x512_moe_persistent": (C3, 32, 8, ("persistent_f32", 0), ("persistent_f32", 0), {}),
    # c4: two chains of lstm_fwd_step_kernel<2,false> on two streams, lstm_bwd_step_kernel<2,false>
    "c4_1024_b64_t16": (C4_1, 64, 16, ("two_stream_train", 2), ("launch_train", 2), {}),
    "c4_1024_b64_t40": (C4_1, 64, 40, ("two_stream_train", 2), ("launch_train", 2), {}),
    "c4_1024_b48_t16": (C4_1, 48, 16, ("two_stream_train", 2), ("launch_train", 2), {}),
    "c4_1024_b33_t5": (C4_1, 33, 5, ("two_stream_train", 2), ("launch_train", 2), {}),
    "c4_5x1024_b64_t8": (C4, 64, 8, ("two_stream_train", 2), ("launch_train", 2), {}),
    # 64-row tiles: lstm_fwd_step_kernel<4,false> (more than 64 batch rows)
    "c4_1024_b100_t6_mt4": (C4_1, 100, 6, ("launch_train", 4), ("launch_train", 2), {}),
    # the launch train as the fallback of the persistent schedule at c3's width
    "c3_512_launch_train": (dict(C3, num_layers=1), 64, 9, ("launch_train", 2), ("launch_train", 1),
                            {"LC_LSTM_PERSISTENT": "0"}),
}


@pytest.mark.parametrize("case", sorted(FP32_CASES))
def test_fp32_configs_vs_oracle(oracle, case, monkeypatch):
    from lstm_ctc_amd.nnet.model import Model
    cfg, B, T, want_f, want_b, env = FP32_CASES[case]
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    rng = np.random.default_rng(sum(map(ord, case)))
    x, seq, labels = _batch(rng, cfg, B, T)
    model = Model(cfg, "cuda", seed=17)
    params = _randomise_biases(model, rng)
    got = _run_model(model, cfg, x, seq, labels)
    assert (got["sched_f"]["kind"], got["sched_f"]["mt"] if want_f[1] else 0) == want_f, got["sched_f"]
    assert (got["sched_b"]["kind"], got["sched_b"]["mt"] if want_b[1] else 0) == want_b, got["sched_b"]
    assert not got["sched_f"]["bf16"] and got["sched_b"]["backward"]
    ref, ref_grads = _oracle_reference(oracle, params, cfg, x, seq, labels)
    _check(got, ref["logits"], ref["loss_per_utt"], ref["dlogits"], ref["tokens"], ref["token_len"], ref_grads, tag=case)


BF16_CASES = {
    # c5 as benched: one persistent launch per layer (bf16 MFMA, one XCD per direction and 16-row group)
    "c5_5x1024_b64_t8_persistent": (C5, 64, 8, "persistent_bf16", 0, "persistent_bf16", 0, {}),
    "c5_1024_b40_t12_persistent": (dict(C5, num_layers=1), 40, 12, "persistent_bf16", 0, "persistent_bf16", 0, {}),
    # its launch-train fallback: lstm_fwd_step_kernel<4,true> / lstm_bwd_step_kernel<2,true>
    "c5_1024_b64_t8_launch_train": (dict(C5, num_layers=2), 64, 8, "launch_train", 4, "launch_train", 2,
                                    {"LC_LSTM_PERSISTENT": "0"}),
}


@pytest.mark.parametrize("case", sorted(BF16_CASES))
def test_bf16_config_vs_emulation(oracle, case, monkeypatch):
    """c5 against the float64 emulation WITH THE SAME OPERAND ROUNDINGS (oracle/bf16_emulation.py).  What is left is
    accumulation order - plus a value that sits on a bf16 rounding boundary and falls to the other side in float32
    (one operand moves by 2^-9 relative, a pre-activation by ~1e-3 at worst): logits within 2e-3 of the logit scale
    with a median error below 2e-5, gradients within 5e-3 of each tensor's largest entry - an order of magnitude
    inside the distance to the fp32 oracle (3e-2 / 6e-2, test_gpu_model.py), so a wrong operand in any one of the
    ~20 products per layer shows."""
    from lstm_ctc_amd.nnet.model import Model
    from oracle import bf16_emulation as emu
    cfg, B, T, kf, mf, kb, mb, env = BF16_CASES[case]
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    rng = np.random.default_rng(sum(map(ord, case)))
    x, seq, labels = _batch(rng, cfg, B, T)
    model = Model(cfg, "cuda", seed=23)
    assert model.bf16
    params = _randomise_biases(model, rng)
    got = _run_model(model, cfg, x, seq, labels)
    assert got["sched_f"]["kind"] == kf and got["sched_f"]["bf16"] and (not mf or got["sched_f"]["mt"] == mf), got["sched_f"]
    assert got["sched_b"]["kind"] == kb and got["sched_b"]["bf16"] and (not mb or got["sched_b"]["mt"] == mb), got["sched_b"]
    ref_logits, saved = emu.forward(params, cfg, x, seq)
    scale = max(np.abs(ref_logits).max(), 1.0)
    err = np.abs(got["logits"] - ref_logits)
    assert err.max() < 2e-3 * scale and np.median(err) < 2e-5 * scale, (case, err.max(), np.median(err), scale)
    # the distance to the fp32 oracle is an order of magnitude larger: the test would notice fp32 operands too
    p64 = {k: v.astype(np.float64) for k, v in params.items()}
    cfg32 = {k: v for k, v in cfg.items() if k != "compute_dtype"}
    fp32_logits, _ = oracle.forward(p64, cfg32, x.astype(np.float64), seq)
    assert np.median(np.abs(got["logits"] - fp32_logits)) > 10 * np.median(err)
    # backward of the emulation from the kernel's own CTC gradient (CTC itself is fp32 and pinned elsewhere)
    ref_grads = emu.backward(params, cfg, saved, got["dlogits"].astype(np.float64))
    assert set(got["grads"]) == set(ref_grads)
    for k in sorted(ref_grads):
        tol = 5e-3 * max(np.abs(ref_grads[k]).max(), 1e-3)
        e = np.abs(got["grads"][k] - ref_grads[k]).max()
        assert e < tol, (case, k, e, tol)
    # CTC on these logits against the oracle's CTC on the same logits
    tbv = np.ascontiguousarray(got["logits"].transpose(1, 0, 2)).astype(np.float64)
    ref_loss, ref_grad, _ = oracle.ctc_loss(tbv, got["flat"], got["offs"], seq)
    fin = np.isfinite(ref_loss)
    assert np.allclose(got["loss"][fin], ref_loss[fin], rtol=1e-4, atol=1e-4)
    assert np.abs(got["dlogits"] - ref_grad.transpose(1, 0, 2)).max() < 2e-4


# ---------------------------------------------------------------------------------------------------- long chains
def _contractive(params, cfg, rng):
    """Scales the recurrent block of every LSTM kernel down and pulls the forget gate below 1 (bias -3 against the
    +5 forget_bias => f ~ sigmoid(1)), so that perturbations decay instead of growing."""
    N = cfg["num_neurons"]
    out = dict(params)
    for k, v in params.items():
        if k.endswith("/kernel") and "projection" not in k:
            v = v.copy()
            I = v.shape[0] - cfg["num_projects"]
            v[I:] *= 0.25
            out[k] = v
        if k.endswith("/bias"):
            b = rng.normal(0, 0.1, size=v.shape).astype(np.float32)
            b[2 * N:3 * N] -= 4.0
            out[k] = b
    return out


def _assert_decay(oracle, p64, cfg, B, D, seed):
    """A 1e-3 perturbation of the first 3 frames must have decayed by 1e-4 after 150 steps (forward direction) - and,
    mirrored, of the last 3 frames for the reverse direction."""
    rng = np.random.default_rng(seed)
    T = 160
    x = rng.normal(size=(B, T, D))
    seq = np.full(B, T, np.int32)
    base, _ = oracle.forward(p64, cfg, x, seq)
    xp = x.copy()
    xp[:, :3] += 1e-3
    xp[:, -3:] += 1e-3
    pert, _ = oracle.forward(p64, cfg, xp, seq)
    near = np.abs(pert - base)[:, :3].max()
    mid = np.abs(pert - base)[:, 75:85].max()          # >= 70 steps from either end
    assert near > 1e-6 and mid < 1e-4 * near, (near, mid)


LONG_FP32 = {
    # launch train lstm_{fwd,bwd}_step_kernel<1,false> over 1000 dependent launches per direction
    "n1024_b8_launch_train": (dict(C4_1, num_layers=1), 8, "launch_train", "launch_train"),
    # c2's layer: the persistent schedule over 1000 exchanges (16-byte tagged dz fragments, 8-byte state granules)
    "n320_b32_persistent": (dict(C2, num_layers=1), 32, "persistent_f32", "persistent_f32"),
}


@pytest.mark.parametrize("case", sorted(LONG_FP32))
def test_long_chain_contractive_vs_oracle(oracle, case):
    """T = 1000 against the fp64 oracle, element by element: a stale, torn or dropped exchange anywhere in the
    1000-step chain shows as an O(1e-2) error against the independent answer (the launch-train-vs-persistent
    comparison of test_gpu_ops.py cannot see an error both schedules share)."""
    from lstm_ctc_amd.nnet.model import Model
    cfg, B, kf, kb = LONG_FP32[case]
    T = 1000
    rng = np.random.default_rng(len(case))
    model = Model(cfg, "cuda", seed=5)
    params = _contractive(model.ps.export_tf(), cfg, rng)
    model.ps.load_tf(params)
    p64 = {k: v.astype(np.float64) for k, v in params.items()}
    _assert_decay(oracle, p64, cfg, 2, cfg["input_dim"], 3)
    x, seq, labels = _batch(rng, cfg, B, T)
    seq[-1] = 611                                       # one utterance ends well inside the chain
    x[-1, 611:] = 0
    labels[:, 40:] = -1                                 # keep the label sequences short: this test is about the chain
    got = _run_model(model, cfg, x, seq, labels)
    assert got["sched_f"]["kind"] == kf and got["sched_b"]["kind"] == kb, (got["sched_f"], got["sched_b"])
    ref, ref_grads = _oracle_reference(oracle, params, cfg, x, seq, labels)
    _check(got, ref["logits"], ref["loss_per_utt"], ref["dlogits"], ref["tokens"], ref["token_len"], ref_grads, tag=case)


def test_long_chain_contractive_bf16_vs_emulation(oracle):
    """The c5 persistent recurrence (N = 1024, bf16 operands, 4-byte tagged state granules forward, 16-byte dz fragments
    with generation bits backward) over T = 1000 against the float64 emulation with the same roundings."""
    from lstm_ctc_amd.nnet.model import Model
    from oracle import bf16_emulation as emu
    cfg = dict(C5, num_layers=1)
    B, T = 16, 1000
    rng = np.random.default_rng(99)
    model = Model(cfg, "cuda", seed=6)
    params = _contractive(model.ps.export_tf(), cfg, rng)
    model.ps.load_tf(params)
    x, seq, labels = _batch(rng, cfg, B, T)
    labels[:, 40:] = -1
    got = _run_model(model, cfg, x, seq, labels)
    assert got["sched_f"]["kind"] == "persistent_bf16" and got["sched_b"]["kind"] == "persistent_bf16"
    ref_logits, saved = emu.forward(params, cfg, x, seq)
    scale = max(np.abs(ref_logits).max(), 1.0)
    err = np.abs(got["logits"] - ref_logits)
    assert err.max() < 3e-3 * scale and np.median(err) < 2e-5 * scale, (err.max(), np.median(err), scale)
    ref_grads = emu.backward(params, cfg, saved, got["dlogits"].astype(np.float64))
    for k in sorted(ref_grads):
        tol = 5e-3 * max(np.abs(ref_grads[k]).max(), 1e-3)
        e = np.abs(got["grads"][k] - ref_grads[k]).max()
        assert e < tol, (k, e, tol)
