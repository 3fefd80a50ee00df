"""GPU parity tests at the widths BASELINE.json names, on the kernel instantiations bench.py actually runs.

Each case builds the model at its REAL width / head (c2: 3 x BiLSTM-320, V = 72; c3: 5 x BiLSTM-512 + high-rank head
E = V = 72; c4: BiLSTM-1024, V = 44; c5: the same with bf16 operands) on a few frames, asserts which recurrence
schedule the C ABI took (lc_debug_last_lstm_schedule) - so a test cannot silently land on another kernel than the one
the benchmark runs - and compares logits, CTC loss / gradient, greedy tokens and every parameter gradient with the
fp64 oracle (c5: with oracle/bf16_emulation.py, the float64 emulation with the product's operand roundings).

Tolerances as in test_gpu_model.py: logits and loss 1e-4 relative (north star), gradients `conftest.GRAD_TOL` (1e-4, measured) of each tensor's largest
entry, tokens bit-exact.  The long-chain cases run T = 1000 in a CONTRACTIVE regime (recurrent weights scaled down,
forget gate < 1; the decay of a perturbation is asserted first), where - unlike with the reference's random
initialisation, DESIGN.md section 6 - a full-length recurrence can be compared with the oracle element by element.
"""
import os

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
UNI_1024 = dict(nnet_type="lstm", input_dim=40, left_context=0, right_context=0, num_layers=1, num_neurons=1024,
                num_projects=1024, num_targets=44, dropout_rate=1.0)


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


def _download_forward(model):
    """The layer inputs and saved activations of the forward pass just run (host copies, time-major rows)."""
    out = []
    for L in model.saved["layers"]:
        out.append(dict(inp=L["inp"].cpu().numpy(), Y=L["Y"].cpu().numpy(),
                        dirs=[dict(gates=d["zx"].cpu().numpy(), cs=d["cs"].cpu().numpy(), hs=d["hs"].cpu().numpy(),
                                   reverse=bool(d["reverse"])) for d in L["dirs"]]))
    return out


def _x3(cfg):
    """The same configuration with fp32 products as bf16x3 (split operands): the mode the parity cases are run in a second time."""
    return dict(cfg, compute_dtype="bf16x3")


def _x3_kind(kind, cfg, backward=False):
    """Schedule a case takes in bf16x3 mode: the split-operand recurrence where a kernel exists for the width (XCD pairs at
    N = 768 / 1024; one XCD at N = 64 .. 512 in steps of 64 - forward and BPTT both), else the fp32 schedule."""
    if kind == "persistent_f32_xcd_pair" and cfg["num_neurons"] in (768, 1024):
        return "persistent_x3_xcd_pair"
    if kind == "persistent_f32" and cfg["num_neurons"] % 64 == 0 and 64 <= cfg["num_neurons"] <= 512:
        # (the FORWARD recurrence up to 448 units stays on the fp32 kernel, the faster one there: model.X3_FWD_MIN_N)
        return "persistent_x3" if backward or cfg["num_neurons"] > 448 else kind
    return kind


def _run_model(model, cfg, x, seq, labels, keep_forward=False):
    """forward + CTC + greedy + backward on the HIP path; returns host arrays, the schedules taken and the kinds of
    products issued (ops.PROFILE: "gemm", "gemm_x3", ...)."""
    from lstm_ctc_amd import ops
    from lstm_ctc_amd.nnet.graph import flatten_labels
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    flat, offs, maxlen = flatten_labels(labels)
    xt, sl = dev(x.transpose(1, 0, 2)), dev(seq)
    ops.PROFILE = []
    try:
        return _run_model_profiled(model, ops, xt, sl, dev, flat, offs, maxlen, keep_forward)
    finally:
        ops.PROFILE = None


def _run_model_profiled(model, ops, xt, sl, dev, flat, offs, maxlen, keep_forward):
    logits = model.forward(xt, sl)
    sched_f = ops.last_lstm_schedule()
    fwd_saved = _download_forward(model) if keep_forward else None
    loss, grad = ops.ctc_loss(logits, dev(flat), dev(offs), sl, maxlen)
    tok, n = ops.ctc_greedy(logits, sl)
    model.backward(grad)
    sched_b = ops.last_lstm_schedule()
    torch.cuda.synchronize()
    return dict(logits=logits.cpu().numpy().transpose(1, 0, 2), loss=loss.cpu().numpy(),
                dlogits=grad.cpu().numpy().transpose(1, 0, 2), tokens=tok.cpu().numpy(), token_len=n.cpu().numpy(),
                grads=model.ps.export_tf(grads=True), sched_f=sched_f, sched_b=sched_b, flat=flat, offs=offs,
                forward=fwd_saved, kinds={k for k, _, _, _ in ops.PROFILE})


def _check(got, ref_logits, ref_loss, ref_dlogits, ref_tokens, ref_len, ref_grads, logit_tol=1e-4, grad_tol=None,
           loss_tol=1e-4, tag="", elementwise=False):
    scale = max(np.abs(ref_logits).max(), 1.0)
    err = np.abs(got["logits"] - ref_logits).max()
    assert err < logit_tol * scale, (tag, "logits", err, scale)
    if elementwise:
        # "1e-4 relative" read per logit, not per tensor: every logit that is not small against the logit scale (>= 10 %
        # of it) is within 1e-4 of ITS OWN magnitude, and no logit is off by more than 2e-5 of the scale (measured on the
        # fp32 path: 2e-6 of the scale, 1.6e-5 relative - tools/relerr_probe.py)
        e = np.abs(got["logits"] - ref_logits)
        assert np.all(e <= logit_tol * np.maximum(np.abs(ref_logits), 0.1 * scale)), (tag, "logits, elementwise")
        assert err < 0.2 * logit_tol * scale, (tag, "logits, absolute", err, scale)
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
    from conftest import check_grad
    for k in sorted(ref_grads):
        check_grad(got["grads"][k], ref_grads[k], tag, k, **({} if grad_tol is None else {"tol": grad_tol}))


def _oracle_reference(oracle, params, cfg, x, seq, labels):
    p64 = {k: v.astype(np.float64) for k, v in params.items()}
    ref = oracle.validation_graph(p64, cfg, x.astype(np.float64), seq, labels, want_grad=True)
    grads, _ = oracle.backward(p64, cfg, ref["saved"], ref["dlogits"])
    return ref, grads


FP32_CASES = {
    # name: (cfg, B, T, expected forward schedule (kind, mt), expected backward schedule (kind, mt), env)
    "c2_3x320_persistent": (C2, 32, 12, ("persistent_f32", 0), ("persistent_f32", 0), {}),
    "c3_5x512_moe_persistent": (C3, 32, 8, ("persistent_f32", 0), ("persistent_f32", 0), {}),
    # c4 as benched: the XCD-pair persistent recurrence (R resident in registers, K split over two XCDs)
    "c4_1024_b64_t16": (C4_1, 64, 16, ("persistent_f32_xcd_pair", 0), ("persistent_f32_xcd_pair", 0), {}),
    "c4_1024_b64_t40": (C4_1, 64, 40, ("persistent_f32_xcd_pair", 0), ("persistent_f32_xcd_pair", 0), {}),
    "c4_1024_b48_t16": (C4_1, 48, 16, ("persistent_f32_xcd_pair", 0), ("persistent_f32_xcd_pair", 0), {}),
    "c4_1024_b33_t5": (C4_1, 33, 5, ("persistent_f32_xcd_pair", 0), ("persistent_f32_xcd_pair", 0), {}),
    "c4_1024_b17_t9": (C4_1, 17, 9, ("persistent_f32_xcd_pair", 0), ("persistent_f32_xcd_pair", 0), {}),
    "c4_5x1024_b64_t8": (C4, 64, 8, ("persistent_f32_xcd_pair", 0), ("persistent_f32_xcd_pair", 0), {}),
    # the same models with every eligible product on the 256 x 256 LDS-DMA GEMM kernel (as at T = 1000; at these sizes it
    # would not be chosen on its own): strided operands, accumulating outputs, all operand forms, in front of the oracle
    "c4_5x1024_b64_t8_big_gemm": (C4, 64, 8, ("persistent_f32_xcd_pair", 0), ("persistent_f32_xcd_pair", 0),
                                  {"LC_GEMM_F32_BIG": "2"}),
    "c3_5x512_moe_big_gemm": (C3, 32, 8, ("persistent_f32", 0), ("persistent_f32", 0), {"LC_GEMM_F32_BIG": "2"}),
    # its checked fallback: two chains of lstm_fwd_step_kernel<2,false> on two streams, lstm_bwd_step_kernel<2,false>
    "c4_1024_b64_t16_launch_train": (C4_1, 64, 16, ("two_stream_train", 2), ("launch_train", 2), {"LC_LSTM_PERSISTENT": "0"}),
    "c4_1024_b33_t5_launch_train": (C4_1, 33, 5, ("two_stream_train", 2), ("launch_train", 2), {"LC_LSTM_PERSISTENT": "0"}),
    "c4_5x1024_b64_t8_launch_train": (C4, 64, 8, ("two_stream_train", 2), ("launch_train", 2), {"LC_LSTM_PERSISTENT": "0"}),
    # more than 64 batch rows: the XCD-pair kernels over 64-row blocks, one launch per block (round 3; the second block of
    # B = 100 has 36 live rows), and - persistent schedules off - lstm_fwd_step_kernel<4,false> on 64-row tiles
    "c4_1024_b100_t6": (C4_1, 100, 6, ("persistent_f32_xcd_pair", 0), ("persistent_f32_xcd_pair", 0), {}),
    "c4_1024_b128_t5": (C4_1, 128, 5, ("persistent_f32_xcd_pair", 0), ("persistent_f32_xcd_pair", 0), {}),
    "c4_2x1024_b70_t9": (dict(C4, num_layers=2), 70, 9, ("persistent_f32_xcd_pair", 0), ("persistent_f32_xcd_pair", 0), {}),
    "c4_1024_b100_t6_mt4": (C4_1, 100, 6, ("launch_train", 4), ("launch_train", 2), {"LC_LSTM_PERSISTENT": "0"}),
    # ONE direction of 1024 units (nnet_type lstm): the two direction slots of a launch are two 64-row blocks of the same
    # direction - 128 rows per launch; B = 40 leaves the second slot idle, B = 150 takes two launches
    "uni_1024_b40_t7": (UNI_1024, 40, 7, ("persistent_f32_xcd_pair", 0), ("persistent_f32_xcd_pair", 0), {}),
    "uni_1024_b128_t5": (UNI_1024, 128, 5, ("persistent_f32_xcd_pair", 0), ("persistent_f32_xcd_pair", 0), {}),
    "uni_2x1024_b150_t6": (dict(UNI_1024, num_layers=2), 150, 6, ("persistent_f32_xcd_pair", 0),
                           ("persistent_f32_xcd_pair", 0), {}),
    # the XCD-pair schedule at the other widths it is built for (N = 128 NKB, NKB = 5, 6, 7: fewer workgroups per XCD, a
    # shorter MFMA stream with the post-processing pieces moved up): both directions, row blocks, one direction
    "pair_640_b40_t9": (dict(C4_1, num_neurons=640, num_projects=640), 40, 9, ("persistent_f32_xcd_pair", 0),
                        ("persistent_f32_xcd_pair", 0), {}),
    "pair_768_b64_t16": (dict(C4_1, num_neurons=768, num_projects=768), 64, 16, ("persistent_f32_xcd_pair", 0),
                         ("persistent_f32_xcd_pair", 0), {}),
    "pair_896_b64_t12": (dict(C4_1, num_neurons=896, num_projects=512), 64, 12, ("persistent_f32_xcd_pair", 0),
                         ("persistent_f32_xcd_pair", 0), {}),
    "pair_2x768_b70_t9": (dict(C4, num_layers=2, num_neurons=768, num_projects=384), 70, 9, ("persistent_f32_xcd_pair", 0),
                          ("persistent_f32_xcd_pair", 0), {}),
    "uni_896_b100_t6": (dict(UNI_1024, num_neurons=896, num_projects=896), 100, 6, ("persistent_f32_xcd_pair", 0),
                        ("persistent_f32_xcd_pair", 0), {}),
    "uni_640_b130_t7": (dict(UNI_1024, num_neurons=640, num_projects=320), 130, 7, ("persistent_f32_xcd_pair", 0),
                        ("persistent_f32_xcd_pair", 0), {}),
    # the launch train as the fallback of the persistent schedule at c3's width
    "c3_512_launch_train": (dict(C3, num_layers=1), 64, 9, ("launch_train", 2), ("launch_train", 1),
                            {"LC_LSTM_PERSISTENT": "0"}),
}


@pytest.mark.parametrize("dtype", ["fp32", "bf16x3"])
@pytest.mark.parametrize("case", sorted(FP32_CASES))
def test_fp32_configs_vs_oracle(oracle, case, dtype, monkeypatch):
    """Every case twice: on the fp32 MFMA kernels, and with compute_dtype = bf16x3 - the products with an activation operand
    as six bf16 term products (lc_gemm_bf16x3_*: forced for these few frames, at T = 1000 they are taken on their own) and,
    at N = 768 / 1024, the split-operand recurrences (lstm_pair_x3.inc) - AT THE SAME TOLERANCES: the split-operand mode is
    offered as fp32 arithmetic and is held to what fp32 is held to."""
    from lstm_ctc_amd.nnet import model as model_mod
    from lstm_ctc_amd.nnet.model import Model
    cfg, B, T, want_f, want_b, env = FP32_CASES[case]
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    if dtype == "bf16x3":
        monkeypatch.setattr(model_mod, "X3_FORCE", True)
        cfg = _x3(cfg)
        want_f, want_b = (_x3_kind(want_f[0], cfg), want_f[1]), (_x3_kind(want_b[0], cfg, True), want_b[1])
    rng = np.random.default_rng(sum(map(ord, case)))
    x, seq, labels = _batch(rng, cfg, B, T)
    model = Model(cfg, "cuda", seed=17)
    assert model.x3 == (dtype == "bf16x3")
    params = _randomise_biases(model, rng)
    got = _run_model(model, cfg, x, seq, labels)
    assert (got["sched_f"]["kind"], got["sched_f"]["mt"] if want_f[1] else 0) == want_f, got["sched_f"]
    assert (got["sched_b"]["kind"], got["sched_b"]["mt"] if want_b[1] else 0) == want_b, got["sched_b"]
    assert not got["sched_f"]["bf16"] and got["sched_b"]["backward"]
    assert ("gemm_x3" in got["kinds"]) == (dtype == "bf16x3"), got["kinds"]      # the split-operand product kernels ran
    ref, ref_grads = _oracle_reference(oracle, params, cfg, x, seq, labels)
    _check(got, ref["logits"], ref["loss_per_utt"], ref["dlogits"], ref["tokens"], ref["token_len"], ref_grads,
           tag=case + "/" + dtype, elementwise=True)


BF16_CASES = {
    # c5 as benched: one persistent launch per layer (bf16 MFMA, one XCD per direction and 16-row group)
    "c5_5x1024_b64_t8_persistent": (C5, 64, 8, "persistent_bf16", 0, "persistent_bf16", 0, {}),
    "c5_1024_b40_t12_persistent": (dict(C5, num_layers=1), 40, 12, "persistent_bf16", 0, "persistent_bf16", 0, {}),
    # its launch-train fallback: lstm_fwd_step_kernel<4,true> / lstm_bwd_step_kernel<2,true>
    "c5_1024_b64_t8_launch_train": (dict(C5, num_layers=2), 64, 8, "launch_train", 4, "launch_train", 2,
                                    {"LC_LSTM_PERSISTENT": "0"}),
}


def _interleaved_to_tf(a, N):
    """[.., 4N] gate-interleaved columns ((n/8)*32 + g*8 + n%8) -> TF's [i|j|f|o] blocks."""
    n = np.arange(N)
    idx = np.concatenate([(n // 8) * 32 + g * 8 + (n % 8) for g in range(4)])
    return a[..., idx]


def _teacher_forced_forward_check(emu, params, cfg, fwd, logits_tb, seq, T, B, tag):
    """Every forward product of the bf16 path on its own, from the KERNEL'S OWN operands (its layer inputs and its
    previous recurrent state), so that a rounding-boundary flip cannot cascade through the recurrence: what is left is
    float32 accumulation order and the hardware exp2 / rcp of the gate math (bound 1e-4 relative to max(1, |value|);
    measured 3e-5).  A wrong operand, rounding mode or layout in any single product shows at 1e-3 and above."""
    from lstm_ctc_amd import ops
    N, P = cfg["num_neurons"], cfg["num_projects"]
    bf = emu._bf
    sig = lambda v: 1.0 / (1.0 + np.exp(-v))
    worst = 0.0
    for i, L in enumerate(fwd):
        inp = L["inp"].astype(np.float64)
        for d, dd in enumerate(L["dirs"]):
            pre = ("bd%d/brnn%d" if d else "fd%d/frnn%d") % (i, i)
            k = params[pre + "/kernel"].astype(np.float64)
            I = inp.shape[1]
            Kx, Kh, proj = k[:I], k[I:], params[pre + "/projection/kernel"].astype(np.float64)
            zx = (bf(inp) @ bf(Kx) + params[pre + "/bias"]).reshape(T, B, 4 * N)
            # R = proj . Kh is a float32 product of two weights in the product (not a bf16 product): take the kernel's own
            # (bit-identical per element whatever the column order), so that its bf16 image is the one the step used
            Rf = ops.gemm(torch.from_numpy(params[pre + "/projection/kernel"]).cuda(),
                          torch.from_numpy(np.ascontiguousarray(params[pre + "/kernel"][I:])).cuda()).cpu().numpy()
            Rb = bf(Rf)
            wf, wi, wo = (params[pre + "/w_%s_diag" % g].astype(np.float64) for g in "fio")
            gates = _interleaved_to_tf(dd["gates"].astype(np.float64), N).reshape(T, B, 4 * N)
            cs = dd["cs"].astype(np.float64).reshape(T, B, N)
            hs = dd["hs"].astype(np.float64).reshape(T, B, N)
            for t in range(T):
                tp = t + 1 if dd["reverse"] else t - 1
                hq = bf(hs[tp]) if 0 <= tp < T else np.zeros((B, N))
                cp = cs[tp] if 0 <= tp < T else np.zeros((B, N))
                z = zx[t] + hq @ Rb
                ia = sig(z[:, :N] + wi * cp); fa = sig(z[:, 2 * N:3 * N] + 5.0 + wf * cp); ja = np.tanh(z[:, N:2 * N])
                cn = fa * cp + ia * ja
                oa = sig(z[:, 3 * N:] + wo * cn)
                act = (t < seq)[:, None]
                want = np.where(act, np.concatenate([ia, ja, fa, oa, cn, oa * np.tanh(cn)], axis=1), 0.0)
                have = np.concatenate([gates[t], cs[t], hs[t]], axis=1)
                e = (np.abs(have - want) / np.maximum(1.0, np.abs(want))).max()
                assert e < 5e-5, (tag, "layer", i, "dir", d, "t", t, e)
                worst = max(worst, e)
            e = np.abs(L["Y"][:, d * P:(d + 1) * P] - bf(hs.reshape(T * B, N)) @ bf(proj)).max()
            assert e < 1e-4, (tag, "projection", i, d, e)
    top = fwd[-1]["Y"].astype(np.float64)
    want = bf(top) @ bf(params["Variable"]) + params["Variable_1"]
    e = np.abs(logits_tb.reshape(T * B, -1) - want).max()
    assert e < 1e-4, (tag, "head", e)
    return worst


@pytest.mark.parametrize("case", sorted(BF16_CASES))
def test_bf16_config_vs_emulation(oracle, case, monkeypatch):
    """c5 against the float64 emulation WITH THE SAME OPERAND ROUNDINGS (oracle/bf16_emulation.py), two ways.
    (1) Teacher-forced, product by product from the kernel's own operands: agreement to float32 accumulation order
    (1e-4 on gates / states / outputs / logits) - a wrong operand in any one of the products shows.
    (2) End to end.  Here a state value that sits on a bf16 rounding boundary falls to the other side in float32, the
    flipped operand (2^-9 relative) perturbs 4N pre-activations by ~1e-5, which makes further flips 100x likelier: the
    two evaluations decorrelate at the level of the rounding noise itself within a few steps (measured: median 1.3e-4,
    max 1.9e-3 of the logit scale at T = 12 - about half the distance between the bf16 path and the fp32 oracle).  So
    this part is a sanity bound (logits within 4e-3 of the logit scale, median below 5e-4, every gradient within 1e-2
    of its largest entry); the discriminating check is (1)."""
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
    got = _run_model(model, cfg, x, seq, labels, keep_forward=True)
    assert got["sched_f"]["kind"] == kf and got["sched_f"]["bf16"] and (not mf or got["sched_f"]["mt"] == mf), got["sched_f"]
    assert got["sched_b"]["kind"] == kb and got["sched_b"]["bf16"] and (not mb or got["sched_b"]["mt"] == mb), got["sched_b"]
    _teacher_forced_forward_check(emu, params, cfg, got["forward"], got["logits"].transpose(1, 0, 2), seq, T, B, case)
    ref_logits, saved = emu.forward(params, cfg, x, seq)
    scale = max(np.abs(ref_logits).max(), 1.0)
    err = np.abs(got["logits"] - ref_logits)
    assert err.max() < 4e-3 * scale and np.median(err) < 5e-4 * scale, (case, err.max(), np.median(err), scale)
    # backward of the emulation from the kernel's own CTC gradient (CTC itself is fp32 and pinned elsewhere)
    ref_grads = emu.backward(params, cfg, saved, got["dlogits"].astype(np.float64))
    assert set(got["grads"]) == set(ref_grads)
    for k in sorted(ref_grads):
        tol = 1e-2 * max(np.abs(ref_grads[k]).max(), 1e-3)
        e = np.abs(got["grads"][k] - ref_grads[k]).max()
        assert e < tol, (case, k, e, tol)
    # CTC on these logits against the oracle's CTC on the same logits
    tbv = np.ascontiguousarray(got["logits"].transpose(1, 0, 2)).astype(np.float64)
    ref_loss, ref_grad, _ = oracle.ctc_loss(tbv, got["flat"], got["offs"], seq)
    fin = np.isfinite(ref_loss)
    assert np.allclose(got["loss"][fin], ref_loss[fin], rtol=1e-4, atol=1e-4)
    assert np.abs(got["dlogits"] - ref_grad.transpose(1, 0, 2)).max() < 2e-4


@pytest.mark.parametrize("persistent", [True, False])
def test_bf16_bptt_teacher_forced_at_c5_width(oracle, persistent, monkeypatch):
    """lc_lstm_bwd_bf16 at N = 1024, B = 64 (persistent launch / lstm_bwd_step_kernel<2,true>): every BPTT step from the
    KERNEL'S OWN previous dz, so nothing cascades: dz_t = f(dh_t + bf16(dz_{t'}) . bf16(R^T), saved gates / cells)."""
    from lstm_ctc_amd import ops
    monkeypatch.setenv("LC_LSTM_PERSISTENT", "1" if persistent else "0")
    T, B, N = 6, 64, 1024
    rng = np.random.default_rng(4)
    seq = np.full(B, T, np.int32); seq[-3:] = [4, 3, 1]
    n = np.arange(N)
    cols = [(n // 8) * 32 + g * 8 + (n % 8) for g in range(4)]
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()
    dirs = []
    for d in range(2):
        gates = rng.uniform(0.05, 0.95, size=(T, B, 4 * N))
        gates[:, :, cols[1]] = rng.uniform(-0.9, 0.9, size=(T, B, N))            # tanh(j)
        for b in range(B):
            gates[seq[b]:, b] = 0
        dirs.append(dict(gates=gates.astype(np.float32), RT=rng.normal(0, 0.03, size=(4 * N, N)).astype(np.float32),
                         cs=rng.normal(0, 0.5, size=(T, B, N)).astype(np.float32),
                         dh=rng.normal(0, 0.1, size=(T, B, N)).astype(np.float32),
                         w=[rng.normal(0, 0.3, size=N).astype(np.float32) for _ in range(3)], reverse=d))
    bd = [dict(gates=dev(dd["gates"].reshape(T * B, 4 * N)), RT=dev(dd["RT"]), w_f=dev(dd["w"][0]), w_i=dev(dd["w"][1]),
               w_o=dev(dd["w"][2]), cs=dev(dd["cs"].reshape(T * B, N)), dh=dev(dd["dh"].reshape(T * B, N)),
               dpeep=torch.zeros(3, N, device="cuda"), dbias=torch.zeros(4 * N, device="cuda"), reverse=dd["reverse"])
          for dd in dirs]
    ops.lstm_bwd(bd, dev(seq).int(), T, B, N, bf16=True)
    sch = ops.last_lstm_schedule()
    assert sch["kind"] == ("persistent_bf16" if persistent else "launch_train") and sch["bf16"] and sch["backward"]
    assert persistent or sch["mt"] == 2
    for d, dd in enumerate(dirs):
        dz = bd[d]["gates"].cpu().numpy().reshape(T, B, 4 * N).astype(np.float64)
        RTb = oracle.bf16_round(dd["RT"]).astype(np.float64)
        wf, wi, wo = (w.astype(np.float64) for w in dd["w"])
        g = dd["gates"].astype(np.float64)
        cs = dd["cs"].astype(np.float64)
        dc = np.zeros((B, N))
        order = list(range(T)) if dd["reverse"] else list(range(T - 1, -1, -1))
        for s_, t in enumerate(order):
            tprev = t + 1 if dd["reverse"] else t - 1
            cp = cs[tprev] if 0 <= tprev < T else np.zeros((B, N))
            dzq = oracle.bf16_round(dz[order[s_ - 1]].astype(np.float32)).astype(np.float64) if s_ else np.zeros((B, 4 * N))
            dh = dd["dh"][t].astype(np.float64) + dzq @ RTb
            ia, ja, fa, oa = (g[t][:, c] for c in cols)
            cn = cs[t]; tc = np.tanh(cn)
            do_pre = dh * tc * oa * (1 - oa)
            dcn = dc + dh * oa * (1 - tc * tc) + do_pre * wo
            di_pre = dcn * ja * ia * (1 - ia); dj_pre = dcn * ia * (1 - ja * ja); df_pre = dcn * cp * fa * (1 - fa)
            act = (t < seq)[:, None]
            dc = np.where(act, dcn * fa + di_pre * wi + df_pre * wf, dc)
            want = np.zeros((B, 4 * N))
            for c, v in zip(cols, (di_pre, dj_pre, df_pre, do_pre)):
                want[:, c] = np.where(act, v, 0.0)
            scale = max(np.abs(want).max(), 1e-6)
            e = np.abs(dz[t] - want).max()
            # the carried cell gradient dc is the emulation's own (exact in both), only dz_{t'} is teacher-forced
            assert e < 3e-5 * max(scale, 1.0) + 2e-6, (d, t, e, scale)
        # bias / peephole gradients from the kernel's dz: deterministic float32 sums
        want_b = dz.reshape(T * B, 4 * N).sum(0)
        assert np.abs(bd[d]["dbias"].cpu().numpy() - want_b).max() < 1e-4 * max(1.0, np.abs(want_b).max())


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
    # c4's layer: the XCD-pair persistent recurrence and BPTT over 1000 steps (state / dz fragments through each XCD's
    # L2, partial sums across the fabric every step)
    "n1024_b8_xcd_pair": (dict(C4_1, num_layers=1), 8, "persistent_f32_xcd_pair", "persistent_f32_xcd_pair"),
    "n1024_b40_xcd_pair": (dict(C4_1, num_layers=1), 40, "persistent_f32_xcd_pair", "persistent_f32_xcd_pair"),
    # more than 64 rows: two launches over row blocks of the same [T, B, *] tensors (the second block: 8 live rows)
    "n1024_b72_two_blocks": (dict(C4_1, num_layers=1), 72, "persistent_f32_xcd_pair", "persistent_f32_xcd_pair"),
    # the other widths of the pair schedule over the same 1000 steps
    "n640_b40_xcd_pair": (dict(C4_1, num_layers=1, num_neurons=640, num_projects=640), 40, "persistent_f32_xcd_pair",
                          "persistent_f32_xcd_pair"),
    "n768_b64_xcd_pair": (dict(C4_1, num_layers=1, num_neurons=768, num_projects=768), 64, "persistent_f32_xcd_pair",
                          "persistent_f32_xcd_pair"),
    "n896_b24_xcd_pair": (dict(C4_1, num_layers=1, num_neurons=896, num_projects=896), 24, "persistent_f32_xcd_pair",
                          "persistent_f32_xcd_pair"),
    # c2's layer: the persistent schedule over 1000 exchanges (16-byte tagged dz fragments, 8-byte state granules)
    "n320_b32_persistent": (dict(C2, num_layers=1), 32, "persistent_f32", "persistent_f32"),
    # c3's width (one XCD per direction and 16-row group; in bf16x3 mode the split-operand forward kernel)
    "n512_b32_persistent": (dict(C2, num_layers=1, num_neurons=512, num_projects=512), 32, "persistent_f32", "persistent_f32"),
    # a width whose 32-blocks do not divide by the four waves (14: the split-operand forward kernel deals 3, 4, 3, 4; its BPTT
    # walks 14 blocks per wave in chunks of 5, 5, 4), ragged row groups (20 rows over four XCDs per direction)
    "n448_b20_persistent": (dict(C2, num_layers=1, num_neurons=448, num_projects=448), 20, "persistent_f32", "persistent_f32"),
}


@pytest.mark.parametrize("dtype", ["fp32", "bf16x3"])
@pytest.mark.parametrize("case", sorted(LONG_FP32))
def test_long_chain_contractive_vs_oracle(oracle, case, dtype):
    """T = 1000 against the fp64 oracle, element by element: a stale, torn or dropped exchange anywhere in the
    1000-step chain shows as an O(1e-2) error against the independent answer (the launch-train-vs-persistent
    comparison of test_gpu_ops.py cannot see an error both schedules share).  In bf16x3 mode (no forcing: at T = 1000 the
    products of the 1024- / 768-wide layers go to the split-operand kernels on their own) the same chains run through the
    split-operand recurrences - XCD pairs and single XCD: consumer- or producer-split state forward, producer-split dz pieces
    backward."""
    from lstm_ctc_amd.nnet.model import Model
    cfg, B, kf, kb = LONG_FP32[case]
    if dtype == "bf16x3":
        cfg = _x3(cfg)
        kf, kb = _x3_kind(kf, cfg), _x3_kind(kb, cfg, True)
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
    if dtype == "bf16x3" and cfg["num_neurons"] >= 640:
        assert "gemm_x3" in got["kinds"], got["kinds"]
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
    # contractive: a rounding-boundary flip is damped instead of cascading; what stays is the per-step flip noise
    # (measured: max 6.6e-4, median 4.3e-5 of the logit scale)
    assert err.max() < 3e-3 * scale and np.median(err) < 1.5e-4 * scale, (err.max(), np.median(err), scale)
    ref_grads = emu.backward(params, cfg, saved, got["dlogits"].astype(np.float64))
    for k in sorted(ref_grads):
        tol = 5e-3 * max(np.abs(ref_grads[k]).max(), 1e-3)
        e = np.abs(got["grads"][k] - ref_grads[k]).max()
        assert e < tol, (k, e, tol)
