"""GPU parity fuzz: random small model shapes (batch sizes across the row-tile boundaries 16/32/48/64/128, unit counts
that are not multiples of 32, with and without projection / peepholes / MoE / residual) against the fp64 oracle,
forward logits and all gradients.  Complements the fixed variants of test_gpu_model.py."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _random_case(seed):
    rng = np.random.default_rng(seed)
    blstm = rng.random() < 0.75
    N = int(rng.choice([16, 32, 48, 64, 80, 96]))
    P = [None, 16, 32, 48][int(rng.integers(0, 4))]
    cfg = dict(nnet_type="blstm" if blstm else "lstm", input_dim=int(rng.choice([5, 8, 13, 24])), left_context=0,
               right_context=0, num_layers=int(rng.integers(1, 4)), num_neurons=N, num_targets=int(rng.integers(3, 12)),
               use_peepholes=bool(rng.random() < 0.7), dropout_rate=float(rng.choice([1.0, 1.0, 0.8])))
    if P:
        cfg["num_projects"] = P
    if blstm and rng.random() < 0.3:
        cfg["num_experts"] = int(rng.integers(2, 5))
        cfg["moe_temp"] = 2.0
    if not blstm and rng.random() < 0.4:
        cfg["use_bn"] = True
    if rng.random() < 0.25:                                   # residual first layer: D == 2P (blstm) or D == P (lstm)
        out = P if P else N
        cfg["input_dim"] = 2 * out if blstm else out
    B = int(rng.choice([1, 2, 7, 16, 17, 31, 33, 48, 49, 64, 65, 100, 130]))
    T = int(rng.integers(1, 9))
    return cfg, B, T, rng


@pytest.mark.parametrize("dtype", ["fp32", "bf16x3"])
@pytest.mark.parametrize("seed", list(range(24)))
def test_random_shape_vs_oracle(oracle, seed, dtype, monkeypatch):
    """(bf16x3: every product with an activation operand forced onto the split-operand kernels, whatever its shape - ragged
    M / N / K, K < 16, row windows - at the fp32 tolerances.)"""
    from lstm_ctc_amd.nnet import model as model_mod
    from lstm_ctc_amd.nnet.model import Model
    cfg, B, T, rng = _random_case(1000 + seed)
    if dtype == "bf16x3":
        monkeypatch.setattr(model_mod, "X3_FORCE", True)
        cfg["compute_dtype"] = "bf16x3"
    D = cfg["input_dim"]
    seq_len = rng.integers(1, T + 1, size=B).astype(np.int32)
    seq_len[rng.integers(0, B)] = T
    x = rng.normal(size=(B, T, D)).astype(np.float32)
    for b in range(B):
        x[b, seq_len[b]:] = 0
    model = Model(cfg, "cuda", seed=seed)
    params = model.ps.export_tf()
    for k in params:
        if "bias" in k or k.endswith("/beta"):
            params[k] = rng.normal(0, 0.2, size=params[k].shape).astype(np.float32)
    model.ps.load_tf(params)
    p64 = {k: v.astype(np.float64) for k, v in params.items()}
    ref_logits, saved = oracle.forward(p64, cfg, x.astype(np.float64), seq_len, drop_seed=5)
    xt = torch.from_numpy(np.ascontiguousarray(x.transpose(1, 0, 2))).cuda()
    sl = torch.from_numpy(seq_len).cuda()
    got = model.forward(xt, sl, drop_seed=5).cpu().numpy().transpose(1, 0, 2)
    scale = max(np.abs(ref_logits).max(), 1.0)
    assert np.abs(got - ref_logits).max() < 1e-4 * scale, (cfg, B, T, np.abs(got - ref_logits).max())
    dl = rng.normal(size=ref_logits.shape)
    for b in range(B):
        dl[b, seq_len[b]:] = 0
    ref_grads, _ = oracle.backward(p64, cfg, saved, dl)
    model.backward(torch.from_numpy(np.ascontiguousarray(dl.transpose(1, 0, 2)).astype(np.float32)).cuda())
    grads = model.ps.export_tf(grads=True)
    from conftest import check_grad
    for k in sorted(ref_grads):
        check_grad(grads[k], ref_grads[k], "fuzz/%s/B%d/T%d" % (cfg.get("num_neurons"), B, T), k)


@pytest.mark.parametrize("seed", list(range(12)))
def test_random_shape_bf16_routes_agree(seed):
    """compute_dtype = bf16 on random shapes: the shadow-operand route and the converting-loader route round the same
    operands identically, so logits and gradients agree to accumulation order whatever the shape (K % 8 != 0
    products silently take the converting loader in both)."""
    from lstm_ctc_amd.nnet.model import Model
    cfg, B, T, rng = _random_case(2000 + seed)
    cfg["num_neurons"] = int(rng.choice([32, 64, 96]))        # the bf16 step kernels need N % 32 == 0
    cfg["compute_dtype"] = "bf16"
    cfg["dropout_rate"] = 1.0
    D = cfg["input_dim"]
    seq_len = rng.integers(1, T + 1, size=B).astype(np.int32)
    seq_len[rng.integers(0, B)] = T
    x = rng.normal(size=(T, B, D)).astype(np.float32)
    for b in range(B):
        x[seq_len[b]:, b] = 0
    dl = rng.normal(size=(T, B, cfg["num_targets"])).astype(np.float32)
    outs = []
    for shadows in (True, False):
        model = Model(dict(cfg, bf16_shadows=shadows), "cuda", seed=seed)
        logits = model.forward(torch.from_numpy(x).cuda(), torch.from_numpy(seq_len).cuda()).cpu().numpy()
        model.backward(torch.from_numpy(dl).cuda())
        outs.append((logits, model.ps.export_tf(grads=True)))
    (l1, g1), (l2, g2) = outs
    assert np.abs(l1 - l2).max() < 5e-5 * max(np.abs(l2).max(), 1.0), (cfg, B, T)
    for k in g2:
        assert np.abs(g1[k] - g2[k]).max() < 5e-4 * max(np.abs(g2[k]).max(), 1e-3), (cfg, B, T, k)
