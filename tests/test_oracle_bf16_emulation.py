"""The float64 emulation of the bf16 path (oracle/bf16_emulation.py) restates the FOLDED algebra the product
evaluates (x.Kx hoisted, R = proj.Kh, projection batched); with the bf16 rounding switched off it must reproduce the
fp64 oracle's BiLSTM-P forward and backward, which follow the reference's concat-matmul form (nnet/bilstm.py:125-250)."""
import numpy as np


def test_emulation_without_rounding_equals_oracle(oracle, monkeypatch):
    from oracle import bf16_emulation as emu
    cfg = dict(nnet_type="blstm", input_dim=7, left_context=0, right_context=0, num_layers=3, num_neurons=16,
               num_projects=8, num_targets=6, use_peepholes=True, dropout_rate=1.0)
    rng = np.random.default_rng(0)
    p = oracle.init_params(cfg, seed=1)
    for k in p:
        if "bias" in k or k == "Variable_1":
            p[k] = rng.normal(0, 0.2, size=p[k].shape).astype(np.float32)
    p64 = {k: v.astype(np.float64) for k, v in p.items()}
    B, T = 5, 9
    x = rng.normal(size=(B, T, 7))
    sl = np.array([9, 7, 7, 4, 1], np.int32)
    for b in range(B):
        x[b, sl[b]:] = 0
    lg, sv = oracle.forward(p64, cfg, x, sl)
    dl = rng.normal(size=lg.shape)
    for b in range(B):
        dl[b, sl[b]:] = 0
    g, _ = oracle.backward(p64, cfg, sv, dl)
    lg_bf, _ = emu.forward(p64, cfg, x, sl)
    assert 1e-5 < np.abs(lg_bf - lg).max() < 5e-2          # the rounding is on by default, and is a bf16-sized effect
    monkeypatch.setattr(emu, "_bf", lambda a: np.asarray(a, np.float64))
    lg2, sv2 = emu.forward(p64, cfg, x, sl)
    g2 = emu.backward(p64, cfg, sv2, dl)
    assert np.abs(lg - lg2).max() < 1e-6                   # R and dR pass through float32, as in the product
    assert set(g) == set(g2)
    for k in g:
        assert np.abs(g[k] - g2[k]).max() < 1e-6 * max(1.0, np.abs(g[k]).max()), k
