"""Committed golden vectors (tests/golden/blstm_ctc_small.npz, made by tests/golden/make_golden.py from the
pinned fp64 oracle): the oracle must keep reproducing them (CPU), and the HIP path must match them (GPU)."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from make_golden import CFG  # noqa: E402


def _load():
    z = np.load(os.path.join(HERE, "golden", "blstm_ctc_small.npz"))
    params = {k[6:]: z[k] for k in z.files if k.startswith("param/")}
    grads = {k[5:]: z[k] for k in z.files if k.startswith("grad/")}
    return z, params, grads


@pytest.mark.parametrize("dt,tol", [(np.float64, 1e-9), (np.float32, 2e-4)])
def test_oracle_reproduces_golden(oracle, dt, tol):
    z, params, grads = _load()
    p = {k: v.astype(dt) for k, v in params.items()}
    out = oracle.validation_graph(p, CFG, z["x"].astype(dt), z["seq"], z["labels"], want_grad=True)
    np.testing.assert_allclose(out["logits"], z["logits"], atol=tol * 10)
    np.testing.assert_allclose(out["loss_per_utt"], z["loss_per_utt"], rtol=tol, atol=tol)
    assert out["eval"] == float(z["eval"]) and out["size"] == int(z["size"])
    assert np.array_equal(out["token_len"], z["token_len"])
    assert z["loss_per_utt"][5] == 0 and np.all(z["dlogits"][5] == 0)       # L > T utterance is skipped


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["fp32", "bf16x3"])
def test_hip_path_matches_golden(dtype, monkeypatch):
    import torch
    from lstm_ctc_amd import ops
    from lstm_ctc_amd.nnet import model as model_mod
    from lstm_ctc_amd.nnet.graph import flatten_labels
    from lstm_ctc_amd.nnet.model import Model
    z, params, grads = _load()
    if dtype == "bf16x3":                      # the split-operand products, forced at this size: same vectors, same tolerances
        monkeypatch.setattr(model_mod, "X3_FORCE", True)
    model = Model(dict(CFG, compute_dtype=dtype), "cuda", seed=0)
    assert model.x3 == (dtype == "bf16x3")
    model.ps.load_tf(params)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    seq = d(z["seq"])
    logits = model.forward(d(z["x"].transpose(1, 0, 2)), seq)
    ref_logits = z["logits"].transpose(1, 0, 2)
    assert np.abs(logits.cpu().numpy() - ref_logits).max() < 1e-4 * max(1.0, np.abs(ref_logits).max())
    flat, offs, maxlen = flatten_labels(z["labels"])
    loss, grad = ops.ctc_loss(logits, d(flat), d(offs), seq, maxlen)
    np.testing.assert_allclose(loss.cpu().numpy(), z["loss_per_utt"], rtol=1e-4, atol=1e-6)
    assert np.abs(grad.cpu().numpy() - z["dlogits"].transpose(1, 0, 2)).max() < 2e-4
    tok, n = ops.ctc_greedy(logits, seq)
    tok, n = tok.cpu().numpy(), n.cpu().numpy()
    assert np.array_equal(n, z["token_len"])
    for b in range(len(n)):
        assert np.array_equal(tok[b, :n[b]], z["tokens"][b, :n[b]])
    assert float(ops.edit_distance_host(tok, n, flat, offs).sum()) == float(z["eval"])
    model.backward(grad)
    got = model.ps.export_tf(grads=True)
    from conftest import check_grad
    for k in grads:
        check_grad(got[k], grads[k], "golden/" + str(dtype), k)
