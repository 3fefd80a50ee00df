"""Generates tests/golden/blstm_ctc_small.npz from the fp64 CPU oracle (run from the repo root:
``python tests/golden/make_golden.py``).  The reference itself cannot produce vectors (Python 2 + TF 1.8, no
tests), so the fixture freezes the pinned oracle's answers: inputs, TF-layout parameters and
the expected logits, per-utterance CTC losses, CTC gradient, greedy tokens, edit distances and parameter
gradients of one small BiLSTM(+projection, peepholes) model on a ragged batch with repeated labels."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as orc  # noqa: E402

CFG = dict(nnet_type="blstm", input_dim=10, left_context=0, right_context=0, num_layers=2, num_neurons=32,
           num_projects=16, num_targets=9, use_peepholes=True, dropout_rate=1.0)


def main():
    rng = np.random.default_rng(20260102)
    B, T = 6, 24
    seq = np.array([24, 22, 19, 15, 9, 3], np.int32)
    x = rng.normal(size=(B, T, CFG["input_dim"]))
    for b in range(B):
        x[b, seq[b]:] = 0
    labels = np.full((B, 8), -1, np.int64)
    for b, lab in enumerate([[0, 1, 1, 2, 7], [3, 3, 3], [4, 5, 6, 0, 1, 2, 3, 4], [7], [2, 2], [1, 0, 5, 6]]):
        labels[b, :len(lab)] = lab                      # utt 5: L = 4 > T = 3 -> skipped (loss 0, grad 0)
    params = orc.init_params(CFG, seed=42, dtype=np.float64)
    for k in params:
        if "bias" in k or k == "Variable_1":
            params[k] = rng.normal(0, 0.2, size=params[k].shape)
    # the fixture stores float32 inputs/parameters: evaluate the oracle on exactly those values
    x = x.astype(np.float32).astype(np.float64)
    params = {k: v.astype(np.float32).astype(np.float64) for k, v in params.items()}
    out = orc.validation_graph(params, CFG, x, seq, labels, want_grad=True)
    grads, _ = orc.backward(params, CFG, out["saved"], np.ascontiguousarray(out["dlogits"]))
    np.savez_compressed(
        os.path.join(os.path.dirname(os.path.abspath(__file__)), "blstm_ctc_small.npz"),
        x=x.astype(np.float32), seq=seq, labels=labels,
        **{"param/" + k: v.astype(np.float32) for k, v in params.items()},
        **{"grad/" + k: v for k, v in grads.items()},
        logits=out["logits"], loss_per_utt=out["loss_per_utt"], dlogits=out["dlogits"],
        tokens=out["tokens"], token_len=out["token_len"], eval=np.float64(out["eval"]), size=np.int64(out["size"]))
    print("loss per utt:", out["loss_per_utt"], "eval:", out["eval"])


if __name__ == "__main__":
    main()
