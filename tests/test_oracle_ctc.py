"""Pins the oracle's CTC / greedy / edit-distance restatement (CPU only).

Anchors (the reference has no tests — SURVEY.md §4/§8c):
  1. TF-upstream known answers (tests/golden/ctc_tf_known_answers.json).
  2. torch-CPU F.ctc_loss cross-check (loss + gradient) on ragged random batches with
     adjacent repeats, L > T (TF: skipped, loss 0 / grad 0) and infeasible-repeat cases.
  3. Hand-computed greedy / edit-distance cases.
"""
import json
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_tf_known_answers(oracle):
    kat = json.load(open(os.path.join(GOLD, "ctc_tf_known_answers.json")))
    T, V = kat["T"], kat["V"]
    B = len(kat["utts"])
    for dt, tol in ((np.float32, 2e-6), (np.float64, 2e-6)):
        logits = np.zeros((T, B, V), dt)
        flat, offs = [], [0]
        for b, u in enumerate(kat["utts"]):
            logits[:, b, :] = np.log(np.asarray(u["probs"], np.float64)).astype(dt)
            flat += u["labels"]
            offs.append(len(flat))
        loss, grad, bad = oracle.ctc_loss(logits, flat, offs, [T] * B)
        assert bad == 0
        for b, u in enumerate(kat["utts"]):
            assert abs(loss[b] - u["loss"]) / u["loss"] < tol + 2e-6, (dt, b, loss[b])
        # gradient rows sum to zero (softmax - posterior, both sum to 1)
        assert np.abs(grad.sum(axis=2)).max() < 1e-5
        # TF's gradient_log_prob_{0,1} (printed to 6 significant digits)
        for b, u in enumerate(kat["utts"]):
            assert np.abs(grad[:, b, :] - np.asarray(u["grad"])).max() < 2e-6, (dt, b)


def test_tf_greedy_known_answers(oracle):
    kat = json.load(open(os.path.join(GOLD, "ctc_tf_greedy_known_answers.json")))
    T, V, B = kat["T"], kat["V"], len(kat["utts"])
    with np.errstate(divide="ignore"):
        logits = np.stack([np.log(np.asarray(u["probs"], np.float32)) for u in kat["utts"]], axis=1)   # [T,B,V], -inf
    tok, n, nsl = oracle.ctc_greedy(logits, [u["seq_len"] for u in kat["utts"]])
    for b, u in enumerate(kat["utts"]):
        assert list(tok[b, :n[b]]) == u["decoded"]
        assert abs(nsl[b] - u["neg_sum_logits"]) < 1e-6


def _random_case(rng, B, T, V, Lmax, force_repeat=True):
    seq_len = rng.integers(max(2, T // 2), T + 1, size=B)
    seq_len[0] = T
    labels = []
    for b in range(B):
        L = int(rng.integers(1, min(Lmax, seq_len[b] // 2) + 1))
        lab = rng.integers(0, V - 1, size=L)
        if force_repeat and b % 2 == 0 and L >= 2:
            lab[1] = lab[0]
        labels.append(lab)
    return seq_len, labels


@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_vs_torch_ctc(oracle, dt):
    rng = np.random.default_rng(0)
    B, T, V = 7, 40, 9
    seq_len, labels = _random_case(rng, B, T, V, 12)
    logits = rng.normal(0, 2.0, size=(T, B, V)).astype(dt)
    flat = np.concatenate(labels).astype(np.int32)
    offs = np.concatenate([[0], np.cumsum([len(l) for l in labels])]).astype(np.int32)
    loss, grad, bad = oracle.ctc_loss(logits, flat, offs, seq_len)
    assert bad == 0
    x = torch.tensor(logits, dtype=torch.float64, requires_grad=True)
    lp = torch.log_softmax(x, dim=2)
    tl = torch.nn.functional.ctc_loss(lp, torch.tensor(flat, dtype=torch.long), torch.tensor(seq_len),
                                      torch.tensor([len(l) for l in labels]), blank=V - 1, reduction="none")
    tl.sum().backward()
    tol = 1e-5 if dt == np.float32 else 1e-10
    np.testing.assert_allclose(loss, tl.detach().numpy(), rtol=tol)
    g = x.grad.numpy()
    for b in range(B):      # torch leaves padded frames at 0 as well
        np.testing.assert_allclose(grad[:seq_len[b], b], g[:seq_len[b], b], atol=tol * 10, rtol=tol * 10)
        assert np.all(grad[seq_len[b]:, b] == 0)


def test_longer_outputs_skipped_and_infeasible(oracle):
    V, T = 5, 6
    rng = np.random.default_rng(1)
    logits = rng.normal(size=(T, 3, V)).astype(np.float32)
    labels = [[0, 1, 2, 3, 0, 1, 2], [1, 1, 1], [2]]        # utt0: L=7 > T=6 ; utt1: L=3, T_b=3 with repeats -> infeasible
    flat = np.concatenate(labels).astype(np.int32)
    offs = np.array([0, 7, 10, 11], np.int32)
    seq_len = np.array([6, 3, 6], np.int32)
    loss, grad, bad = oracle.ctc_loss(logits, flat, offs, seq_len)
    assert loss[0] == 0 and np.all(grad[:, 0] == 0)         # ignore_longer_outputs_than_inputs=True
    assert np.isinf(loss[1]) and bad == 1                   # "No valid path found": grad = softmax
    sm = np.exp(logits[:3, 1] - logits[:3, 1].max(-1, keepdims=True))
    sm /= sm.sum(-1, keepdims=True)
    np.testing.assert_allclose(grad[:3, 1], sm, rtol=1e-6)
    assert np.all(grad[3:, 1] == 0)
    assert np.isfinite(loss[2]) and loss[2] > 0


def test_ctc_grad_finite_difference(oracle):
    rng = np.random.default_rng(2)
    T, B, V = 9, 2, 5
    logits = rng.normal(size=(T, B, V))
    flat = np.array([0, 0, 1, 3], np.int32)
    offs = np.array([0, 3, 4], np.int32)
    seq_len = np.array([9, 7], np.int32)
    loss, grad, _ = oracle.ctc_loss(logits, flat, offs, seq_len)
    eps = 1e-6
    for (t, b, k) in [(0, 0, 0), (3, 0, 4), (5, 1, 3), (8, 0, 1), (8, 1, 2)]:
        xp, xm = logits.copy(), logits.copy()
        xp[t, b, k] += eps
        xm[t, b, k] -= eps
        fd = (oracle.ctc_loss(xp, flat, offs, seq_len, False)[0].sum()
              - oracle.ctc_loss(xm, flat, offs, seq_len, False)[0].sum()) / (2 * eps)
        assert abs(fd - grad[t, b, k]) < 1e-7, (t, b, k, fd, grad[t, b, k])


def test_greedy_hand_cases(oracle):
    V = 4  # blank = 3
    def onehot(seq):
        x = np.full((len(seq), 1, V), -1.0, np.float32)
        for t, k in enumerate(seq):
            x[t, 0, k] = 1.0
        return x
    tok, n, nsl = oracle.ctc_greedy(onehot([0, 0, 3, 0, 1, 1, 3, 3, 2]), [9])
    assert list(tok[0, :n[0]]) == [0, 0, 1, 2]              # a,a -> a ; a,blank,a -> a,a
    assert abs(nsl[0] + 9.0) < 1e-6                         # -sum of max raw logits
    x = np.zeros((3, 1, V), np.float32)                     # all ties -> argmax = lowest index 0
    tok, n, _ = oracle.ctc_greedy(x, [3])
    assert list(tok[0, :n[0]]) == [0]
    x[:, 0, 3] = 1.0                                        # all blank -> empty
    tok, n, _ = oracle.ctc_greedy(x, [3])
    assert n[0] == 0
    tok, n, _ = oracle.ctc_greedy(onehot([0, 1, 2, 0]), [2])  # only seq_len frames are decoded
    assert list(tok[0, :n[0]]) == [0, 1]


def test_edit_distance_hand_cases(oracle):
    hyp = np.array([[1, 2, 3, 0], [0, 0, 0, 0], [5, 6, 0, 0], [1, 2, 3, 4]], np.int32)
    hyp_len = [3, 0, 2, 4]
    truth = [[1, 3], [7, 8, 9], [5, 6], [4, 3, 2, 1]]
    flat = np.concatenate(truth).astype(np.int32)
    offs = np.array([0, 2, 5, 7, 11], np.int32)
    d = oracle.edit_distance(hyp, hyp_len, flat, offs)
    assert list(d) == [1, 3, 0, 4]                          # empty hyp -> len(truth)
    # cross-check vs a tiny pure-python Levenshtein
    rng = np.random.default_rng(3)
    for _ in range(20):
        a = rng.integers(0, 4, size=rng.integers(0, 8))
        b = rng.integers(0, 4, size=rng.integers(0, 8))
        D = np.zeros((len(a) + 1, len(b) + 1), int)
        D[:, 0] = np.arange(len(a) + 1)
        D[0, :] = np.arange(len(b) + 1)
        for i in range(1, len(a) + 1):
            for j in range(1, len(b) + 1):
                D[i, j] = min(D[i - 1, j] + 1, D[i, j - 1] + 1, D[i - 1, j - 1] + (a[i - 1] != b[j - 1]))
        h = np.zeros((1, 8), np.int32)
        h[0, :len(a)] = a
        got = oracle.edit_distance(h, [len(a)], b.astype(np.int32), np.array([0, len(b)], np.int32))
        assert got[0] == D[-1, -1]
