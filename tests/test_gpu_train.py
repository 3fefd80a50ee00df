"""GPU parity tests, train-step level and CLI level.

* CTCGraph.step (fwd + CTC + greedy/edit distance + bwd + L2/clip/optimizer) vs the oracle's train_step over
  several steps: losses, decoded tokens (bit-exact), gradient norm, updated parameters.
* bin/nnet-init -> nnet-train -> nnet-validate -> nnet-forward on synthetic tfrecords: exit codes, the
  machine-parsed log lines, checkpoint round trip, Kaldi ark output vs the oracle forward.
"""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _batch(rng, B, T, D, V, zero_len=False):
    seq = np.sort(rng.integers(T // 2, T + 1, size=B))[::-1].astype(np.int32).copy()
    seq[0] = T
    x = rng.normal(size=(B, T, D)).astype(np.float32)
    labels = np.full((B, 6), -1, np.int64)
    for b in range(B):
        x[b, seq[b]:] = 0
        n = int(rng.integers(1, 6))
        labels[b, :n] = rng.integers(0, V - 1, size=n)
    return {"nnet_input": x, "sequence_length": seq, "nnet_target": labels}


@pytest.mark.parametrize("optimizer,cfgkw", [("adam", {}), ("sgd", {}), ("momentum", dict(num_experts=3)),
                                             ("adam", dict(nnet_type="lstm", num_projects=16)),
                                             ("sgd", dict(uniform_label_sm=0.2)),
                                             ("adam", dict(nnet_type="lstm", num_projects=16, use_bn=True)),
                                             ("sgd", dict(nnet_type="lstm", num_projects=12, use_bn=True)),
                                             ("adam", dict(nnet_type="cudnnlstm", num_projects=32, dropout_rate=0.5))])
def test_train_steps_vs_oracle(oracle, optimizer, cfgkw):
    from lstm_ctc_amd.nnet.graph import create_graph_for_training_ctc
    cfg = dict(nnet_type="blstm", input_dim=12, left_context=0, right_context=0, num_layers=2, num_neurons=32,
               num_projects=16, num_targets=8, use_peepholes=True, dropout_rate=1.0)
    cfg.update(cfgkw)
    rng = np.random.default_rng(3)
    graph = create_graph_for_training_ctc(None, cfg, learn_rate=1e-2, clip_norm=5.0, optimizer=optimizer, seed=11)
    params = {k: v.copy() for k, v in graph.model.ps.export_tf().items()}
    state = {}
    for step in range(3):
        batch = _batch(rng, 5, 14, 12, 8)
        out = graph.step(batch, fetch_eval=True)
        ref = oracle.train_step(params, cfg, batch["nnet_input"], batch["sequence_length"], batch["nnet_target"],
                                state, optimizer=optimizer, lr=1e-2, clip_norm=5.0, l2=1e-5)
        assert out["size"] == ref["size"]
        assert abs(out["eval_loss"] - ref["eval_loss"]) / ref["eval_loss"] < 1e-4        # north-star tolerance
        assert abs(out["loss"] - ref["loss"]) / abs(ref["loss"]) < 1e-4                  # incl. label-smoothing reg
        assert out["eval"] == ref["eval"]                                                # edit distance: exact
        tok, n = out["decoded"]
        assert np.array_equal(n, ref["token_len"])
        for b in range(len(n)):
            assert np.array_equal(tok[b, :n[b]], ref["tokens"][b, :n[b]])                # greedy tokens: bit-exact
        assert abs(out["grad_norm"] - ref["grad_norm"]) / ref["grad_norm"] < 2e-3
        got = graph.model.ps.export_tf()
        for k in params:
            assert np.abs(got[k] - params[k]).max() < 2e-4 * max(1.0, np.abs(params[k]).max()), (step, k)


def test_checkpoint_roundtrip(tmp_path):
    from lstm_ctc_amd.nnet.graph import create_graph_for_validation_ctc
    cfg = dict(nnet_type="blstm", input_dim=8, left_context=0, right_context=0, num_layers=1, num_neurons=16,
               num_projects=16, num_targets=5, use_peepholes=True, dropout_rate=1.0)
    g1 = create_graph_for_validation_ctc(None, cfg, seed=1)
    g2 = create_graph_for_validation_ctc(None, cfg, seed=2)
    path = str(tmp_path / "nnet.0")
    g1.save(path)
    assert os.path.exists(path)                 # single file at exactly the prefix the scripts pass around
    g2.restore(path)
    assert torch.equal(g1.model.ps.flat, g2.model.ps.flat)


def _write_corpus(tmp_path, rng, n, D, V):
    from lstm_ctc_amd.nnet import write_tfrecord
    lines, utts = [], []
    for i in range(n):
        T = int(rng.integers(20, 41))
        x = rng.normal(size=(T, D)).astype(np.float32)
        y = rng.integers(0, V - 1, size=int(rng.integers(1, 5)))
        path = str(tmp_path / ("utt%03d.tfrecords" % i))
        write_tfrecord(path, x, y)
        lines.append((T, "utt%03d %d %d 1 %s" % (i, T, D, path)))
        utts.append((x, y))
    order = np.argsort([l[0] for l in lines], kind="stable")          # recipes sort by length
    scp = tmp_path / "tfrecords.scp"
    scp.write_text("\n".join(lines[i][1] for i in order) + "\n")
    return str(scp), utts


def _run(cli, *args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bin", cli)] + list(args), capture_output=True, timeout=600)
    return r.returncode, r.stdout, r.stderr.decode()


CLI_CASES = {
    # the recipes' shape in miniature: spliced + subsampled input, 2 x BiLSTM-P, dropout
    "blstm": (6, 9, "nnet_type = blstm\ninput_dim = 6\nleft_context = 1\nright_context = 1\nsubsample = 2\n"
                    "num_layers = 2\nnum_neurons = 32\nnum_projects = 16\nnum_targets = 9\nuse_peepholes = true\n"
                    "dropout_rate = 0.9\n", 2e-4),
    # BASELINE config c1 (plumbing): 1 x uniLSTM-256, 40-d fbank, 72 targets (P = N as in run_wsj_phn.sh:17,26)
    "c1_unilstm256": (40, 72, "nnet_type = lstm\ninput_dim = 40\nleft_context = 1\nright_context = 1\nsubsample = 2\n"
                              "num_layers = 1\nnum_neurons = 256\nnum_projects = 256\nnum_targets = 72\n"
                              "dropout_rate = 1.0\n", 5e-4),
    # uni-LSTM with batch normalisation: moving averages travel in the checkpoint and are used by nnet-forward
    "unilstm_bn": (6, 9, "nnet_type = lstm\ninput_dim = 6\nleft_context = 0\nright_context = 0\nsubsample = 1\n"
                         "num_layers = 2\nnum_neurons = 32\nnum_projects = 6\nnum_targets = 9\nuse_bn = true\n"
                         "dropout_rate = 1.0\n", 5e-4),
    # nnet_type = cudnnlstm (nnet/lstm.py:26-122): plain LSTM cells, keep-prob / peephole flag / num_projects inert
    "cudnnlstm": (6, 9, "nnet_type = cudnnlstm\ninput_dim = 6\nleft_context = 1\nright_context = 1\nsubsample = 2\n"
                        "num_layers = 2\nnum_neurons = 32\nnum_projects = 32\nnum_targets = 9\nuse_peepholes = true\n"
                        "dropout_rate = 0.9\n", 2e-4),
    # BASELINE config c5 in miniature: bf16 GEMM operands (extension key), compared at bf16 operand precision
    "bf16": (6, 9, "nnet_type = blstm\ninput_dim = 6\nleft_context = 1\nright_context = 1\nsubsample = 2\n"
                   "num_layers = 2\nnum_neurons = 32\nnum_projects = 16\nnum_targets = 9\nuse_peepholes = true\n"
                   "dropout_rate = 1.0\ncompute_dtype = bf16\n", 1e-1),
}


@pytest.mark.parametrize("case", sorted(CLI_CASES))
def test_cli_pipeline_end_to_end(tmp_path, oracle, case):
    rng = np.random.default_rng(7)
    D, V, config_text, atol = CLI_CASES[case]
    scp, utts = _write_corpus(tmp_path, rng, 10, D, V)
    config = tmp_path / "nnet.config"
    config.write_text(config_text)
    d = str(tmp_path)
    rc, _, err = _run("nnet-init.py", "--objective=ctc", "--evaluate=true", "--batch-size", "4", scp, str(config), d + "/nnet.0")
    assert rc == 0, err
    cv0 = [l for l in err.split("\n") if l.startswith("INFO:tensorflow:cv_loss")]
    assert len(cv0) == 1 and any(l.startswith("INFO:tensorflow:cv_eval") for l in err.split("\n"))
    rc, _, err = _run("nnet-train.py", "--objective=ctc", "--learn-rate=0.01", "--optimizer=adam", "--seed=1",
                      "--shuffle=false", "--batch-size", "4", "--report-interval=1", scp, str(config),
                      d + "/nnet.0", d + "/nnet.1")
    assert rc == 0, err
    tr = [l for l in err.split("\n") if l.startswith("INFO:tensorflow:tr_loss")]
    assert len(tr) == 1 and np.isfinite(float(tr[0].split()[-1]))
    assert any(l.startswith("INFO:tensorflow:step = 1, batch_size = ") for l in err.split("\n"))
    rc, _, err = _run("nnet-validate.py", "--objective=ctc", "--evaluate=true", "--batch-size", "4", scp, str(config), d + "/nnet.1")
    assert rc == 0, err
    cv1 = float([l for l in err.split("\n") if l.startswith("INFO:tensorflow:cv_loss")][0].split()[-1])
    if case != "unilstm_bn":      # (with batch norm the CV pass runs on moving averages that 3 steps barely moved)
        assert cv1 < float(cv0[0].split()[-1])                  # one epoch of adam on 10 utterances lowers the CV loss
    assert np.isfinite(cv1)
    counts = tmp_path / "label.counts"
    counts.write_text("[ " + " ".join(str(10 + i) for i in range(V)) + " ]\n")
    ark = d + "/post.ark"
    rc, _, err = _run("nnet-forward.py", "--apply-log=true", "--class-prior=" + str(counts), "--batch-utts", "3",
                      scp, str(config), d + "/nnet.1", "ark:" + ark)
    assert rc == 0, err
    from lstm_ctc_amd.kaldi_io import read_float_matrix_ark
    from lstm_ctc_amd.nnet import get_class_prior, parse_config
    from lstm_ctc_amd.nnet.tfrecord import splice, subsample
    from safetensors.numpy import load_file
    post = read_float_matrix_ark(ark)
    assert sorted(post) == ["utt%03d" % i for i in range(10)]
    cfg = parse_config(str(config))
    cfg["is_training"] = False
    params = {k: v.astype(np.float64) for k, v in load_file(d + "/nnet.1").items()}
    prior = get_class_prior(str(counts))
    lc, rc_, ss = int(cfg.get("left_context") or 0), int(cfg.get("right_context") or 0), int(cfg.get("subsample") or 1)
    for i in (0, 4, 9):                                            # oracle forward, one utterance at a time (B = 1)
        x = subsample(splice(utts[i][0], lc, rc_), ss).astype(np.float64)[None]
        logits, _ = oracle.forward(params, cfg, x, np.array([x.shape[1]], np.int32))
        z = logits[0]
        ref = z - z.max(1, keepdims=True) - np.log(np.exp(z - z.max(1, keepdims=True)).sum(1, keepdims=True)) - prior
        assert post["utt%03d" % i].shape == ref.shape
        np.testing.assert_allclose(post["utt%03d" % i], ref, atol=atol)
