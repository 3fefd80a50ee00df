"""GPU tests added in round 2: prior label smoothing, -inf logits in the CTC loss, the persistent-recurrence
failure path (forced timeout -> NaN + sticky status -> in-process re-run on the launch train), the reference-named
create_logits callable, invalid labels, deterministic reductions, and the multi-process paths: bench.py under
torch.distributed.run on real RCCL, and CTCGraph itself (not the oracle) in two processes sharing the GPU."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cfg(**kw):
    cfg = dict(nnet_type="blstm", input_dim=12, left_context=0, right_context=0, num_layers=2, num_neurons=32,
               num_projects=16, num_targets=8, use_peepholes=True, dropout_rate=1.0)
    cfg.update(kw)
    return cfg


def _batch(rng, B, T, D, V, Lmax=6):
    seq = np.sort(rng.integers(T // 2, T + 1, size=B))[::-1].astype(np.int32).copy()
    seq[0] = T
    x = rng.normal(size=(B, T, D)).astype(np.float32)
    labels = np.full((B, Lmax), -1, np.int64)
    for b in range(B):
        x[b, seq[b]:] = 0
        n = int(rng.integers(1, Lmax))
        labels[b, :n] = rng.integers(0, V - 1, size=n)
    return {"nnet_input": x, "sequence_length": seq, "nnet_target": labels}


# ---------------------------------------------------------------------------------------------- label smoothing, prior
@pytest.mark.parametrize("optimizer", ["sgd", "adam"])
def test_prior_label_smoothing_train_steps_vs_oracle(oracle, tmp_path, optimizer):
    """nnet/bilstm.py:262-269: kl = p * (log p - prior) with the rotated log prior of nnet/class_prior.py, summed over
    all [B,T,V] and added to the CTC loss (graph.py:120-133) - forward value, gradient and three optimizer steps."""
    from lstm_ctc_amd.nnet import get_class_prior
    from lstm_ctc_amd.nnet.graph import create_graph_for_training_ctc
    counts = tmp_path / "label.counts"
    counts.write_text("[ 900 40 0 25 310 7 55 120 ]\n")           # blank first, one zero count (-> -1e10)
    cfg = _cfg(prior_label_sm=0.15, uniform_label_sm=0, prior_label_path=str(counts))
    prior = get_class_prior(str(counts))
    assert prior.shape == (8,) and prior[1] == np.float32(-1e10)
    rng = np.random.default_rng(8)
    graph = create_graph_for_training_ctc(None, cfg, learn_rate=1e-2, clip_norm=5.0, optimizer=optimizer, seed=4)
    assert graph.sm_weight == 0.15 and graph.sm_logq is not None
    params = {k: v.copy() for k, v in graph.model.ps.export_tf().items()}
    state = {}
    for step in range(3):
        batch = _batch(rng, 5, 13, 12, 8)
        out = graph.step(batch, fetch_eval=True)
        p64 = {k: v.astype(np.float64) for k, v in params.items()}
        ref = oracle.validation_graph(p64, cfg, batch["nnet_input"].astype(np.float64), batch["sequence_length"],
                                      batch["nnet_target"], want_grad=True, class_prior=prior.astype(np.float64))
        assert ref["loss"] != ref["eval_loss"]                     # the regulariser is really on
        assert abs(out["eval_loss"] - ref["eval_loss"]) / ref["eval_loss"] < 1e-4
        assert abs(out["loss"] - ref["loss"]) / abs(ref["loss"]) < 1e-4
        grads, _ = oracle.backward(p64, cfg, ref["saved"], np.ascontiguousarray(ref["dlogits"]))
        clipped, norm = oracle.l2_and_clip(params, {k: v.astype(np.float32) for k, v in grads.items()}, 5.0, 1e-5)
        oracle.apply_optimizer(optimizer, params, clipped, state, 1e-2)
        assert abs(out["grad_norm"] - norm) / norm < 2e-3
        got = graph.model.ps.export_tf()
        for k in params:
            assert np.abs(got[k] - params[k]).max() < 2e-4 * max(1.0, np.abs(params[k]).max()), (step, k)


def test_label_smoothing_kernel_prior_vs_oracle(oracle):
    """lc_label_smoothing(log_q != NULL) alone: value and gradient on random logits incl. a -1e10 prior entry."""
    from lstm_ctc_amd import ops
    rng = np.random.default_rng(2)
    rows, V = 37, 44
    logits = rng.normal(0, 2.0, size=(rows, V)).astype(np.float32)
    q = rng.dirichlet(np.ones(V))
    logq = np.log(q).astype(np.float32)
    logq[3] = -1e10
    base = rng.normal(size=(rows, V)).astype(np.float32)
    d = torch.from_numpy(base.copy()).cuda()
    acc = ops.label_smoothing(torch.from_numpy(logits).cuda(), 0.3, torch.from_numpy(logq).cuda(), d)
    cfg = dict(prior_label_sm=0.3)
    rl, rg = oracle.label_smoothing(logits.astype(np.float64)[None], cfg, logq.astype(np.float64))
    assert abs(float(acc.item()) - rl) / abs(rl) < 1e-5
    got = d.cpu().numpy() - base
    assert np.abs(got - rg[0]).max() < 1e-4 * max(1.0, np.abs(rg).max())


# ---------------------------------------------------------------------------------------------- CTC with -inf logits
@pytest.mark.parametrize("lse2", [None, 1])
def test_ctc_loss_with_minus_inf_logits(oracle, lse2):
    """A class masked with -inf (zero probability): finite loss when a path avoids it, +inf ("no valid path",
    gradient = softmax) when the labelling needs it.  The scan's "log zero" is a finite sentinel internally."""
    from lstm_ctc_amd import ops
    rng = np.random.default_rng(5)
    T, B, V = 9, 4, 6
    logits = rng.normal(size=(T, B, V)).astype(np.float32)
    logits[:, 0, 2] = -np.inf          # utt 0 never uses class 2: loss stays finite
    logits[:, 1, 1] = -np.inf          # utt 1 needs class 1: no valid path
    logits[3, 2, 5] = -np.inf          # utt 2: blank impossible at one frame only (paths through labels remain)
    logits[:2, 3, :4] = -np.inf        # utt 3: only classes 4 / blank possible in the first two frames
    labels = [np.array([0, 1, 3]), np.array([1, 0]), np.array([2, 2, 4]), np.array([4, 0])]
    flat = np.concatenate(labels).astype(np.int32)
    offs = np.concatenate([[0], np.cumsum([len(l) for l in labels])]).astype(np.int32)
    seq = np.array([9, 9, 8, 7], np.int32)
    ref_loss, ref_grad, _ = oracle.ctc_loss(logits.astype(np.float64), flat, offs, seq)
    dev = lambda a: torch.from_numpy(a).cuda()
    ops.set_option("ctc_lse2", lse2)           # the frame statistics in phase 1 / in phase 2 (calls with >= 512 utterances)
    try:
        loss, grad = ops.ctc_loss(dev(logits), dev(flat), dev(offs), dev(seq), 3)
    finally:
        ops.set_option("ctc_lse2", None)
    loss, grad = loss.cpu().numpy(), grad.cpu().numpy()
    assert np.isinf(ref_loss[1]) and np.isinf(loss[1]) and loss[1] > 0
    fin = np.isfinite(ref_loss)
    assert fin.sum() == 3 and np.array_equal(np.isfinite(loss), fin)
    np.testing.assert_allclose(loss[fin], ref_loss[fin], rtol=1e-4)
    assert np.isfinite(grad).all()
    assert np.abs(grad - ref_grad).max() < 2e-4


def test_ctc_rejects_labels_outside_the_alphabet():
    """tf.nn.ctc_loss raises InvalidArgument for a label >= num_classes - 1 (the blank is not a label) or < 0: the
    host validates before anything reaches the GPU; the kernel itself flags such an utterance with a NaN loss and
    never indexes out of bounds."""
    from lstm_ctc_amd import ops
    from lstm_ctc_amd.nnet.graph import create_graph_for_validation_ctc
    cfg = _cfg()
    graph = create_graph_for_validation_ctc(None, cfg, seed=1)
    rng = np.random.default_rng(0)
    batch = _batch(rng, 3, 10, 12, 8)
    batch["nnet_target"][1, 0] = 7                                   # == blank
    with pytest.raises(ValueError):
        graph.step(batch)
    batch["nnet_target"][1, 0] = 8
    with pytest.raises(ValueError):
        graph.step(batch)
    T, B, V = 6, 2, 5
    dev = lambda a: torch.from_numpy(a).cuda()
    logits = rng.normal(size=(T, B, V)).astype(np.float32)
    flat = np.array([1, 9, 2, 0], np.int32)                          # utt 0 holds label 9 >= V
    offs = np.array([0, 2, 4], np.int32)
    loss, grad = ops.ctc_loss(dev(logits), dev(flat), dev(offs), dev(np.array([6, 6], np.int32)), 2)
    loss = loss.cpu().numpy()
    assert np.isnan(loss[0]) and np.isfinite(loss[1])
    assert np.isfinite(grad.cpu().numpy()[:, 1]).all()


# ---------------------------------------------------------------------------------------------- persistent failure path
def test_persistent_timeout_is_loud_and_recovered_in_process(oracle, monkeypatch, capfd):
    """LC_LSTM_SPIN_LIMIT=0 makes every wait of the persistent recurrence give up at once.  At the C ABI that must be
    LOUD (every output row NaN, sticky status word set); at graph level the step must be re-run in this process on the
    launch train - same losses and parameters as the oracle, one log line, parameters untouched by the failed try."""
    from lstm_ctc_amd import ops
    from lstm_ctc_amd.nnet.graph import create_graph_for_training_ctc
    cfg = _cfg(num_neurons=64, num_projects=32)
    rng = np.random.default_rng(12)
    batch = _batch(rng, 6, 15, 12, 8)
    # --- C ABI level
    T, B, N = 12, 6, 64
    seq = torch.full((B,), T, dtype=torch.int32, device="cuda")
    mk = lambda *s: (torch.randn(*s, device="cuda") * 0.3)
    d = dict(zx=mk(T * B, 4 * N), R=mk(N, 4 * N) * 0.2, w_f=mk(N), w_i=mk(N), w_o=mk(N),
             cs=torch.zeros(T * B, N, device="cuda"), hs=torch.zeros(T * B, N, device="cuda"), reverse=0)
    ops.lstm_status("cuda").zero_()
    ops.lstm_fwd([dict(d, zx=d["zx"].clone())], seq, T, B, N, 1.0)
    assert ops.last_lstm_schedule()["kind"] == "persistent_f32"
    assert int(ops.lstm_status("cuda").item()) == 0
    monkeypatch.setenv("LC_LSTM_SPIN_LIMIT", "0")
    ops.lstm_fwd([d], seq, T, B, N, 1.0)
    torch.cuda.synchronize()
    assert int(ops.lstm_status("cuda").item()) != 0                  # sticky: survives until the caller clears it
    assert torch.isnan(d["hs"]).all()                                # every row, not one poisoned element
    monkeypatch.delenv("LC_LSTM_SPIN_LIMIT")
    # --- graph level
    graph = create_graph_for_training_ctc(None, cfg, learn_rate=1e-2, clip_norm=5.0, optimizer="adam", seed=9)
    params = {k: v.copy() for k, v in graph.model.ps.export_tf().items()}
    state = {}
    monkeypatch.setenv("LC_LSTM_SPIN_LIMIT", "0")
    out = graph.step(batch, fetch_eval=True)
    assert graph.persist_fallbacks == 1
    assert "re-running the step with the per-step launch train" in capfd.readouterr().err
    ref = oracle.train_step(params, cfg, batch["nnet_input"], batch["sequence_length"], batch["nnet_target"], state,
                            optimizer="adam", lr=1e-2, clip_norm=5.0, l2=1e-5)
    assert abs(out["eval_loss"] - ref["eval_loss"]) / ref["eval_loss"] < 1e-4
    assert out["eval"] == ref["eval"]
    got = graph.model.ps.export_tf()
    for k in params:                                                 # exactly ONE update was applied
        assert np.isfinite(got[k]).all()
        assert np.abs(got[k] - params[k]).max() < 2e-4 * max(1.0, np.abs(params[k]).max()), k
    monkeypatch.delenv("LC_LSTM_SPIN_LIMIT")
    out2 = graph.step(batch, fetch_eval=False)                       # and the persistent schedule still works afterwards
    assert graph.persist_fallbacks == 1 and np.isfinite(out2["eval_loss"])
    assert ops.last_lstm_schedule()["kind"] == "persistent_f32" and not graph._fallback.latched
    # a cause that does not go away must not be paid for on every step: two failures IN A ROW latch the launch train
    monkeypatch.setenv("LC_LSTM_SPIN_LIMIT", "0")
    graph.step(batch, fetch_eval=False)
    assert graph.persist_fallbacks == 2 and not graph._fallback.latched
    graph.step(batch, fetch_eval=False)
    assert graph.persist_fallbacks == 3 and graph._fallback.latched
    assert "staying on it for the rest of this run" in capfd.readouterr().err
    out5 = graph.step(batch, fetch_eval=False)                       # no failed try, no second run any more
    assert graph.persist_fallbacks == 3 and np.isfinite(out5["eval_loss"])
    assert ops.last_lstm_schedule()["kind"] == "launch_train"
    assert ops.get_option("lstm_persistent") is None                 # the override is scoped to the step


@pytest.mark.parametrize("N,B,kind", [(128, 8, "persistent_x3"), (1024, 16, "persistent_x3_xcd_pair")])
def test_failed_split_operand_bptt_poisons_its_x3_shadow(monkeypatch, N, B, kind):
    """the split-operand BPTT kernels write the x3 shadow of dz themselves (schedule-word bit 18: no
    split pass follows), so a FAILED launch must poison that shadow too - the dX / dKx / dR products read it, and a caller of
    the C ABI that ignores the status word must find NaN, never the finite terms a half-finished launch left behind."""
    from lstm_ctc_amd import ops
    T = 6
    g = torch.Generator().manual_seed(N)
    rows = T * B
    seq = torch.full((B,), T, dtype=torch.int32).cuda()

    def dirs():
        gg = torch.Generator().manual_seed(N)
        return [dict(gates=torch.rand(rows, 4 * N, generator=gg).cuda(), RT=(torch.randn(4 * N, N, generator=gg) * 0.02).cuda(),
                     w_f=torch.randn(N, generator=gg).cuda() * 0.1, w_i=torch.randn(N, generator=gg).cuda() * 0.1,
                     w_o=torch.randn(N, generator=gg).cuda() * 0.1, cs=torch.randn(rows, N, generator=gg).cuda(),
                     dh=torch.randn(rows, N, generator=gg).cuda() * 0.1, dpeep=torch.zeros(3, N, device="cuda"),
                     dbias=torch.zeros(4 * N, device="cuda"), reverse=d,
                     dz_x3=torch.zeros((rows, 12 * N), dtype=torch.bfloat16, device="cuda")) for d in range(2)]

    ops.lstm_status("cuda").zero_()
    good = dirs()
    ops.lstm_bwd(good, seq, T, B, N, x3=True)
    sched = ops.last_lstm_schedule()
    assert sched["kind"] == kind and sched["dz_shadow_in_kernel"], sched
    torch.cuda.synchronize()
    assert int(ops.lstm_status("cuda").item()) == 0
    for d in good:
        assert torch.isfinite(d["gates"]).all() and torch.isfinite(d["dz_x3"].float()).all()
        assert torch.equal(d["dz_x3"], ops.split_bf16x3(d["gates"]))
    monkeypatch.setenv("LC_LSTM_SPIN_LIMIT", "0")
    bad = dirs()
    ops.lstm_bwd(bad, seq, T, B, N, x3=True)
    torch.cuda.synchronize()
    monkeypatch.delenv("LC_LSTM_SPIN_LIMIT")
    assert int(ops.lstm_status("cuda").item()) != 0
    for d in bad:
        assert torch.isnan(d["gates"]).all()
        assert torch.isnan(d["dz_x3"].float()).all()                 # every term of every element
    ops.lstm_status("cuda").zero_()


def test_all_ones_nans_pass_through_the_bf16_exchange(monkeypatch):
    """the bf16 persistent recurrences mark "piece not arrived" with the sentinel ff..ff, and a NaN whose bits
    are all ones keeps them through v_cvt_pk_bf16_f32 - two of them in one published dword used to read as stale for ever (a
    bounded spin, a failed launch, a latched fall-back) instead of arriving as the NaNs they are.  The producers now rewrite that
    one dword value: a batch poisoned with 0xffffffff NaNs must END the recurrence with NaN outputs and a CLEAN status word -
    the route to `FATAL: nan loss detected` (nnet/funcs.py:57-84), not to the launch train."""
    from lstm_ctc_amd import ops
    monkeypatch.delenv("LC_LSTM_SPIN_LIMIT", raising=False)
    T, B, N = 8, 16, 256
    g = torch.Generator().manual_seed(2)
    rows = T * B
    seq = torch.full((B,), T, dtype=torch.int32).cuda()
    ones_nan = torch.full((1,), -1, dtype=torch.int32).view(torch.float32).item()      # 0xffffffff
    zx = (torch.randn(rows, 4 * N, generator=g) * 0.3)
    zx[:B] = ones_nan                                    # every unit of every row at t = 0: all published dwords are NaN pairs
    fd = [dict(zx=zx.clone().cuda(), R=(torch.randn(N, 4 * N, generator=g) * 0.03).cuda(), w_f=None, w_i=None, w_o=None,
               cs=torch.zeros(rows, N, device="cuda"), hs=torch.zeros(rows, N, device="cuda"), reverse=0)]
    assert fd[0]["zx"].view(torch.int32)[0, 0].item() == -1
    ops.lstm_status("cuda").zero_()
    ops.lstm_fwd(fd, seq, T, B, N, 1.0, bf16=True)
    torch.cuda.synchronize()
    assert ops.last_lstm_schedule()["kind"] == "persistent_bf16"
    assert int(ops.lstm_status("cuda").item()) == 0      # no time-out: the NaNs ARRIVED
    assert torch.isnan(fd[0]["hs"]).all()                 # ... and spread through the recurrent product to every later step
    # BPTT: NaN derivatives at the first step of the backward walk
    gates = torch.rand(rows, 4 * N, generator=g)
    dh = torch.randn(rows, N, generator=g) * 0.1
    dh[(T - 1) * B:] = ones_nan
    bd = [dict(gates=gates.cuda(), RT=(torch.randn(4 * N, N, generator=g) * 0.03).cuda(), w_f=None, w_i=None, w_o=None,
               cs=torch.randn(rows, N, generator=g).cuda(), dh=dh.cuda(), dpeep=None, dbias=torch.zeros(4 * N, device="cuda"),
               reverse=0)]
    ops.lstm_bwd(bd, seq, T, B, N, bf16=True)
    torch.cuda.synchronize()
    assert ops.last_lstm_schedule()["kind"] == "persistent_bf16"
    assert int(ops.lstm_status("cuda").item()) == 0
    assert torch.isnan(bd[0]["gates"]).all()


def test_unit_gradient_reductions_are_deterministic():
    """dbias / dpeep / colsum are two-stage reductions in a fixed order: bit-identical from run to run."""
    from lstm_ctc_amd import ops
    g = torch.Generator().manual_seed(3)
    T, B, N = 300, 40, 96
    rows = T * B
    seq = torch.full((B,), T, dtype=torch.int32).cuda()
    base = dict(gates=torch.rand(rows, 4 * N, generator=g).cuda(), RT=(torch.randn(4 * N, N, generator=g) * 0.05).cuda(),
                w_f=torch.randn(N, generator=g).cuda() * 0.1, w_i=torch.randn(N, generator=g).cuda() * 0.1,
                w_o=torch.randn(N, generator=g).cuda() * 0.1, cs=torch.randn(rows, N, generator=g).cuda(),
                dh=torch.randn(rows, N, generator=g).cuda() * 0.1, reverse=0)
    outs = []
    for _ in range(3):
        d = dict(base, gates=base["gates"].clone(), dpeep=torch.zeros(3, N, device="cuda"),
                 dbias=torch.zeros(4 * N, device="cuda"))
        ops.lstm_bwd([d], seq, T, B, N)
        outs.append((d["dpeep"].cpu(), d["dbias"].cpu(), ops.colsum(d["gates"]).cpu()))
    for o in outs[1:]:
        for a, b in zip(outs[0], o):
            assert torch.equal(a, b)


# ---------------------------------------------------------------------------------------------- reference-named surface
def test_create_logits_callable(oracle):
    """nnet/graph.py:24-34,63-67: get_create_logits(type)(nnet_input, sequence_length, nnet_config) ->
    (logits [B,T,V], encoder, reg_loss list)."""
    from lstm_ctc_amd import nnet
    assert nnet.get_create_logits("gru") is None and nnet.get_create_logits(None) is None
    assert nnet.get_create_logits("cudnnlstm") is nnet.create_logits_cudnnlstm       # nnet/graph.py:24-34
    cfg = _cfg(uniform_label_sm=0.1, seed=777)
    rng = np.random.default_rng(1)
    b = _batch(rng, 4, 9, 12, 8)
    create_logits = nnet.get_create_logits("blstm")
    logits, encoder, reg_loss = create_logits(torch.from_numpy(b["nnet_input"]).cuda(),
                                              torch.from_numpy(b["sequence_length"]).cuda(), cfg)
    model = create_logits.model
    p64 = {k: v.astype(np.float64) for k, v in model.ps.export_tf().items()}
    ref, saved = oracle.forward(p64, cfg, b["nnet_input"].astype(np.float64), b["sequence_length"])
    assert tuple(logits.shape) == ref.shape
    assert np.abs(logits.cpu().numpy() - ref).max() < 1e-4 * max(1.0, np.abs(ref).max())
    np.testing.assert_allclose(encoder.cpu().numpy(), saved["encoder"], atol=1e-4)
    assert len(reg_loss) == 1 and reg_loss[0][1] == 0.1
    rl, _ = oracle.label_smoothing(ref, cfg)
    assert abs(float(reg_loss[0][0].item()) - rl) / abs(rl) < 1e-4
    lo, enc, reg = nnet.get_create_logits("lstm")(b["nnet_input"], b["sequence_length"],
                                                  _cfg(nnet_type="lstm", num_projects=12, seed=1))
    assert enc is None and reg == [] and tuple(lo.shape) == (4, 9, 8)
    ccfg = _cfg(nnet_type="cudnnlstm", num_projects=None, seed=5)
    lo, enc, reg = nnet.get_create_logits("cudnnlstm")(b["nnet_input"], b["sequence_length"], ccfg)
    names = nnet.create_logits_cudnnlstm.model.ps.names()
    assert "rnn/multi_rnn_cell/cell_1/cudnn_compatible_lstm_cell/kernel" in names and not any("diag" in n for n in names)
    p64 = {k: v.astype(np.float64) for k, v in nnet.create_logits_cudnnlstm.model.ps.export_tf().items()}
    ref, _ = oracle.forward(p64, ccfg, b["nnet_input"].astype(np.float64), b["sequence_length"])
    assert enc is None and reg == [] and np.abs(lo.cpu().numpy() - ref).max() < 1e-4 * max(1.0, np.abs(ref).max())
    with pytest.raises(ValueError):                       # lstm.py:99 would reshape [.., N] into [-1, P]
        nnet.get_create_logits("cudnnlstm")(b["nnet_input"], b["sequence_length"], _cfg(nnet_type="cudnnlstm"))


# ---------------------------------------------------------------------------------------------- multi-process paths
def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_bench_under_torchrun_on_rccl():
    """`python -m torch.distributed.run --nproc-per-node 1 bench.py ...` in a FRESH child process: init_process_group
    ("nccl" = RCCL), the all-reduce of the flat gradient, the MAX-reduce of the timing and the barrier all execute on
    real RCCL (world size 1 on this box), and the JSON line says how many ranks RCCL saw."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1",
           "--workload", "c2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-secondary",
           "--no-cli-corpus"]
    r = subprocess.run(cmd, capture_output=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 1 and line["config"]["parallelism"] == "dp1"
    assert line["config"]["rccl_ranks"] == 1 and line["config"]["collective_backend"] == "nccl"
    assert line["value"] > 0 and np.isfinite(line["config"]["last_loss_per_label"])
    assert line["config"]["persist_fallbacks"] == 0
    # the per-layer gradient buckets on RCCL (async all-reduces issued during the backward, waited for in front of every
    # recurrence): forced on with one rank, on the headline workload; the losses must be those of the unbucketed run
    losses = []
    for buckets in ("2", "0"):
        cmd4 = cmd[:cmd.index("--workload")] + ["--workload", "c4", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                                                 "--no-secondary", "--no-cli-corpus"]
        cmd4[cmd4.index("--master-port") + 1] = str(_free_port())
        r = subprocess.run(cmd4, capture_output=True, timeout=900, env=dict(env, LC_DP_BUCKETS=buckets), cwd=ROOT)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        line = json.loads([l for l in r.stdout.decode().splitlines() if l.startswith("{")][0])
        assert line["config"]["rccl_ranks"] == 1 and line["config"]["persist_fallbacks"] == 0
        losses.append(line["config"]["last_loss_per_label"])
    assert np.isfinite(losses[0]) and losses[0] == losses[1], losses


def test_bench_self_launch_runs_rccl_ranks():
    """`python bench.py --gpus N` WITHOUT torchrun starts its N ranks itself (a child torch.distributed.run; the parent
    never touches the GPU).  One GPU here, so the path is driven with --launch torchrun at N = 1: the line must come from
    an RCCL process group of exactly the asked size.  And a rank count that differs from --gpus is an error, not a
    smaller job."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "TORCHELASTIC_RUN_ID"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--launch", "torchrun", "--workload", "c2",
           "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 1 and line["config"]["rccl_ranks"] == 1 and line["config"]["collective_backend"] == "nccl"
    assert "self-launch" in line["config"]["launched_by"] and line["config"]["lc_overrides"] == {}
    assert line["config"]["per_rank_ms_per_step"]["max"] == line["ms_per_step"]
    # two GPUs asked for on a one-GPU box: refused before anything runs, no JSON line
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode != 0 and b"refusing to run" in r.stderr and b"{" not in r.stdout
    # torchrun with ONE rank but --gpus 2: the rank refuses
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2",
           "--steps", "1", "--warmup", "0", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode != 0 and b"rank(s) were launched" in r.stderr and b"{" not in r.stdout


_DP_WORKER = r"""
import os, sys, json
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, {root!r})
rank, world = int(sys.argv[1]), int(sys.argv[2])
os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = sys.argv[3]
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=world)       # two ranks share the one GPU of this box: gloo
from lstm_ctc_amd.nnet.graph import create_graph_for_training_ctc
cfg = json.loads(sys.argv[4])
data = np.load(sys.argv[5])
graph = create_graph_for_training_ctc(None, cfg, learn_rate=1e-2, clip_norm=5.0, optimizer="adam", seed=21,
                                      process_group=dist.group.WORLD if world > 1 else None)
sl = slice(rank, None, world)
losses = []
for step in range(3):
    batch = dict(nnet_input=data["x%d" % step][sl], sequence_length=data["seq%d" % step][sl],
                 nnet_target=data["lab%d" % step][sl])
    out = graph.step(batch, fetch_eval=True)
    t = torch.tensor([out["eval_loss"], out["eval"], out["size"]], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t)
    losses.append(t.numpy().copy())
if rank == 0:
    np.savez(sys.argv[6], losses=np.stack(losses), norm=out["grad_norm"],
             **{{"p_" + k.replace("/", "__"): v for k, v in graph.model.ps.export_tf().items()}})
if world > 1:
    dist.barrier()
    dist.destroy_process_group()
"""


def test_ctcgraph_two_ranks_equal_one_full_batch(tmp_path):
    """CTCGraph._apply_gradients under a process group, with the HIP kernels (not the oracle): two processes, each
    with every other utterance, all-reduce the flat gradient before L2 / clip / Adam - three steps must give the
    parameters of ONE process on the full batch (dropout off; clip acts on the TOTAL gradient of a SUM loss)."""
    cfg = _cfg(num_layers=3)
    rng = np.random.default_rng(31)
    arrays = {}
    for step in range(3):
        b = _batch(rng, 6, 12, 12, 8)
        arrays.update({"x%d" % step: b["nnet_input"], "seq%d" % step: b["sequence_length"],
                       "lab%d" % step: b["nnet_target"]})
    data = str(tmp_path / "data.npz")
    np.savez(data, **arrays)
    script = tmp_path / "worker.py"
    script.write_text(_DP_WORKER.format(root=ROOT))

    def run(world, tag, **env):
        port = str(_free_port())
        out = str(tmp_path / ("out_%s.npz" % tag))
        procs = [subprocess.Popen([sys.executable, str(script), str(r), str(world), port, json.dumps(cfg), data, out],
                                  stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=ROOT, env=dict(os.environ, **env))
                 for r in range(world)]
        for p in procs:
            so, se = p.communicate(timeout=600)
            assert p.returncode == 0, se.decode()[-2000:]
        return np.load(out)

    # LC_OVERLAP_WGRAD=0: the weight gradients stay on the main stream, so the per-layer gradient buckets
    # (dp.GradientBuckets: layer i + 1's range goes out behind the BPTT of layer i) are active at this small width too;
    # LC_DP_BUCKETS=0: one all-reduce of the whole flat gradient after the backward.  Both must equal one process.
    one = run(1, "one", LC_OVERLAP_WGRAD="0")
    for tag, env in (("buckets", dict(LC_OVERLAP_WGRAD="0")), ("whole", dict(LC_OVERLAP_WGRAD="0", LC_DP_BUCKETS="0")),
                     ("overlapped_wgrad", {})):
        two = run(2, tag, **env)
        np.testing.assert_allclose(two["losses"], one["losses"], rtol=1e-5)
        assert abs(float(two["norm"]) - float(one["norm"])) / float(one["norm"]) < 1e-4
        for k in one.files:
            if k.startswith("p_"):
                # (Adam divides by sqrt(v) + 1e-8: entries with gradients near 1e-8 turn the 1e-7 summation-order difference
                # between two reduced half batches and one full batch into a few 1e-5 after three steps at lr = 1e-2; a
                # range that missed its all-reduce is off by ~1e-2)
                assert np.abs(two[k] - one[k]).max() < 5e-5 * max(1.0, np.abs(one[k]).max()), (tag, k)


def test_dropout_masks_differ_across_ranks():
    """Rank r's utterance b must not reuse rank 0's dropout mask: the per-rank dropout stream is seeded with the rank,
    the INIT seed stays common."""
    from lstm_ctc_amd.nnet.graph import CTCGraph

    class FakePG:
        pass

    import lstm_ctc_amd.nnet.dp as dp
    cfg = _cfg(dropout_rate=0.8)
    seeds = []
    flats = []
    orig_w, orig_r = dp.world_size, dp.rank
    try:
        for r in range(2):
            dp.world_size = lambda pg: 2
            dp.rank = lambda pg, r=r: r
            g = CTCGraph(None, cfg, learn_rate=1e-3, optimizer="sgd", seed=5, process_group=FakePG())
            seeds.append(g.drop_seed)
            flats.append(g.model.ps.flat.clone())
    finally:
        dp.world_size, dp.rank = orig_w, orig_r
    assert seeds[0] != seeds[1]
    assert torch.equal(flats[0], flats[1])
