"""The N > 1 hazard that one GPU can rehearse (`nnet/dp.py:41-77`).

A persistent recurrence needs one workgroup on EVERY CU of the XCDs it uses (the whole register file of each), so a foreign
kernel that is resident when it starts - under data parallelism: a collective's kernel waiting for a slower peer - takes CUs it
cannot have.  What must happen then: the launch completes once the foreign kernel has drained (its late workgroups claim the
missing slices; every wait is bounded), or the bounded waits run out LOUDLY (NaN outputs, sticky status word) and the step is
re-run on the launch train with correct results - never a hang, never stale data.  And with `dp.GradientBuckets.wait()` in front
- what `Model.backward` does before every recurrence - the persistent launch does not meet the foreign kernel at all.
The foreign kernel is `lc_debug_spin` (idle resident workgroups on a side stream)."""
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

N, B, T = 1024, 64, 24          # the XCD-pair schedule: 32 workgroups on each of the 8 XCDs = every CU of the chip


def _dirs(seed=5):
    g = torch.Generator().manual_seed(seed)
    mk = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).cuda()
    return [dict(zx=mk(T * B, 4 * N, sc=0.5), R=mk(N, 4 * N, sc=0.5 / N ** 0.5), w_f=mk(N, sc=0.2), w_i=mk(N, sc=0.2),
                 w_o=mk(N, sc=0.2), cs=torch.zeros(T * B, N, device="cuda"), hs=torch.zeros(T * B, N, device="cuda"),
                 reverse=d) for d in range(2)]


def _run(ops, seq, x3, d=None):
    d = _dirs() if d is None else d
    ops.lstm_fwd(d, seq, T, B, N, 1.0, x3=x3)
    return d


def _timed(ops, seq, x3):
    d = _dirs()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    _run(ops, seq, x3, d)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


@pytest.mark.parametrize("x3", [False, True], ids=["fp32_pair", "x3_pair"])
def test_persistent_launch_beside_a_foreign_resident_kernel_completes(monkeypatch, x3):
    """48 foreign workgroups (6 per XCD) resident for 80 ms, then the XCD-pair recurrence on the main stream: the 26 workgroups
    per XCD that fit start and wait (bounded) for the slices nobody holds yet; when the foreign kernel leaves, the remaining
    candidates become resident, claim those slices and the recurrence runs - results bit-identical to an undisturbed launch,
    status word clean, and the call cannot have finished before the foreign kernel did."""
    from lstm_ctc_amd import ops
    monkeypatch.delenv("LC_LSTM_SPIN_LIMIT", raising=False)
    seq = torch.full((B,), T, dtype=torch.int32).cuda()
    ops.lstm_status("cuda").zero_()
    ref = _run(ops, seq, x3)
    kind = ops.last_lstm_schedule()["kind"]
    assert kind == ("persistent_x3_xcd_pair" if x3 else "persistent_f32_xcd_pair")
    torch.cuda.synchronize()
    assert int(ops.lstm_status("cuda").item()) == 0
    side = torch.cuda.Stream(priority=-1)
    foreign_ms = 80
    undisturbed_ms = _timed(ops, seq, x3)
    got = _dirs()                                      # (host-side generation + upload: before the foreign kernel starts)
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        ops.debug_spin(48, foreign_ms * 1000)
    time.sleep(0.01)                                   # resident before the persistent launch is enqueued
    t0 = time.perf_counter()
    _run(ops, seq, x3, got)
    torch.cuda.current_stream().synchronize()
    dt_ms = (time.perf_counter() - t0) * 1e3
    torch.cuda.synchronize()
    assert ops.last_lstm_schedule()["kind"] == kind
    status = int(ops.lstm_status("cuda").item())
    if status == 0:
        for a, b in zip(ref, got):
            assert torch.equal(a["hs"], b["hs"]) and torch.equal(a["cs"], b["cs"])
        assert undisturbed_ms < 20 and dt_ms > 0.6 * foreign_ms, \
            "the recurrence cannot have had every CU while the foreign kernel was resident: %.1f ms (alone: %.1f)" % (dt_ms, undisturbed_ms)
    else:                                              # the bounded waits ran out first: loud, whole outputs
        for b in got:
            assert torch.isnan(b["hs"]).all()
        ops.lstm_status("cuda").zero_()
    assert dt_ms < 30000


def test_spin_limit_beside_a_foreign_kernel_is_loud_and_wait_in_front_avoids_it(monkeypatch):
    """The same meeting with short bounded waits (LC_LSTM_SPIN_LIMIT = 20000 polls, a few milliseconds) and a foreign kernel
    that stays 300 ms: the launch must give up loudly (every output NaN, status word set) - what `CTCGraph` turns into a re-run
    on the launch train - unless the runtime serialised the two streams on one hardware queue (seen on one box of the pool):
    then it ran behind the foreign kernel and must simply be right.  With `dp.GradientBuckets.wait()` in front (the foreign kernel as an outstanding bucket: the compute
    stream waits for it) the SAME limits are never reached: clean status, results of the undisturbed launch."""
    from lstm_ctc_amd import ops
    from lstm_ctc_amd.nnet import dp
    seq = torch.full((B,), T, dtype=torch.int32).cuda()
    monkeypatch.delenv("LC_LSTM_SPIN_LIMIT", raising=False)
    ops.lstm_status("cuda").zero_()
    ref = _run(ops, seq, False)
    torch.cuda.synchronize()
    monkeypatch.setenv("LC_LSTM_SPIN_LIMIT", "20000")
    undisturbed = _run(ops, seq, False)                # the short limit alone does not trip an undisturbed launch
    torch.cuda.synchronize()
    assert int(ops.lstm_status("cuda").item()) == 0 and torch.equal(undisturbed[0]["hs"], ref[0]["hs"])
    side = torch.cuda.Stream(priority=-1)              # (a priority stream does not share a hardware queue with the caller's)
    bad = _dirs()
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        ops.debug_spin(48, 300000)
    time.sleep(0.01)
    t0 = time.perf_counter()
    _run(ops, seq, False, bad)
    torch.cuda.synchronize()
    dt_ms = (time.perf_counter() - t0) * 1e3
    if int(ops.lstm_status("cuda").item()) != 0:       # they met and the short waits ran out: loud, whole outputs
        for b in bad:
            assert torch.isnan(b["hs"]).all()
        ops.lstm_status("cuda").zero_()
    else:                                              # the runtime put both streams on one hardware queue: the launch never met
        assert dt_ms > 200, dt_ms                      # the foreign kernel - it ran behind it, and then it must be right
        for a, b in zip(ref, bad):
            assert torch.equal(a["hs"], b["hs"])

    class _Outstanding:                                # what dist.all_reduce(..., async_op=True) hands back: wait() = stream wait
        def __init__(self, ev):
            self.ev = ev

        def wait(self):
            torch.cuda.current_stream().wait_event(self.ev)

    good = _dirs()
    buckets = dp.GradientBuckets(torch.zeros(16, device="cuda"), None)
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        ops.debug_spin(48, 150000)
        ev = torch.cuda.Event()
        ev.record()
    time.sleep(0.01)
    buckets.pending.append(_Outstanding(ev))
    t0 = time.perf_counter()
    buckets.wait()                                     # Model.backward: in front of every persistent recurrence
    _run(ops, seq, False, good)
    torch.cuda.synchronize()
    dt_ms = (time.perf_counter() - t0) * 1e3
    assert int(ops.lstm_status("cuda").item()) == 0
    assert ops.last_lstm_schedule()["kind"] == "persistent_f32_xcd_pair"
    for a, b in zip(ref, good):
        assert torch.equal(a["hs"], b["hs"])
    assert dt_ms > 100                                 # it did wait for the foreign kernel - outside the recurrence


def test_train_step_beside_a_foreign_kernel_recovers_on_the_launch_train(monkeypatch, capfd):
    """Graph level: a c4-width train step that meets a resident foreign kernel with short bounded waits fails its persistent
    launch, re-runs on the launch train IN PROCESS and ends with the loss and parameters of an undisturbed step."""
    from lstm_ctc_amd import ops
    from lstm_ctc_amd.nnet.graph import create_graph_for_training_ctc
    cfg = dict(nnet_type="blstm", input_dim=40, left_context=0, right_context=0, num_layers=1, num_neurons=1024,
               num_projects=1024, num_targets=44, use_peepholes=True, dropout_rate=1.0)
    rng = np.random.default_rng(4)
    Bs, Ts, L = 32, 20, 5
    batch = {"nnet_input": rng.normal(size=(Bs, Ts, 40)).astype(np.float32),
             "sequence_length": np.full(Bs, Ts, np.int32),
             "nnet_target": rng.integers(0, 43, size=(Bs, L)).astype(np.int64)}
    monkeypatch.delenv("LC_LSTM_SPIN_LIMIT", raising=False)
    losses, params = {}, {}
    for name in ("undisturbed", "beside_foreign"):
        graph = create_graph_for_training_ctc(None, cfg, learn_rate=1e-3, clip_norm=5.0, optimizer="adam", seed=9)
        if name == "beside_foreign":
            monkeypatch.setenv("LC_LSTM_SPIN_LIMIT", "20000")
            side = torch.cuda.Stream(priority=-1)
            with torch.cuda.stream(side):
                ops.debug_spin(48, 400000)
            time.sleep(0.01)
        t0 = time.perf_counter()
        out = graph.step(batch, fetch_eval=True)
        torch.cuda.synchronize()
        dt_ms = (time.perf_counter() - t0) * 1e3
        losses[name], params[name] = out["eval_loss"], graph.model.ps.export_tf()
        if name == "beside_foreign":
            if graph.persist_fallbacks >= 1:           # met the foreign kernel, gave up loudly, recovered in process
                assert "re-running the step with the per-step launch train" in capfd.readouterr().err
            else:                                      # (one hardware queue for both streams: the step ran behind it)
                assert dt_ms > 300, dt_ms
        else:
            assert graph.persist_fallbacks == 0
        del graph
    assert np.isfinite(losses["beside_foreign"])
    assert abs(losses["beside_foreign"] - losses["undisturbed"]) <= 1e-4 * abs(losses["undisturbed"])
    for k, v in params["undisturbed"].items():
        assert np.isfinite(params["beside_foreign"][k]).all()
        assert np.abs(params["beside_foreign"][k] - v).max() <= 2e-4 * max(1.0, np.abs(v).max()), k
