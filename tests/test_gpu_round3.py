"""GPU tests added in round 3: the batching pipeline's time-major staging buffer through CTCGraph (one DMA instead of a
copy + transposing kernel), the one-batch-ahead upload of Session.run, and the per-thread library switches."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pytestmark = pytest.mark.gpu


def _corpus(tmp_path, n=10, D=6, V=9):
    from lstm_ctc_amd.nnet import tfrecord as tr
    rng = np.random.default_rng(4)
    lines = []
    for i in range(n):
        T, L = int(rng.integers(20, 60)), int(rng.integers(1, 6))
        p = str(tmp_path / ("u%d.tfrecords" % i))
        tr.write_tfrecord(p, rng.normal(size=(T, D)).astype(np.float32), rng.integers(0, V - 1, size=L))
        lines.append("u%d %d %d 1 %s" % (i, T, D, p))
    scp = tmp_path / "tfrecords.scp"
    scp.write_text("\n".join(lines) + "\n")
    return str(scp)


CFG = dict(nnet_type="blstm", input_dim=6, left_context=1, right_context=1, subsample=2, num_layers=2, num_neurons=32,
           num_projects=16, num_targets=9, use_peepholes=True, dropout_rate=1.0)


def test_pipeline_view_uploads_like_a_contiguous_batch(tmp_path):
    """The loader hands out a [B,T,D] VIEW of page-locked time-major memory; CTCGraph uploads the buffer underneath.
    Same losses, bit for bit, as the same batch as a contiguous [B,T,D] array (the reference's contract)."""
    from lstm_ctc_amd import nnet
    scp = _corpus(tmp_path)
    _, ds, dim = nnet.dataset_from_tfrecords(scp, 1, 1, 2, num_parallel_calls=4)
    _, pipe = nnet.create_pipeline_sequence_batch(ds, dim, batch_size=4)
    g1 = nnet.create_graph_for_training_ctc(pipe, CFG, learn_rate=1e-2, optimizer="adam", seed=3)
    g2 = nnet.create_graph_for_training_ctc(None, CFG, learn_rate=1e-2, optimizer="adam", seed=3)
    n = 0
    for batch in pipe:
        assert not batch["nnet_input"].flags.c_contiguous and batch["nnet_input"].transpose(1, 0, 2).flags.c_contiguous
        a = g1.step(batch, fetch_eval=True)
        b = g2.step(dict(batch, nnet_input=np.ascontiguousarray(batch["nnet_input"])), fetch_eval=True)
        assert a["eval_loss"] == b["eval_loss"] and a["eval"] == b["eval"] and a["size"] == b["size"]
        n += 1
    assert n == 3
    assert torch.equal(g1.model.ps.flat, g2.model.ps.flat)


def test_session_uploads_one_batch_ahead(tmp_path, capfd):
    """Session.run stages batch k + 1 on a copy stream before step k is enqueued; the run loop's results are those of
    plain step() calls over the same batches, and it logs the throughput line in front of tr_loss."""
    from lstm_ctc_amd import nnet
    scp = _corpus(tmp_path, n=22)
    _, ds, dim = nnet.dataset_from_tfrecords(scp, 1, 1, 2)
    _, pipe = nnet.create_pipeline_sequence_batch(ds, dim, batch_size=4)
    g1 = nnet.create_graph_for_training_ctc(pipe, CFG, learn_rate=1e-2, optimizer="adam", seed=5)
    nnet.train(nnet.Session(g1), g1, evaluate=True, report_interval=2)
    err = capfd.readouterr().err.splitlines()
    assert err[-1].startswith("INFO:tensorflow:tr_loss = ") and err[-2].startswith("INFO:tensorflow:throughput: steps = 3,")
    g2 = nnet.create_graph_for_training_ctc(None, CFG, learn_rate=1e-2, optimizer="adam", seed=5)
    for batch in pipe:
        g2.step(batch, fetch_eval=True)
    assert g1.global_step == g2.global_step == 6
    assert torch.equal(g1.model.ps.flat, g2.model.ps.flat)


def test_thread_local_library_switches():
    """lc_set_option overrides are per thread and win over the environment; force_launch_train is built on them."""
    import threading
    from lstm_ctc_amd import ops
    T, B, N = 6, 4, 64
    seq = torch.full((B,), T, dtype=torch.int32, device="cuda")
    mk = lambda *s: (torch.randn(*s, device="cuda") * 0.3)

    def run():
        d = dict(zx=mk(T * B, 4 * N), R=mk(N, 4 * N) * 0.2, w_f=mk(N), w_i=mk(N), w_o=mk(N),
                 cs=torch.zeros(T * B, N, device="cuda"), hs=torch.zeros(T * B, N, device="cuda"), reverse=0)
        ops.lstm_fwd([d], seq, T, B, N, 1.0)
        return ops.last_lstm_schedule()["kind"]

    assert run() == "persistent_f32"
    with ops.force_launch_train():
        assert run() == "launch_train"
        seen = []
        t = threading.Thread(target=lambda: seen.append(ops.get_option("lstm_persistent")))
        t.start()
        t.join()
        assert seen == [None]                     # another thread does not see this thread's override
    assert run() == "persistent_f32" and ops.get_option("lstm_persistent") is None


def test_bench_two_ranks_share_one_gpu_over_gloo():
    """The N > 1 code of bench.py - per-rank seeds and batches, gradient buckets / whole-buffer all-reduce, the all-gather of
    the ranks' times and the MAX, `allreduce` breakdown, rank 0 printing alone, the group torn down together - executed
    with TWO ranks on this box's one GPU (gloo; RCCL refuses two ranks on one device).  Not a performance figure: it
    protects the driver's 1 / 2 / 4 / 8 run from a launcher bug.  The persistent recurrences need the whole GPU, so the
    two co-tenants run the launch train."""
    import json
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, LC_BENCH_SHARED_GPU="1", LC_LSTM_PERSISTENT="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "TORCHELASTIC_RUN_ID"):
        env.pop(k, None)
    for buckets in ("1", "0"):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
               "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo",
               "--workload", "c4" if buckets == "1" else "c2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
        r = subprocess.run(cmd, capture_output=True, timeout=1200, env=dict(env, LC_DP_BUCKETS=buckets), cwd=ROOT)
        assert r.returncode == 0, r.stderr.decode()[-3000:]
        lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
        assert len(lines) == 1                                   # rank 0 alone prints
        line = json.loads(lines[0])
        cfg = line["config"]
        assert line["n_gpus"] == 2 and cfg["parallelism"] == "dp2" and cfg["rccl_ranks"] == 2
        assert cfg["collective_backend"] == "gloo" and cfg["global_batch"] == (128 if buckets == "1" else 64)
        assert cfg["per_rank_ms_per_step"]["max"] == line["ms_per_step"] >= cfg["per_rank_ms_per_step"]["min"] > 0
        assert line["value"] > 0 and np.isfinite(cfg["last_loss_per_label"])
        assert line["allreduce"]["whole_gradient_alone"]["bytes"] > 0
        assert "secondary" not in line and "cli_corpus" not in line and "cpu_baseline" not in line
        assert cfg["lc_overrides"] == {"LC_DP_BUCKETS": buckets, "LC_LSTM_PERSISTENT": "0"}
        rk = cfg["ranks"]                                        # all-gathered health of BOTH ranks
        assert rk["n"] == 2 and rk["persist_fallbacks"] == {"min": 0, "max": 0} and rk["lstm_schedule"] == ["launch_train"]
        assert rk["last_loss_per_label"]["min"] <= cfg["last_loss_per_label"] <= rk["last_loss_per_label"]["max"]


def test_bench_refuses_a_run_with_one_rank_on_the_launch_train():
    """Two gloo ranks on the box's one GPU; rank 1's persistent recurrences give up at once (LC_LSTM_SPIN_LIMIT = 0 on that
    rank only), so the step is re-run on the launch train - by every rank, the status word travels with the gradient's collective - and
    the ranks latch there.  `value` is all ranks' frames over the MAX of
    their times: such a run must end non-zero WITHOUT a JSON line, and say which rank.  (The healthy control is
    test_bench_two_ranks_share_one_gpu_over_gloo, whose line now carries the all-gathered `ranks` object.)"""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, LC_BENCH_SHARED_GPU="1", LC_LSTM_PERSISTENT="0", HSA_ENABLE_IPC_MODE_LEGACY="0",
               LC_BENCH_FAIL_RANK="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "TORCHELASTIC_RUN_ID"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo",
           "--workload", "c2", "--steps", "2", "--warmup", "2", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, timeout=1200, env=env, cwd=ROOT)
    err = r.stderr.decode()
    assert r.returncode != 0, err[-2000:]
    assert not [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    # the update guard is summed over the ranks, so BOTH ranks re-run the failed steps together and both latch
    assert "refusing to print a value" in err and "rank(s) [0, 1]" in err and "latched" in err


def test_bad_label_in_the_next_batch_does_not_pre_empt_this_step():
    """Session.run uploads batch k + 1 before step k runs; a label outside the alphabet in batch k + 1 must surface when
    THAT batch is consumed - step k trains and reports first (the check used to run inside stage())."""
    from lstm_ctc_amd import nnet
    rng = np.random.default_rng(0)

    def batch(bad):
        x = rng.normal(size=(3, 12, 6)).astype(np.float32)
        y = np.full((3, 4), -1, np.int64)
        y[:, :2] = rng.integers(0, 8, size=(3, 2))
        if bad:
            y[1, 0] = 8                                     # the blank (V - 1) is not a label
        return {"nnet_input": x, "sequence_length": np.full(3, 12, np.int32), "nnet_target": y}

    cfg = dict(CFG, left_context=0, right_context=0, subsample=0)
    graph = nnet.create_graph_for_training_ctc([batch(False), batch(True), batch(False)], cfg, learn_rate=1e-2,
                                               optimizer="sgd", seed=1)
    sess = nnet.Session(graph)
    nodes = {"size": "size", "loss": "loss", "train": "train", "eval_loss": "eval_loss"}
    out = sess.run(nodes)                                   # step 1 - batch 2 (bad) is already staged
    assert out["size"] == 6 and np.isfinite(out["eval_loss"]) and graph.global_step == 1
    with pytest.raises(ValueError, match="label outside"):
        sess.run(nodes)
    assert graph.global_step == 1
