"""CPU tests of the host-side data formats and run loops (no GPU, no kernels)."""
import io
import os
import struct
import subprocess
import sys
import time

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_tfrecord_roundtrip_and_framing(tmp_path):
    from lstm_ctc_amd.nnet import tfrecord as tr
    rng = np.random.default_rng(0)
    x = rng.normal(size=(7, 5)).astype(np.float32)
    y = np.array([3, 0, 71, 2], np.int64)
    p = str(tmp_path / "a.tfrecords")
    tr.write_tfrecord(p, x, y)
    raw = open(p, "rb").read()
    (ln,) = struct.unpack("<Q", raw[:8])
    assert len(raw) == 8 + 4 + ln + 4
    assert struct.unpack("<I", raw[8:12])[0] == tr.masked_crc(raw[:8])
    assert struct.unpack("<I", raw[-4:])[0] == tr.masked_crc(raw[12:12 + ln])
    assert tr.crc32c(b"123456789") == 0xE3069283                       # CRC-32C check value
    ex = tr.parse_sequence_example(tr.read_tfrecord(p)[0])
    np.testing.assert_array_equal(np.stack(ex["nnet_input"]), x)
    assert [int(v[0]) for v in ex["nnet_target"]] == list(y)


def test_splice_and_subsample_semantics():
    """tfrecord.py:28-51: edge frames replicated; subsample keeps 0,f,2f.. (floor(T/f) frames)."""
    from lstm_ctc_amd.nnet.tfrecord import splice, subsample
    x = np.arange(5, dtype=np.float32)[:, None] * np.ones((1, 2), np.float32)
    s = splice(x, 1, 2)
    assert s.shape == (5, 8)
    assert list(s[0, ::2]) == [0, 0, 1, 2] and list(s[4, ::2]) == [3, 4, 4, 4] and list(s[2, ::2]) == [1, 2, 3, 4]
    assert list(subsample(np.arange(10)[:, None], 3)[:, 0]) == [0, 3, 6]
    assert list(subsample(np.arange(9)[:, None], 3)[:, 0]) == [0, 3, 6]


def test_scp_consistency_checks(tmp_path):
    from lstm_ctc_amd.nnet import dataset_from_tfrecords
    scp = tmp_path / "bad.scp"
    scp.write_text("a 3 40 1 a.tfrecords\nb 3 41 1 b.tfrecords\n")
    with pytest.raises(SystemExit):
        dataset_from_tfrecords(str(scp))


def test_kaldi_ark_bytes(tmp_path):
    from lstm_ctc_amd.kaldi_io import BaseFloatMatrixWriter, read_float_matrix_ark
    m = np.arange(6, dtype=np.float32).reshape(2, 3)
    ark, scp = str(tmp_path / "o.ark"), str(tmp_path / "o.scp")
    w = BaseFloatMatrixWriter("ark,scp:%s,%s" % (ark, scp))
    w.Write("utt1", m)
    w.Write("utt2", m * 2)
    w.Close()
    raw = open(ark, "rb").read()
    assert raw.startswith(b"utt1 \0BFM \x04" + struct.pack("<i", 2) + b"\x04" + struct.pack("<i", 3) + m.tobytes())
    got = read_float_matrix_ark(ark)
    np.testing.assert_array_equal(got["utt2"], m * 2)
    lines = open(scp).read().split("\n")
    assert lines[0] == "utt1 %s:5" % ark                                # offset points at the \0B marker


def test_flatten_labels_matches_oracle(oracle):
    from lstm_ctc_amd.nnet.graph import flatten_labels
    dense = np.array([[3, 1, -1, -1], [-1, -1, -1, -1], [0, 0, 5, 2]], np.int64)
    flat, offs, mx = flatten_labels(dense)
    of, oo = oracle.flatten_labels(dense)
    assert list(flat) == list(of) == [3, 1, 0, 0, 5, 2] and list(offs) == list(oo) == [0, 2, 2, 6] and mx == 4


class _FakeGraph:
    keys = ["size", "train", "summary", "loss", "eval_loss", "sequence_length", "eval"]
    pg, world = None, 1

    class model:
        device = "cpu"

    def __getitem__(self, k):
        return k


class _FakeSession:
    def __init__(self, triples):
        self.it = iter(triples)

    def run(self, nodes):
        from lstm_ctc_amd.nnet.graph import OutOfRangeError
        try:
            size, loss, ev = next(self.it)
        except StopIteration:
            raise OutOfRangeError()
        return {"size": size, "eval_loss": loss, "loss": loss, "eval": ev, "train": None, "summary": None,
                "sequence_length": None}


def test_run_loops_log_contract(capfd, oracle):
    """nnet.train / nnet.validate on a synthetic sequence of (size, eval_loss, eval) triples: same running
    means as the oracle restatement of funcs.py:48-54 and the exact machine-parsed log lines (App. D)."""
    from lstm_ctc_amd.nnet import funcs
    triples = [(10, 25.0, 4.0), (0, 0.0, 0.0), (30, 45.0, 3.0), (7, 3.5, 7.0)]
    rs = oracle.RunningStats()
    for s, l, e in triples:
        rs.update(s, l, e)
    funcs.train(_FakeSession(triples), _FakeGraph(), evaluate=True, report_interval=2)
    err = capfd.readouterr().err.strip().split("\n")
    assert err[0].startswith("INFO:tensorflow:step = 2, batch_size = 0, loss = ")
    assert err[-2] == "INFO:tensorflow:done"
    assert err[-1] == "INFO:tensorflow:tr_loss = %f" % rs.loss
    funcs.validate(_FakeSession(triples), _FakeGraph(), evaluate=True, report_interval=None)
    err = capfd.readouterr().err.strip().split("\n")
    assert err == ["INFO:tensorflow:done", "INFO:tensorflow:cv_loss = %f" % rs.loss, "INFO:tensorflow:cv_eval = %f" % rs.acc]
    # the bash side takes the last whitespace field (scripts/train.sh:145)
    assert float(err[1].split()[-1]) == pytest.approx(rs.loss, abs=1e-6)


def test_nan_loss_exits_1(capfd):
    from lstm_ctc_amd.nnet import funcs
    with pytest.raises(SystemExit) as e:
        funcs.train(_FakeSession([(5, 1.0, 0.0), (5, float("nan"), 0.0)]), _FakeGraph(), evaluate=False)
    assert e.value.code == 1
    err = capfd.readouterr().err.strip().split("\n")
    assert err[-2] == "INFO:tensorflow:tr_loss = nan" and err[-1] == "FATAL:tensorflow:nan loss detected"


def test_stalled_step_ends_the_process_with_status_1(tmp_path):
    """A step that never completes (a wedged runtime call, a collective without its peer) must not hang scripts/train*.sh:
    the run loop's watchdog logs FATAL:tensorflow: and ends the process with status 1 - the exit contract of
    nnet/funcs.py:64-81 (NaN => exit 1) - after LC_STEP_TIMEOUT seconds without a completed sess.run."""
    prog = tmp_path / "stall.py"
    prog.write_text(
        "import sys, time\n"
        "sys.path.insert(0, %r)\n"
        "from lstm_ctc_amd.nnet import funcs\n"
        "class G:\n"
        "    pg = None\n"
        "    class model: device = 'cpu'\n"
        "    def __getitem__(self, k): return k\n"
        "class S:\n"
        "    n = 0\n"
        "    def run(self, nodes):\n"
        "        S.n += 1\n"
        "        if S.n > 2:\n"
        "            time.sleep(3600)          # the hung call: no exception can leave it\n"
        "        return {'size': 5, 'eval_loss': 1.0, 'loss': 1.0, 'eval': 0.0, 'sequence_length': None}\n"
        "funcs.train(S(), G(), evaluate=False, report_interval=1)\n" % ROOT)
    t0 = time.time()
    r = subprocess.run([sys.executable, str(prog)], capture_output=True, text=True, timeout=120,
                       env=dict(os.environ, LC_STEP_TIMEOUT="1.5"))
    assert r.returncode == 1, (r.returncode, r.stderr[-500:])
    lines = r.stderr.strip().split("\n")
    assert lines[0].startswith("INFO:tensorflow:step = 1,") and lines[1].startswith("INFO:tensorflow:step = 2,")
    assert lines[-1].startswith("FATAL:tensorflow:no training step completed for ") and "after 2 completed step(s)" in lines[-1]
    assert not any(l.startswith("INFO:tensorflow:tr_loss") for l in lines)     # train.sh must not read a loss from this run
    assert time.time() - t0 < 60


def test_watchdog_is_quiet_on_a_healthy_loop():
    from lstm_ctc_amd.nnet.funcs import StepWatchdog
    fired = []
    with StepWatchdog(timeout=0.4, _exit=fired.append) as dog:
        for _ in range(8):
            time.sleep(0.1)
            dog.kick()
    assert fired == []
    with StepWatchdog(timeout=0.2, _exit=fired.append):
        time.sleep(0.8)
    assert fired == [1]
    assert StepWatchdog(timeout=0).start()._thread is None           # LC_STEP_TIMEOUT=0: off
    # nnet-forward pauses it while it writes the archive (a slow reader on `ark:-` is back-pressure, not a hang - ADVICE
    # round 4) and resumes it for the next batch's device work, which is timed from the resume
    fired = []
    with StepWatchdog(timeout=0.2, _exit=fired.append) as dog:
        dog.kick()
        dog.pause()
        time.sleep(0.7)                                               # "writing"
        assert fired == []
        dog.resume()
        time.sleep(0.1)
        assert fired == []
        time.sleep(0.6)                                               # a device call that does not return
    assert fired == [1]


def test_cli_flag_surface():
    """The four command lines accept the reference's flags (bin/nnet-train.py:113-151 etc.)."""
    for cli, must in (("nnet-train.py", ["--objective", "--optimizer", "--evaluate", "--learn-rate", "--batch-size",
                                         "--batch-threads", "--seed", "--num-parallel-calls", "--report-interval",
                                         "--shuffle", "--clip-norm"]),
                      ("nnet-validate.py", ["--objective", "--evaluate", "--batch-size", "--report-interval"]),
                      ("nnet-init.py", ["--objective", "--evaluate", "--batch-size"]),
                      ("nnet-forward.py", ["--apply-softmax", "--apply-log", "--report-interval", "--class-prior",
                                           "--smooth-factor"])):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bin", cli), "--help"], capture_output=True,
                             text=True, timeout=120)
        assert out.returncode == 0, out.stderr
        for flag in must:
            assert flag in out.stdout, (cli, flag)
