"""TensorFlow tensor-bundle checkpoints without TensorFlow (lstm_ctc_amd/nnet/tf_checkpoint.py): the table / bundle
reader against the writer, against hand-built snappy streams and against the format's constants.  CPU only.  (No
TensorFlow-written file exists in this environment: the reader follows the LevelDB table format and tensor_bundle.proto.)"""
import os
import struct

import numpy as np
import pytest


def _tensors(rng, n=40):
    t = {"fd0/frnn0/lstm_cell/kernel": rng.normal(size=(72, 128)).astype(np.float32),
         "fd0/frnn0/lstm_cell/bias": rng.normal(size=128).astype(np.float32),
         "global_step": np.asarray(12345, np.int64), "Variable": rng.normal(size=(64, 9)).astype(np.float32),
         "flags": np.array([True, False, True]), "counts": rng.integers(-5, 5, size=(3, 2, 4)).astype(np.int32),
         "empty": np.zeros((0, 7), np.float32), "wide": rng.normal(size=(3, 5)).astype(np.float64)}
    for i in range(n):                                   # enough entries for several data blocks and shared key prefixes
        t["bd%d/brnn%d/lstm_cell/projection/kernel" % (i, i)] = rng.normal(size=(i % 5 + 1, 3)).astype(np.float32)
    return t


def test_bundle_round_trip(tmp_path):
    from lstm_ctc_amd.nnet import tf_checkpoint as tc
    rng = np.random.default_rng(0)
    want = _tensors(rng)
    prefix = str(tmp_path / "nnet.3")
    tc.write_bundle(prefix, want, block_bytes=512)
    assert tc.is_bundle(prefix) and sorted(os.listdir(tmp_path)) == ["nnet.3.data-00000-of-00001", "nnet.3.index"]
    got = tc.read_bundle(prefix)
    assert sorted(got) == sorted(want)
    for k in want:
        assert got[k].dtype == want[k].dtype and got[k].shape == want[k].shape, k
        np.testing.assert_array_equal(got[k], want[k])
    keys = [k for k, _ in tc.read_table(prefix + ".index")]
    assert keys[0] == b"" and keys == sorted(keys)                       # header first, table sorted
    raw = open(prefix + ".index", "rb").read()
    assert struct.unpack("<Q", raw[-8:])[0] == 0xdb4775248b80fb57 == tc.MAGIC


def test_corruption_is_detected(tmp_path):
    from lstm_ctc_amd.nnet import tf_checkpoint as tc
    rng = np.random.default_rng(1)
    prefix = str(tmp_path / "m")
    tc.write_bundle(prefix, _tensors(rng, 4))
    data = bytearray(open(prefix + ".data-00000-of-00001", "rb").read())
    data[100] ^= 4
    open(prefix + ".data-00000-of-00001", "wb").write(bytes(data))
    with pytest.raises(tc.BundleError, match="checksum"):
        tc.read_bundle(prefix)
    assert tc.read_bundle(prefix, verify=False)                          # the bytes are still there
    tc.write_bundle(prefix, _tensors(rng, 4))
    idx = bytearray(open(prefix + ".index", "rb").read())
    idx[10] ^= 1
    open(prefix + ".index", "wb").write(bytes(idx))
    with pytest.raises(tc.BundleError, match="checksum"):
        tc.read_bundle(prefix)
    open(prefix + ".index", "wb").write(bytes(idx[:-3]))
    with pytest.raises(tc.BundleError, match="magic"):
        tc.read_bundle(prefix)


def test_snappy_blocks():
    """Index blocks may be snappy-compressed (type byte 1): literals of every length class, copies with 1- / 2- / 4-byte
    offsets, an overlapping (run-length) copy; a stream that lies about its length is refused."""
    from lstm_ctc_amd.nnet import tf_checkpoint as tc
    lit = lambda b: (bytes([(len(b) - 1) << 2]) if len(b) <= 60 else bytes([60 << 2, len(b) - 1]) if len(b) <= 256
                     else bytes([61 << 2]) + (len(b) - 1).to_bytes(2, "little")) + b
    text = b"the quick brown fox "
    s = lit(text)                                                  # 20 literal bytes
    s += bytes([((8 - 4) << 2) | 1 | (0 << 5), 20])                # copy 8 bytes from 20 back (1-byte offset form)
    s += bytes([((5 - 1) << 2) | 2]) + (10).to_bytes(2, "little")  # copy 5 bytes from 10 back (2-byte offset)
    s += bytes([((3 - 1) << 2) | 3]) + (33).to_bytes(4, "little")  # copy 3 bytes from 33 back (4-byte offset)
    s += lit(b"ab") + bytes([((10 - 1) << 2) | 2]) + (2).to_bytes(2, "little")     # overlapping: "ab" repeated 5 times
    big = bytes(range(256)) * 2
    s += lit(big[:200]) + lit(big[:300])
    want = bytearray(text)
    want += want[-20:-12]
    want += want[-10:-5]
    want += want[-33:-30]
    want += b"ab" + b"ab" * 5
    want += big[:200] + big[:300]
    stream = bytes(tc._enc_varint(len(want))) + s
    assert tc.snappy_decompress(stream) == bytes(want)
    with pytest.raises(tc.BundleError):
        tc.snappy_decompress(bytes(tc._enc_varint(len(want) + 1)) + s)
    with pytest.raises(tc.BundleError):
        tc.snappy_decompress(bytes(tc._enc_varint(4)) + bytes([((4 - 1) << 2) | 2, 9, 0]))      # copy from before the start


def test_load_params_accepts_a_saver_prefix(tmp_path):
    """graph.load_params: <nnet-in> may be the prefix of a TF Saver checkpoint (reference bin/nnet-train.py:83,97)."""
    torch = pytest.importorskip("torch")
    from lstm_ctc_amd.nnet import tf_checkpoint as tc
    from lstm_ctc_amd.nnet.graph import load_params

    class _PS:                                   # the two calls load_params makes on a ParamStore
        def load_tf(self, params):
            self.got = params

    rng = np.random.default_rng(2)
    want = _tensors(rng, 3)
    prefix = str(tmp_path / "nnet.7")
    tc.write_bundle(prefix, want)
    ps = _PS()
    load_params(ps, prefix)
    assert sorted(ps.got) == sorted(want)
    np.testing.assert_array_equal(ps.got["Variable"], want["Variable"])
