"""TFRecord / tf.train.SequenceExample I/O without TensorFlow, plus splice and subsample.

Mirror of mobvoi/lstm_ctc ``nnet/tfrecord.py`` (``_splice`` 28-40, ``_subsample`` 43-51,
``dataset_from_tfrecords`` 54-125, ``write_tfrecord`` 128-156) on the data formats of SURVEY.md App. C:

* ``tfrecords.scp`` line: ``<key> <num_rows> <num_cols> <has_label 0|1> <path>``
* one ``SequenceExample`` per ``.tfrecords`` file with feature lists ``nnet_input`` (T x FloatList[D]) and
  ``nnet_target`` (L x Int64List[1]).

TFRecord framing: ``uint64 len | uint32 masked_crc32c(len) | payload | uint32 masked_crc32c(payload)``.
The protobuf payload is decoded by a ~40-line wire-format reader (varint / length-delimited only).
"""
import random
import struct
import sys
import time

import numpy as np

from . import tflog

# ------------------------------------------------------------------------------------------------ crc32c
_CRC_TABLE = None


def _crc_table():
    global _CRC_TABLE
    if _CRC_TABLE is None:
        t = np.zeros(256, np.uint32)
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
            t[i] = c
        _CRC_TABLE = t
    return _CRC_TABLE


def crc32c(data):
    t = _crc_table()
    c = 0xFFFFFFFF
    for b in bytes(data):
        c = int(t[(c ^ b) & 0xFF]) ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def masked_crc(data):
    c = crc32c(data)
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


# ------------------------------------------------------------------------------------------------ protobuf wire format
def _varint(buf, pos):
    out = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        out |= (b & 0x7F) << shift
        if not b & 0x80:
            return out, pos
        shift += 7


def _fields(buf):
    """Yields (field_number, wire_type, value) of one message; value is an int or a memoryview."""
    pos, n = 0, len(buf)
    while pos < n:
        key, pos = _varint(buf, pos)
        fnum, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            v = buf[pos:pos + ln]
            pos += ln
        elif wt == 5:
            v = buf[pos:pos + 4]
            pos += 4
        elif wt == 1:
            v = buf[pos:pos + 8]
            pos += 8
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        yield fnum, wt, v


def _decode_feature(buf):
    """tf.train.Feature -> ('float', np.float32[...]) | ('int64', np.int64[...]) | ('bytes', [...])."""
    for fnum, wt, v in _fields(buf):
        if fnum == 2:                                   # FloatList
            vals = []
            for f2, w2, x in _fields(v):
                if f2 == 1 and w2 == 2:                 # packed
                    vals.append(np.frombuffer(bytes(x), dtype="<f4"))
                elif f2 == 1 and w2 == 5:
                    vals.append(np.frombuffer(bytes(x), dtype="<f4"))
            return "float", (np.concatenate(vals) if vals else np.zeros(0, np.float32))
        if fnum == 3:                                   # Int64List
            vals = []
            for f2, w2, x in _fields(v):
                if f2 == 1 and w2 == 2:
                    p = 0
                    while p < len(x):
                        iv, p = _varint(x, p)
                        vals.append(iv - (1 << 64) if iv >= (1 << 63) else iv)
                elif f2 == 1 and w2 == 0:
                    vals.append(x - (1 << 64) if x >= (1 << 63) else x)
            return "int64", np.asarray(vals, np.int64)
        if fnum == 1:
            return "bytes", [bytes(x) for f2, w2, x in _fields(v) if f2 == 1]
    return "float", np.zeros(0, np.float32)


def parse_sequence_example(payload):
    """Returns {feature_list_name: list of per-step arrays} for a serialized tf.train.SequenceExample."""
    out = {}
    buf = memoryview(payload)
    for fnum, wt, v in _fields(buf):
        if fnum != 2:                                   # 1 = context (unused by the reference)
            continue
        for f2, w2, entry in _fields(v):                # FeatureLists.feature_list map entries
            name, flist = None, None
            for f3, w3, x in _fields(entry):
                if f3 == 1:
                    name = bytes(x).decode()
                elif f3 == 2:
                    flist = x
            steps = []
            if flist is not None:
                for f4, w4, feat in _fields(flist):     # FeatureList.feature
                    if f4 == 1:
                        steps.append(_decode_feature(feat)[1])
            out[name] = steps
    return out


def read_tfrecord(path):
    """All records of one TFRecord file (payload bytes).  CRCs are not verified."""
    recs = []
    with open(path, "rb") as f:
        while True:
            head = f.read(12)
            if len(head) < 12:
                break
            (ln,) = struct.unpack("<Q", head[:8])
            recs.append(f.read(ln))
            f.read(4)
    return recs


# ------------------------------------------------------------------------------------------------ writer (tests, data prep)
def _enc_varint(v):
    v &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _ld(fnum, payload):
    return _enc_varint((fnum << 3) | 2) + _enc_varint(len(payload)) + payload


def write_tfrecord(filename, nnet_input, nnet_target=None):
    """One SequenceExample per file, as the reference's converter writes them (tfrecord.py:128-156)."""
    nnet_input = np.asarray(nnet_input, np.float32)
    lists = {}
    lists["nnet_input"] = b"".join(
        _ld(1, _ld(2, _ld(1, row.astype("<f4").tobytes()))) for row in nnet_input)
    if nnet_target is not None:
        lists["nnet_target"] = b"".join(
            _ld(1, _ld(3, _ld(1, _enc_varint(int(v))))) for v in nnet_target)
    fl = b"".join(_ld(1, _ld(1, k.encode()) + _ld(2, v)) for k, v in lists.items())
    payload = _ld(2, fl)
    head = struct.pack("<Q", len(payload))
    with open(filename, "wb") as f:
        f.write(head + struct.pack("<I", masked_crc(head)) + payload + struct.pack("<I", masked_crc(payload)))


# ------------------------------------------------------------------------------------------------ splice / subsample
def splice(x, left_context, right_context):
    """Frame splicing with edge replication (tfrecord.py:28-40): row t = [x[t-l] .. x[t] .. x[t+r]]."""
    T = x.shape[0]
    padded = np.concatenate([np.repeat(x[:1], left_context, 0), x, np.repeat(x[-1:], right_context, 0)], 0)
    return np.concatenate([padded[i:i + T] for i in range(left_context + right_context + 1)], axis=1)


def subsample(x, factor):
    """Keeps frames 0, f, 2f, ... — floor(T/f) of them (tfrecord.py:43-51)."""
    return x[np.arange(x.shape[0] // factor) * factor]


# ------------------------------------------------------------------------------------------------ dataset
class TFRecordDataset:
    """Iterable over the utterances of a tfrecords.scp, yielding the dict the reference's ``_parse`` builds:
    nnet_input [T,D'], sequence_length, and (with labels) nnet_target [L] int64, target_length."""

    def __init__(self, files, input_dim, has_label, left_context, right_context, subsample_factor):
        self.files, self.input_dim, self.has_label = files, input_dim, has_label
        self.l, self.r, self.sub = left_context or 0, right_context or 0, subsample_factor or 0

    def load(self, path):
        ex = parse_sequence_example(read_tfrecord(path)[0])
        x = np.stack(ex["nnet_input"]).astype(np.float32) if ex.get("nnet_input") else np.zeros((0, self.input_dim), np.float32)
        if x.shape[1] != self.input_dim:
            raise ValueError("%s: feature dim %d, expected %d" % (path, x.shape[1], self.input_dim))
        if self.l or self.r:
            x = splice(x, self.l, self.r)
        if self.sub:
            x = subsample(x, self.sub)
        item = {"nnet_input": x, "sequence_length": np.int32(x.shape[0])}
        if self.has_label:
            y = np.asarray([int(v[0]) for v in ex.get("nnet_target", [])], np.int64)
            item["nnet_target"] = y
            item["target_length"] = np.int32(len(y))
        return item

    def __len__(self):
        return len(self.files)

    def __iter__(self):
        for path in self.files:
            yield self.load(path)


def dataset_from_tfrecords(tfrecords_scp, left_context=0, right_context=0, subsample=0, shuffle=False, seed=None,
                           num_parallel_calls=32):
    """Returns (filename list, dataset, input_dim incl. context) — tfrecord.py:54-125."""
    files, input_dim, has_label = [], None, None
    for line in open(tfrecords_scp, "r"):
        token = line.rstrip().split()
        if not token:
            continue
        num_cols, has_label_, path = int(token[2]), int(token[3]), token[4]
        files.append(path)
        input_dim = num_cols if input_dim is None else input_dim
        has_label = has_label_ if has_label is None else has_label
        if input_dim != num_cols:
            tflog.fatal("inconsistent nnet_input dimension in tfrecords: %d vs. %d" % (input_dim, num_cols))
            sys.exit(1)
        if has_label != has_label_:
            tflog.fatal("inconsistent has_label in tfrecords: %d vs. %d" % (has_label, has_label_))
            sys.exit(1)
    if shuffle:                                                         # permutes the FILE list (tfrecord.py:87-91)
        random.seed(time.time() if seed is None else seed)
        random.shuffle(files)
    ds = TFRecordDataset(files, input_dim, bool(has_label), left_context, right_context, subsample)
    return list(files), ds, input_dim * (1 + (left_context or 0) + (right_context or 0))
