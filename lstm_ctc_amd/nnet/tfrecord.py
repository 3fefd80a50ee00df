"""TFRecord / tf.train.SequenceExample I/O without TensorFlow, plus splice and subsample.

Mirror of mobvoi/lstm_ctc ``nnet/tfrecord.py`` (``_splice`` 28-40, ``_subsample`` 43-51,
``dataset_from_tfrecords`` 54-125, ``write_tfrecord`` 128-156) on the data formats of SURVEY.md App. C:

* ``tfrecords.scp`` line: ``<key> <num_rows> <num_cols> <has_label 0|1> <path>``
* one ``SequenceExample`` per ``.tfrecords`` file with feature lists ``nnet_input`` (T x FloatList[D]) and
  ``nnet_target`` (L x Int64List[1]).

TFRecord framing: ``uint64 len | uint32 masked_crc32c(len) | payload | uint32 masked_crc32c(payload)``.

Two decoders of the same format live here.  The PRODUCT path (``TFRecordDataset.load`` / ``load_into``) calls the native
host functions of liblstm_ctc_hip.so (``lc_tfrecord_inspect`` / ``lc_tfrecord_decode``, csrc/tfrecord.cpp): framing with
both CRC-32C words VERIFIED (as TF's reader does), the SequenceExample walk, splice and subsample fused into one copy
straight into the caller's (batch) buffer, GIL released - so ``--num-parallel-calls`` worker threads really run in
parallel.  The pure-Python wire-format reader below (``parse_sequence_example``) is the readable statement of the same
format; the tests hold the two against each other and against google.protobuf.
"""
import ctypes
import random
import struct
import sys
import time

import numpy as np

from . import tflog
from .. import _lib

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
    """CRC-32C (Castagnoli).  Native (lc_crc32c) when the library is built; the table loop is the fallback for data-prep
    tooling on a box without it."""
    try:
        lib = _lib.load()
    except _lib.LibraryError:
        lib = None
    if lib is not None:
        b = bytes(data)
        return int(lib.lc_crc32c(b, len(b)))
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


class CorruptRecordError(ValueError):
    """A TFRecord whose length / payload CRC does not match (tf.errors.DataLossError in TF's reader)."""


def read_tfrecord(path, verify_crc=True):
    """All records of one TFRecord file (payload bytes), both masked CRC-32C words of every record checked."""
    recs = []
    with open(path, "rb") as f:
        while True:
            head = f.read(12)
            if len(head) < 12:
                break
            (ln,) = struct.unpack("<Q", head[:8])
            payload = f.read(ln)
            tail = f.read(4)
            if verify_crc:
                if struct.unpack("<I", head[8:])[0] != masked_crc(head[:8]):
                    raise CorruptRecordError("%s: corrupted record header (length CRC mismatch)" % path)
                if len(payload) != ln or len(tail) != 4 or struct.unpack("<I", tail)[0] != masked_crc(payload):
                    raise CorruptRecordError("%s: corrupted record (payload CRC mismatch)" % path)
            recs.append(payload)
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


def serialize_sequence_example(nnet_input, nnet_target=None):
    """The SequenceExample bytes the reference's converter hands to TFRecordWriter (tfrecord.py:128-156): feature list
    "nnet_input" = one packed FloatList per frame, "nnet_target" = one Int64List per label.  Every frame has the same
    encoding length, so the whole list is assembled as one [T, pitch] byte matrix instead of a Python loop over frames."""
    nnet_input = np.ascontiguousarray(nnet_input, dtype="<f4")
    T, D = nnet_input.shape if nnet_input.ndim == 2 else (0, 0)
    lists = {}
    if T:
        packed = _enc_varint((1 << 3) | 2) + _enc_varint(4 * D)                     # FloatList.value, packed
        flist = _enc_varint((2 << 3) | 2) + _enc_varint(len(packed) + 4 * D) + packed  # Feature.float_list
        head = np.frombuffer(_enc_varint((1 << 3) | 2) + _enc_varint(len(flist) + 4 * D) + flist, np.uint8)
        rows = np.empty((T, len(head) + 4 * D), np.uint8)
        rows[:, :len(head)] = head
        rows[:, len(head):] = nnet_input.view(np.uint8).reshape(T, 4 * D)
        lists["nnet_input"] = rows.tobytes()
    else:
        lists["nnet_input"] = b""
    if nnet_target is not None:
        lists["nnet_target"] = b"".join(
            _ld(1, _ld(3, _ld(1, _enc_varint(int(v))))) for v in nnet_target)
    fl = b"".join(_ld(1, _ld(1, k.encode()) + _ld(2, v)) for k, v in lists.items())
    return _ld(2, fl)


def write_tfrecord(filename, nnet_input, nnet_target=None):
    """One SequenceExample per file, as the reference's converter writes them (tfrecord.py:128-156)."""
    payload = serialize_sequence_example(nnet_input, nnet_target)
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
def _native():
    return _lib.load()          # raises LibraryError when the library is not built: the loader has no second path


def _raise_native(what):
    msg = _native().lc_last_error().decode("utf-8", "replace")
    raise (CorruptRecordError if "corrupted" in msg else ValueError)("%s%s" % (what, msg))


class _NativeBatch:
    """lc_batch_open / lc_batch_decode / lc_batch_close around one batch of files (all work in native threads)."""

    def __init__(self, ds, paths, nthreads):
        self.ds, self.n, self.nthreads = ds, len(paths), max(1, int(nthreads))
        arr = (ctypes.c_char_p * self.n)(*[p.encode() for p in paths])
        T = np.zeros(self.n, np.int64)
        L = np.zeros(self.n, np.int64)
        self.handle = ctypes.c_void_p()
        rc = _native().lc_batch_open(arr, self.n, int(ds.verify_crc), ds.input_dim, self.nthreads,
                                     ctypes.byref(self.handle), T.ctypes.data, L.ctypes.data)
        if rc != 0:
            self.handle = None
            _raise_native("")
        if ds.has_label and (L < 0).any():
            # tfrecords.scp says has_label = 1: the reference's parse (tfrecord.py:94-105, a FixedLenSequenceFeature without
            # allow_missing) fails on a record that lacks the list - training on it as an utterance with no labels must not
            bad = paths[int(np.argmax(L < 0))]
            self.close()
            raise ValueError("%s: no nnet_target feature list, but tfrecords.scp says has_label = 1" % bad)
        self.frames = (T // ds.sub if ds.sub else T).astype(np.int32)
        self.labels = (np.maximum(L, 0) if ds.has_label else np.zeros_like(L)).astype(np.int32)

    def decode(self, x, utt_stride, row_stride, max_rows, y):
        """x: float32 buffer; utterance i's row j goes to x.flat[i * utt_stride + j * row_stride ...]; y: [n, Lmax] int64
        (padded with -1) or None."""
        ds = self.ds
        try:
            rc = _native().lc_batch_decode(self.handle, ds.l, ds.r, ds.sub, x.ctypes.data, utt_stride, row_stride,
                                           max_rows, y.ctypes.data if y is not None and y.size else None,
                                           y.shape[1] if y is not None else 0, y.shape[1] if y is not None else 0, -1,
                                           self.nthreads)
            if rc != 0:
                _raise_native("")
        finally:
            self.close()

    def close(self):
        if self.handle is not None:
            _native().lc_batch_close(self.handle)
            self.handle = None

    def __del__(self):
        self.close()


class TFRecordDataset:
    """Iterable over the utterances of a tfrecords.scp, yielding the dict the reference's ``_parse`` builds:
    nnet_input [T,D'], sequence_length, and (with labels) nnet_target [L] int64, target_length.

    ``open`` + ``inspect`` (framing, CRCs, counts) and ``decode_into`` (the copy, with splice / subsample) are separate
    so that a batch can be decoded straight into its padded buffer; ``load`` is the two in a row."""

    def __init__(self, files, input_dim, has_label, left_context, right_context, subsample_factor, verify_crc=True):
        self.files, self.input_dim, self.has_label = files, input_dim, has_label
        self.l, self.r, self.sub = left_context or 0, right_context or 0, subsample_factor or 0
        self.verify_crc = verify_crc

    @property
    def out_dim(self):
        return self.input_dim * (1 + self.l + self.r)

    def inspect(self, path):
        """-> (file bytes, frames after subsampling, labels).  Raises CorruptRecordError / ValueError."""
        with open(path, "rb") as f:
            raw = f.read()
        info = _lib.SeqExInfo()
        rc = _native().lc_tfrecord_inspect(raw, len(raw), int(self.verify_crc), ctypes.byref(info))
        if rc != 0:
            msg = _native().lc_last_error().decode("utf-8", "replace")
            raise (CorruptRecordError if "corrupted" in msg else ValueError)("%s: %s" % (path, msg))
        if info.num_frames and info.dim != self.input_dim:
            raise ValueError("%s: feature dim %d, expected %d" % (path, info.dim, self.input_dim))
        if self.has_label and not info.has_target:       # (see _NativeBatch: the reference's parse fails on this)
            raise ValueError("%s: no nnet_target feature list, but tfrecords.scp says has_label = 1" % path)
        T = int(info.num_frames)
        return raw, (T // self.sub if self.sub else T), int(info.num_labels)

    def decode_into(self, path, raw, x, labels):
        """x: float32 [T', D'] view whose LAST axis is contiguous (any row stride: one utterance's rows of a time-major
        batch work); labels: contiguous int64 [L] or None."""
        assert x.dtype == np.float32 and x.ndim == 2 and x.shape[1] == self.out_dim
        assert x.shape[0] == 0 or x.strides[1] == 4
        stride = (x.strides[0] // 4) if x.shape[0] > 1 else max(x.strides[0] // 4, self.out_dim)
        rc = _native().lc_tfrecord_decode(
            raw, len(raw), self.input_dim, self.l, self.r, self.sub,
            x.ctypes.data if x.shape[0] else None, stride, x.shape[0],
            labels.ctypes.data if labels is not None and len(labels) else None, len(labels) if labels is not None else 0)
        if rc != 0:
            raise ValueError("%s: %s" % (path, _native().lc_last_error().decode("utf-8", "replace")))

    def open_batch(self, paths, nthreads=1):
        """Reads and checks ``paths`` with ``nthreads`` native threads -> a _NativeBatch: ``.frames`` (after
        subsampling) and ``.labels`` per utterance, ``.decode(...)`` fills a padded batch and releases the files."""
        return _NativeBatch(self, paths, nthreads)

    def load(self, path):
        raw, T, L = self.inspect(path)
        x = np.empty((T, self.out_dim), np.float32)
        y = np.empty(L, np.int64) if self.has_label else None
        self.decode_into(path, raw, x, y)
        item = {"nnet_input": x, "sequence_length": np.int32(T)}
        if self.has_label:
            item["nnet_target"] = y
            item["target_length"] = np.int32(L)
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
    ds.num_parallel_calls = max(1, int(num_parallel_calls or 1))       # worker threads of the batching pipeline
    return list(files), ds, input_dim * (1 + (left_context or 0) + (right_context or 0))
