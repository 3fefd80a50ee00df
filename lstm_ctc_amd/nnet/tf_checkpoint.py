"""TensorFlow "tensor bundle" checkpoints (what ``tf.train.Saver.save`` writes: ``<prefix>.index`` +
``<prefix>.data-00000-of-00001``) read and written WITHOUT TensorFlow, so that models trained by the reference
(``saver = tf.train.Saver(tf.trainable_variables())``, bin/nnet-train.py:83,97; scripts/train.sh ``--srcdir`` flow) can be
continued or decoded here: ``graph.load_params`` falls back to this reader when ``<nnet-in>`` is a bundle prefix.

Format (tensorflow/core/util/tensor_bundle + tensorflow/core/lib/io/table, a port of LevelDB's table format):

* ``.index`` is a sorted string table.  Blocks of prefix-compressed entries ``varint32 shared | varint32 non_shared |
  varint32 value_len | key suffix | value``, followed by the restart array (uint32 each) and its length; every block has a
  5-byte trailer: compression type (0 none, 1 snappy) + masked CRC-32C of (contents + type).  The 48-byte footer holds the
  metaindex and index block handles (varint64 offset, size) and the magic 0xdb4775248b80fb57.
* key ``""`` -> ``BundleHeaderProto`` (num_shards, endianness, version); every other key is a variable name ->
  ``BundleEntryProto`` {1 dtype, 2 shape {2 dim {1 size}}, 3 shard_id, 4 offset, 5 size, 6 fixed32 masked crc32c}.
* the data shard holds the raw little-endian tensor bytes at [offset, offset + size).

Written from the format's definition; TensorFlow is not installable in this environment, so the reader is verified
against the writer below, hand-built snappy streams and the format constants - NOT against a TensorFlow-written file
(DESIGN.md section 7).
"""
import os
import struct

import numpy as np

from .tfrecord import crc32c, _enc_varint, _fields, _varint

MAGIC = 0xdb4775248b80fb57
_DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 9: np.int64, 4: np.uint8, 6: np.int8, 5: np.int16, 10: np.bool_}
_DTYPE_IDS = {np.dtype(v): k for k, v in _DTYPES.items()}


def _mask(c):
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


class BundleError(ValueError):
    pass


# ------------------------------------------------------------------------------------------------ snappy (decompress only)
def snappy_decompress(buf):
    """Raw snappy block format: varint32 uncompressed length, then literals (tag & 3 == 0) and copies with 1- / 2- /
    4-byte offsets (tags 1 / 2 / 3).  Overlapping copies repeat bytes, as the format demands."""
    buf = bytes(buf)
    n, pos = _varint(buf, 0)
    out = bytearray()
    while pos < len(buf):
        tag = buf[pos]
        pos += 1
        kind = tag & 3
        if kind == 0:
            ln = tag >> 2
            if ln >= 60:
                nb = ln - 59
                ln = int.from_bytes(buf[pos:pos + nb], "little")
                pos += nb
            ln += 1
            out += buf[pos:pos + ln]
            pos += ln
            continue
        if kind == 1:
            ln = ((tag >> 2) & 7) + 4
            off = ((tag >> 5) << 8) | buf[pos]
            pos += 1
        elif kind == 2:
            ln = (tag >> 2) + 1
            off = int.from_bytes(buf[pos:pos + 2], "little")
            pos += 2
        else:
            ln = (tag >> 2) + 1
            off = int.from_bytes(buf[pos:pos + 4], "little")
            pos += 4
        if off == 0 or off > len(out):
            raise BundleError("corrupt snappy stream (copy offset %d at output %d)" % (off, len(out)))
        for _ in range(ln):                       # byte by byte: the source may overlap the destination
            out.append(out[-off])
    if len(out) != n:
        raise BundleError("corrupt snappy stream (%d bytes, header says %d)" % (len(out), n))
    return bytes(out)


# ------------------------------------------------------------------------------------------------ table reader
def _read_block(data, offset, size, verify=True):
    raw = data[offset:offset + size]
    trailer = data[offset + size:offset + size + 5]
    if len(raw) != size or len(trailer) != 5:
        raise BundleError("truncated table block at %d" % offset)
    ctype = trailer[0]
    if verify and struct.unpack("<I", trailer[1:])[0] != _mask(crc32c(raw + trailer[:1])):
        raise BundleError("table block at %d: checksum mismatch" % offset)
    if ctype == 1:
        raw = snappy_decompress(raw)
    elif ctype != 0:
        raise BundleError("table block at %d: unknown compression type %d" % (offset, ctype))
    return raw


def _block_entries(block):
    """(key, value) pairs of one block, in order."""
    (nrestarts,) = struct.unpack("<I", block[-4:])
    end = len(block) - 4 - 4 * nrestarts
    pos, key = 0, b""
    while pos < end:
        shared, pos = _varint(block, pos)
        non_shared, pos = _varint(block, pos)
        vlen, pos = _varint(block, pos)
        key = key[:shared] + block[pos:pos + non_shared]
        pos += non_shared
        yield key, block[pos:pos + vlen]
        pos += vlen


def _handle(buf, pos):
    off, pos = _varint(buf, pos)
    size, pos = _varint(buf, pos)
    return off, size, pos


def read_table(path, verify=True):
    """All (key, value) pairs of a table file, in key order."""
    with open(path, "rb") as f:
        data = f.read()
    if len(data) < 48 or struct.unpack("<Q", data[-8:])[0] != MAGIC:
        raise BundleError("%s is not a TensorFlow / LevelDB table (bad magic)" % path)
    footer = data[-48:]
    _, _, pos = _handle(footer, 0)                 # metaindex (unused: no filter policy in bundles)
    ioff, isize, _ = _handle(footer, pos)
    out = []
    for _, hv in _block_entries(_read_block(data, ioff, isize, verify)):
        boff, bsize, _ = _handle(hv, 0)
        out.extend(_block_entries(_read_block(data, boff, bsize, verify)))
    return out


# ------------------------------------------------------------------------------------------------ bundle reader
def _parse_entry(value):
    e = dict(dtype=0, shape=[], shard_id=0, offset=0, size=0, crc=None, sliced=False)
    for fnum, wt, v in _fields(memoryview(value)):
        if fnum == 1 and wt == 0:
            e["dtype"] = v
        elif fnum == 2 and wt == 2:
            for f2, w2, dim in _fields(v):
                if f2 == 2 and w2 == 2:
                    size = 0
                    for f3, w3, x in _fields(dim):
                        if f3 == 1 and w3 == 0:
                            size = x - (1 << 64) if x >= (1 << 63) else x
                    e["shape"].append(size)
        elif fnum == 3 and wt == 0:
            e["shard_id"] = v
        elif fnum == 4 and wt == 0:
            e["offset"] = v
        elif fnum == 5 and wt == 0:
            e["size"] = v
        elif fnum == 6 and wt == 5:
            e["crc"] = struct.unpack("<I", bytes(v))[0]
        elif fnum == 7:
            e["sliced"] = True
    return e


def is_bundle(prefix):
    return os.path.exists(prefix + ".index")


def read_bundle(prefix, verify=True):
    """{variable name: numpy array} of the checkpoint at ``prefix`` (partitioned variables are not supported: the
    reference has none)."""
    entries = read_table(prefix + ".index", verify)
    num_shards = 1
    tensors, shards = {}, {}
    for key, value in entries:
        if key == b"":
            for fnum, wt, v in _fields(memoryview(value)):
                if fnum == 1 and wt == 0:
                    num_shards = v
                elif fnum == 2 and wt == 0 and v != 0:
                    raise BundleError("big-endian checkpoints are not supported")
            continue
        e = _parse_entry(value)
        name = key.decode("utf-8")
        if e["sliced"]:
            raise BundleError("%s: partitioned (sliced) variables are not supported" % name)
        if e["dtype"] not in _DTYPES:
            raise BundleError("%s: unsupported dtype enum %d" % (name, e["dtype"]))
        sid = e["shard_id"]
        if sid not in shards:
            shard_path = "%s.data-%05d-of-%05d" % (prefix, sid, num_shards)
            with open(shard_path, "rb") as f:
                shards[sid] = f.read()
        raw = shards[sid][e["offset"]:e["offset"] + e["size"]]
        dt = np.dtype(_DTYPES[e["dtype"]])
        count = int(np.prod(e["shape"])) if e["shape"] else 1
        if len(raw) != e["size"] or e["size"] != count * dt.itemsize:
            raise BundleError("%s: %d bytes in the data shard, shape %s of %s needs %d" %
                              (name, len(raw), e["shape"], dt, count * dt.itemsize))
        if verify and e["crc"] is not None and e["crc"] != _mask(crc32c(raw)):
            raise BundleError("%s: tensor checksum mismatch" % name)
        tensors[name] = np.frombuffer(raw, dtype=dt.newbyteorder("<")).astype(dt).reshape(e["shape"])
    return tensors


# ------------------------------------------------------------------------------------------------ writer
def _ld(fnum, payload):
    return _enc_varint((fnum << 3) | 2) + _enc_varint(len(payload)) + payload


def _vi(fnum, v):
    return _enc_varint((fnum << 3) | 0) + _enc_varint(v)


def _block(entries, restart_interval=16):
    out, restarts, prev = bytearray(), [], b""
    for i, (k, v) in enumerate(entries):
        shared = 0
        if i % restart_interval == 0:
            restarts.append(len(out))
        else:
            while shared < min(len(prev), len(k)) and prev[shared] == k[shared]:
                shared += 1
        out += _enc_varint(shared) + _enc_varint(len(k) - shared) + _enc_varint(len(v)) + k[shared:] + v
        prev = k
    if not restarts:
        restarts = [0]
    for r in restarts:
        out += struct.pack("<I", r)
    out += struct.pack("<I", len(restarts))
    return bytes(out)


def _emit_block(f, contents):
    off = f.tell()
    f.write(contents + b"\x00" + struct.pack("<I", _mask(crc32c(contents + b"\x00"))))
    return _enc_varint(off) + _enc_varint(len(contents))


def write_bundle(prefix, tensors, block_bytes=4096):
    """Writes ``tensors`` ({name: array}) as a one-shard bundle TensorFlow's BundleReader accepts (uncompressed blocks,
    entries sorted by name, header under the empty key)."""
    names = sorted(tensors)
    data = bytearray()
    entries = [(b"", _vi(1, 1) + _vi(2, 0) + _ld(3, _vi(1, 1)))]          # num_shards 1, LITTLE, version {producer 1}
    for name in names:
        a = np.array(tensors[name], order="C")            # (ascontiguousarray would turn a scalar into shape (1,))
        dt = a.dtype.newbyteorder("=")
        if np.dtype(dt) not in _DTYPE_IDS:
            raise BundleError("%s: dtype %s cannot be written" % (name, a.dtype))
        raw = a.astype(a.dtype.newbyteorder("<")).tobytes()
        shape = b"".join(_ld(2, _vi(1, int(d))) for d in a.shape)
        entry = (_vi(1, _DTYPE_IDS[np.dtype(dt)]) + _ld(2, shape) + (_vi(4, len(data)) if len(data) else b"") +
                 _vi(5, len(raw)) + _enc_varint((6 << 3) | 5) + struct.pack("<I", _mask(crc32c(raw))))
        entries.append((name.encode("utf-8"), entry))
        data += raw
    with open(prefix + ".data-00000-of-00001", "wb") as f:
        f.write(bytes(data))
    with open(prefix + ".index", "wb") as f:
        index, cur, size = [], [], 0
        for k, v in entries:
            cur.append((k, v))
            size += len(k) + len(v) + 3
            if size >= block_bytes:
                index.append((cur[-1][0], _emit_block(f, _block(cur))))
                cur, size = [], 0
        if cur:
            index.append((cur[-1][0], _emit_block(f, _block(cur))))
        meta = _emit_block(f, _block([]))
        idx = _emit_block(f, _block(index, restart_interval=1))
        footer = meta + idx
        f.write(footer + b"\x00" * (40 - len(footer)) + struct.pack("<Q", MAGIC))
