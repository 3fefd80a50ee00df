"""The native input path (csrc/tfrecord.cpp behind lc_tfrecord_* / lc_batch_*, host code - no GPU) against the two
independent statements of the same format: the pure-Python wire-format reader of lstm_ctc_amd/nnet/tfrecord.py and the
google.protobuf runtime (through tests/test_tfrecord_protobuf.py's schema).  Byte-identical features, labels, splice /
subsample (reference: nnet/tfrecord.py:28-51, 94-125), padding (nnet/pipeline.py:35-61: 0.0 / -1), CRC verification
(TF's reader raises DataLossError on a mismatch; so does this), error paths, and the parallel pipeline's batches."""
import ctypes
import struct

import numpy as np
import pytest


def _framed(payload):
    from lstm_ctc_amd.nnet import tfrecord as tr
    head = struct.pack("<Q", len(payload))
    return head + struct.pack("<I", tr.masked_crc(head)) + payload + struct.pack("<I", tr.masked_crc(payload))


def _write(tmp_path, name, x, y):
    from lstm_ctc_amd.nnet import tfrecord as tr
    p = str(tmp_path / name)
    tr.write_tfrecord(p, x, y)
    return p


def _python_item(path, l, r, sub):
    """The old pure-Python reader: parse -> stack -> splice -> subsample."""
    from lstm_ctc_amd.nnet import tfrecord as tr
    ex = tr.parse_sequence_example(tr.read_tfrecord(path)[0])
    x = np.stack(ex["nnet_input"]).astype(np.float32)
    if l or r:
        x = tr.splice(x, l, r)
    if sub:
        x = tr.subsample(x, sub)
    return x, np.asarray([int(v[0]) for v in ex.get("nnet_target", [])], np.int64)


def test_crc32c_known_answers():
    """RFC 3720 B.4 test vectors for CRC-32C, native and table-driven Python."""
    from lstm_ctc_amd import _lib
    from lstm_ctc_amd.nnet import tfrecord as tr
    lib = _lib.load()
    vectors = [(b"", 0x00000000), (b"123456789", 0xE3069283), (bytes(32), 0x8A9136AA), (bytes([0xFF] * 32), 0x62A8AB43),
               (bytes(range(32)), 0x46DD794E), (bytes(range(31, -1, -1)), 0x113FDB5C)]
    t = tr._crc_table()
    for data, want in vectors:
        assert lib.lc_crc32c(data, len(data)) == want
        c = 0xFFFFFFFF
        for b in data:
            c = int(t[(c ^ b) & 0xFF]) ^ (c >> 8)
        assert c ^ 0xFFFFFFFF == want
    rng = np.random.default_rng(0)
    for n in (1, 7, 8, 9, 63, 1000, 4097):                      # unaligned heads / tails of the 8-byte loop
        blob = rng.integers(0, 256, n + 3, dtype=np.uint8).tobytes()
        c = 0xFFFFFFFF
        for b in blob[3:]:
            c = int(t[(c ^ b) & 0xFF]) ^ (c >> 8)
        buf = ctypes.create_string_buffer(blob, len(blob))
        assert lib.lc_crc32c(ctypes.addressof(buf) + 3, n) == c ^ 0xFFFFFFFF


@pytest.mark.parametrize("T,D,L,l,r,sub", [(7, 5, 4, 0, 0, 0), (1, 40, 1, 2, 3, 0), (33, 12, 17, 1, 1, 3), (10, 3, 0, 0, 2, 4),
                                            (5, 8, 2, 1, 0, 7), (300, 40, 50, 1, 1, 3)])
def test_native_decoder_equals_python_reader(tmp_path, T, D, L, l, r, sub):
    from lstm_ctc_amd.nnet import tfrecord as tr
    rng = np.random.default_rng(T * 100 + D)
    x = rng.normal(size=(T, D)).astype(np.float32)
    x[0, 0] = np.float32(-0.0)
    y = rng.integers(0, 1 << 40, size=L).astype(np.int64)
    if L > 1:
        y[1] = -3
    p = _write(tmp_path, "u.tfrecords", x, y)
    ds = tr.TFRecordDataset([p], D, True, l, r, sub)
    item = ds.load(p)
    want_x, want_y = _python_item(p, l, r, sub)
    assert item["nnet_input"].shape == want_x.shape and item["sequence_length"] == want_x.shape[0]
    np.testing.assert_array_equal(item["nnet_input"].view(np.uint32), want_x.view(np.uint32))     # bit for bit
    np.testing.assert_array_equal(item["nnet_target"], want_y)
    assert item["target_length"] == L
    # without labels requested the targets are not touched
    assert "nnet_target" not in tr.TFRecordDataset([p], D, False, l, r, sub).load(p)


def test_native_decoder_on_unpacked_encodings_and_context(tmp_path):
    """Repeated scalars one key per element, map entries value-first, a context to skip, an unknown feature list."""
    from lstm_ctc_amd.nnet import tfrecord as tr
    ld = lambda fnum, b: tr._enc_varint((fnum << 3) | 2) + tr._enc_varint(len(b)) + b
    rows = [(1.5, -2.0, 0.25), (3.0, 4.0, -0.0)]
    feats = b"".join(ld(1, ld(2, b"".join(tr._enc_varint((1 << 3) | 5) + struct.pack("<f", v) for v in row)))
                     for row in rows)
    labels = b"".join(ld(1, ld(3, tr._enc_varint((1 << 3) | 0) + tr._enc_varint(v))) for v in (7, 300, (1 << 64) - 2))
    other = ld(1, ld(1, b"speaker") + ld(2, ld(1, ld(1, ld(1, b"abc")))))
    entry_in = ld(2, feats) + ld(1, b"nnet_input")                            # value before key
    entry_tg = ld(1, b"nnet_target") + ld(2, labels)
    context = ld(1, ld(1, ld(1, b"utt") + ld(2, ld(1, ld(1, b"011c0201")))))
    payload = context + ld(2, other + ld(1, entry_in) + ld(1, entry_tg))
    p = tmp_path / "odd.tfrecords"
    p.write_bytes(_framed(payload))
    item = tr.TFRecordDataset([str(p)], 3, True, 0, 0, 0).load(str(p))
    np.testing.assert_array_equal(item["nnet_input"].view(np.uint32), np.asarray(rows, np.float32).view(np.uint32))
    np.testing.assert_array_equal(item["nnet_target"], np.array([7, 300, -2], np.int64))
    want_x, want_y = _python_item(str(p), 0, 0, 0)
    np.testing.assert_array_equal(item["nnet_input"], want_x)
    np.testing.assert_array_equal(item["nnet_target"], want_y)


def test_empty_and_missing_lists(tmp_path):
    from lstm_ctc_amd.nnet import tfrecord as tr
    p = _write(tmp_path, "e.tfrecords", np.zeros((0, 4), np.float32), [])
    item = tr.TFRecordDataset([p], 4, True, 1, 1, 2).load(p)
    assert item["nnet_input"].shape == (0, 12) and item["sequence_length"] == 0 and len(item["nnet_target"]) == 0
    q = _write(tmp_path, "n.tfrecords", np.ones((3, 4), np.float32), None)          # no nnet_target list at all
    item = tr.TFRecordDataset([q], 4, False, 0, 0, 0).load(q)                        # fine for an unlabelled dataset ...
    assert item["nnet_input"].shape == (3, 4) and "nnet_target" not in item
    with pytest.raises(ValueError, match="no nnet_target feature list"):            # ... an error under has_label = 1
        tr.TFRecordDataset([q], 4, True, 0, 0, 0).load(q)


def test_corruption_is_detected(tmp_path):
    """Every single-bit flip in the length word, its CRC, the payload or the payload CRC must be refused: a damaged
    utterance must not train (tf.data raises DataLossError; nnet/tfrecord.py:122)."""
    from lstm_ctc_amd.nnet import tfrecord as tr
    rng = np.random.default_rng(3)
    x = rng.normal(size=(6, 5)).astype(np.float32)
    p = _write(tmp_path, "ok.tfrecords", x, [1, 2, 3])
    good = open(p, "rb").read()
    ds = tr.TFRecordDataset([p], 5, True, 0, 0, 0)
    ds.load(p)
    bad = tmp_path / "bad.tfrecords"
    for pos in [0, 9, 12, 40, len(good) // 2, len(good) - 6, len(good) - 1]:
        blob = bytearray(good)
        blob[pos] ^= 0x10
        bad.write_bytes(bytes(blob))
        with pytest.raises(ValueError):                      # CorruptRecordError is a ValueError
            ds.load(str(bad))
        with pytest.raises(ValueError):
            ds.open_batch([p, str(bad)], 2)
        with pytest.raises(ValueError):
            tr.read_tfrecord(str(bad))
    blob = bytearray(good)
    blob[good.index(x[3].tobytes()) + 5] ^= 0x10             # a flipped float bit: only the CRC can notice
    bad.write_bytes(bytes(blob))
    with pytest.raises(tr.CorruptRecordError):
        ds.load(str(bad))
    lax = tr.TFRecordDataset([str(bad)], 5, True, 0, 0, 0, verify_crc=False)
    assert lax.load(str(bad))["nnet_input"].shape == (6, 5)
    bad.write_bytes(good[:len(good) - 9])                    # truncated file
    with pytest.raises(ValueError):
        ds.load(str(bad))
    with pytest.raises(ValueError):                          # wrong feature dimension
        tr.TFRecordDataset([p], 6, True, 0, 0, 0).load(p)
    with pytest.raises(ValueError, match="missing"):
        ds.open_batch([p, str(tmp_path / "missing.tfrecords")], 2)


@pytest.mark.parametrize("threads,batch_threads", [(1, 1), (4, 2), (32, 8)])
def test_parallel_pipeline_batches(tmp_path, threads, batch_threads):
    """create_pipeline_sequence_batch over the native loader: consecutive runs of batch_size utterances, features padded
    with 0, labels with -1, a smaller last batch (pipeline.py:35-61); the batch is a [B,T,D] VIEW of time-major memory."""
    from lstm_ctc_amd.nnet import create_pipeline_sequence_batch, dataset_from_tfrecords
    rng = np.random.default_rng(11)
    lines, ref = [], []
    for i in range(11):
        T, L = int(rng.integers(3, 40)), int(rng.integers(0, 9))
        x = rng.normal(size=(T, 6)).astype(np.float32)
        y = rng.integers(0, 20, size=L)
        p = _write(tmp_path, "u%d.tfrecords" % i, x, y)
        lines.append("u%d %d 6 1 %s" % (i, T, p))
        ref.append(_python_item(p, 1, 2, 2))
    scp = tmp_path / "tfrecords.scp"
    scp.write_text("\n".join(lines) + "\n")
    _, ds, dim = dataset_from_tfrecords(str(scp), left_context=1, right_context=2, subsample=2, num_parallel_calls=threads)
    assert dim == 24
    _, pipe = create_pipeline_sequence_batch(ds, dim, batch_size=4, batch_threads=batch_threads)
    batches = list(pipe)
    assert [b["nnet_input"].shape[0] for b in batches] == [4, 4, 3]
    k = 0
    for b in batches:
        xs, ys = b["nnet_input"], b["nnet_target"]
        assert xs.dtype == np.float32 and ys.dtype == np.int64 and b["sequence_length"].dtype == np.int32
        assert xs.transpose(1, 0, 2).flags.c_contiguous                  # time-major underneath
        assert xs.shape[1] == max(b["sequence_length"]) and ys.shape[1] == max(b["target_length"])
        for j in range(xs.shape[0]):
            wx, wy = ref[k]
            k += 1
            assert b["sequence_length"][j] == len(wx) and b["target_length"][j] == len(wy)
            np.testing.assert_array_equal(xs[j, :len(wx)], wx)
            assert not xs[j, len(wx):].any()                             # padding value 0
            np.testing.assert_array_equal(ys[j, :len(wy)], wy)
            assert (ys[j, len(wy):] == -1).all()                         # padding value -1
    assert k == 11
    # data parallelism: rank r takes every world-th batch, ragged tail dropped so that all ranks step together
    _, p0 = create_pipeline_sequence_batch(ds, dim, batch_size=4, rank=0, world_size=2)
    _, p1 = create_pipeline_sequence_batch(ds, dim, batch_size=4, rank=1, world_size=2)
    b0, b1 = list(p0), list(p1)
    assert len(b0) == len(b1) == 1
    np.testing.assert_array_equal(b0[0]["nnet_target"], batches[0]["nnet_target"])
    np.testing.assert_array_equal(b1[0]["nnet_target"], batches[1]["nnet_target"])


def test_silent_data_loss_is_refused(tmp_path):
    """Two divergences from tf.data.TFRecordDataset + parse_single_sequence_example (nnet/tfrecord.py:94-125) that used to
    lose data silently: a file with MORE than one record (TF yields every record; this loader maps a file
    to one utterance) and a record without the nnet_target list under has_label = 1 (TF's parse fails)."""
    from lstm_ctc_amd.nnet import create_pipeline_sequence_batch, dataset_from_tfrecords
    from lstm_ctc_amd.nnet import tfrecord as tr
    x = np.ones((4, 3), np.float32)
    one = _framed(tr.serialize_sequence_example(x, [1, 2]))
    (tmp_path / "two.tfrecords").write_bytes(one + one)                   # two records
    (tmp_path / "tail.tfrecords").write_bytes(one + b"\x00" * 5)          # trailing bytes
    (tmp_path / "ok.tfrecords").write_bytes(one)
    for name in ("two", "tail"):
        scp = tmp_path / (name + ".scp")
        scp.write_text("a 4 3 1 %s\n" % (tmp_path / (name + ".tfrecords")))
        _, ds, dim = dataset_from_tfrecords(str(scp))
        with pytest.raises(ValueError, match="behind the first record"):
            ds.load(ds.files[0])
        with pytest.raises(ValueError, match="behind the first record"):
            list(create_pipeline_sequence_batch(ds, dim, batch_size=1)[1])
    assert len(tr.read_tfrecord(str(tmp_path / "two.tfrecords"))) == 2    # (the Python framing reader sees both)
    # no target list
    nolab = _framed(tr.serialize_sequence_example(x, None))
    (tmp_path / "nolab.tfrecords").write_bytes(nolab)
    scp = tmp_path / "nolab.scp"
    scp.write_text("a 4 3 1 %s\nb 4 3 1 %s\n" % (tmp_path / "ok.tfrecords", tmp_path / "nolab.tfrecords"))
    _, ds, dim = dataset_from_tfrecords(str(scp))
    with pytest.raises(ValueError, match="no nnet_target feature list"):
        ds.load(ds.files[1])
    with pytest.raises(ValueError, match="no nnet_target feature list"):
        list(create_pipeline_sequence_batch(ds, dim, batch_size=2)[1])
    scp.write_text("a 4 3 0 %s\n" % (tmp_path / "nolab.tfrecords"))        # has_label = 0: fine
    _, ds, dim = dataset_from_tfrecords(str(scp))
    assert next(iter(create_pipeline_sequence_batch(ds, dim, batch_size=1)[1]))["nnet_target"].shape == (1, 0)


def test_batches_a_consumer_keeps_are_never_overwritten(tmp_path, capfd):
    """The reference's padded_batch hands out arrays the consumer owns (pipeline.py:35-61); here a batch is a view of a
    recycled staging slot.  A consumer that keeps MORE batches than the ring is deep (list(pipe), a cached CV set) must still
    read what it was given: a slot somebody holds a view of is not reused.  A consumer that drops its batches recycles."""
    import gc
    from lstm_ctc_amd.nnet import create_pipeline_sequence_batch, dataset_from_tfrecords
    from lstm_ctc_amd.nnet import pipeline as pl
    rng = np.random.default_rng(5)
    lines, ref = [], []
    for i in range(40):
        T = int(rng.integers(5, 12))
        x = rng.normal(size=(T, 4)).astype(np.float32)
        p = _write(tmp_path, "u%d.tfrecords" % i, x, [i % 7])
        lines.append("u%d %d 4 1 %s" % (i, T, p))
        ref.append(x)
    scp = tmp_path / "tfrecords.scp"
    scp.write_text("\n".join(lines) + "\n")
    _, ds, dim = dataset_from_tfrecords(str(scp), num_parallel_calls=2)
    _, pipe = create_pipeline_sequence_batch(ds, dim, batch_size=2, batch_threads=1)
    kept = list(pipe)                                   # 20 batches; the ring is prefetch + 2 * batch_threads + 3 = 9 deep
    assert len(kept) == 20 > pipe.prefetch + 2 * pipe.batch_threads + 3
    for k, b in enumerate(kept):
        for j in range(2):
            x = ref[2 * k + j]
            np.testing.assert_array_equal(b["nnet_input"][j, :len(x)], x)
    starts = {b["nnet_input"].__array_interface__["data"][0] for b in kept}
    assert len(starts) == 20                            # twenty live batches, twenty disjoint buffers
    # a consumer that lets go of its batches gets the ring's slots back (no allocation per batch)
    ring = pl._HostBuffers(3)
    seen = set()
    for _ in range(12):
        v = ring.take(100)
        seen.add(v.__array_interface__["data"][0])
        del v
        gc.collect()
    assert len(seen) == 3 and ring.pageable_handouts == 0
    # ownership is a lease, not a reference count: a DERIVED view keeps the slot - here the [B, T, D] view a
    # batch is made of, held while the hand-out itself is gone - and the ring never owns more than `depth` buffers: a batch
    # made while its slot is held lives in a plain array of its own
    capfd.readouterr()                                  # (the pipeline above kept 20 batches: its ring has logged already)
    ring = pl._HostBuffers(2)
    a = ring.take(24)
    a[:] = 1.0
    view = a.reshape(2, 3, 4).transpose(1, 0, 2)
    del a
    gc.collect()
    b = ring.take(24)
    c = ring.take(24)                                   # slot 0 again: still leased through `view`
    c[:] = 3.0
    assert ring.pageable_handouts == 1 and (view == 1.0).all()
    slot_ptrs = {s.__array_interface__["data"][0] for s in ring.slots if s is not None}
    assert len(slot_ptrs) == 2 and c.__array_interface__["data"][0] not in slot_ptrs
    del view
    gc.collect()
    d = ring.take(24)                                   # slot 1 (held by b) -> pageable; then slot 0 is free again
    e = ring.take(24)
    assert ring.pageable_handouts == 2 and e.__array_interface__["data"][0] in slot_ptrs
    assert sum(s is not None for s in ring.slots) == 2  # never more than `depth` ring buffers
    # the first pageable hand-out is logged, once (nothing used to say that retained batches lose their pinned uploads)
    err = capfd.readouterr().err
    assert err.count("staging buffers are handed out as pageable memory") == 1


def test_loader_error_reaches_the_consumer(tmp_path):
    from lstm_ctc_amd.nnet import create_pipeline_sequence_batch, dataset_from_tfrecords
    p = _write(tmp_path, "a.tfrecords", np.ones((4, 3), np.float32), [1])
    blob = bytearray(open(p, "rb").read())
    blob[20] ^= 1
    (tmp_path / "b.tfrecords").write_bytes(bytes(blob))
    scp = tmp_path / "tfrecords.scp"
    scp.write_text("a 4 3 1 %s\nb 4 3 1 %s\n" % (p, tmp_path / "b.tfrecords"))
    _, ds, dim = dataset_from_tfrecords(str(scp))
    _, pipe = create_pipeline_sequence_batch(ds, dim, batch_size=1)
    it = iter(pipe)
    assert next(it)["nnet_input"].shape == (1, 4, 3)
    with pytest.raises(ValueError, match="corrupted"):
        next(it)


def test_vectorised_writer_equals_per_row_encoding(tmp_path):
    """serialize_sequence_example assembles the frame list as one byte matrix; the bytes must be those of the
    frame-by-frame encoding (which tests/test_tfrecord_protobuf.py holds against google.protobuf)."""
    from lstm_ctc_amd.nnet import tfrecord as tr
    rng = np.random.default_rng(2)
    for T, D in [(1, 1), (3, 31), (3, 32), (50, 40), (2, 5000)]:         # D = 32 / 5000: multi-byte length varints
        x = rng.normal(size=(T, D)).astype(np.float32)
        y = [5, 0, 1 << 35]
        per_row = b"".join(tr._ld(1, tr._ld(2, tr._ld(1, row.astype("<f4").tobytes()))) for row in x)
        tg = b"".join(tr._ld(1, tr._ld(3, tr._ld(1, tr._enc_varint(int(v))))) for v in y)
        want = tr._ld(2, tr._ld(1, tr._ld(1, b"nnet_input") + tr._ld(2, per_row)) +
                      tr._ld(1, tr._ld(1, b"nnet_target") + tr._ld(2, tg)))
        assert tr.serialize_sequence_example(x, y) == want
