"""Cross-check of the TensorFlow-free TFRecord / SequenceExample reader and writer (lstm_ctc_amd/nnet/tfrecord.py) against an
INDEPENDENT protobuf implementation: the google.protobuf runtime with message classes built at run time from the public
tf.train schema (tensorflow/core/example/{feature,example}.proto: BytesList / FloatList / Int64List with field 1, packed;
Feature oneof 1/2/3; Features / FeatureLists maps on field 1; SequenceExample context = 1, feature_lists = 2).

Direction 1: what google.protobuf serialises (the bytes tf.python_io.TFRecordWriter would be handed by the reference's
converter, nnet/tfrecord.py:128-156) is parsed by the product's wire-format reader.  Direction 2: what the product's writer
emits is parsed by google.protobuf.  Covers packed and (hand-built) unpacked repeated fields, negative int64, a context the
reader must skip, empty feature lists and map entries in either order.  CPU only."""
import struct

import numpy as np
import pytest

pb = pytest.importorskip("google.protobuf")
from google.protobuf import descriptor_pb2, descriptor_pool, message_factory  # noqa: E402


def _schema():
    f = descriptor_pb2.FileDescriptorProto()
    f.name, f.package, f.syntax = "lc_test_tf_example.proto", "lc_test_tf", "proto3"
    T = descriptor_pb2.FieldDescriptorProto

    def msg(name):
        m = f.message_type.add()
        m.name = name
        return m

    def field(m, name, number, ftype, label=T.LABEL_OPTIONAL, type_name=None, oneof=None, packed=None):
        x = m.field.add()
        x.name, x.number, x.type, x.label = name, number, ftype, label
        if type_name:
            x.type_name = ".lc_test_tf." + type_name
        if oneof is not None:
            x.oneof_index = oneof
        if packed is not None:
            x.options.packed = packed
        return x

    def map_entry(parent, entry_name, value_type):
        e = parent.nested_type.add()
        e.name = entry_name
        e.options.map_entry = True
        field(e, "key", 1, T.TYPE_STRING)
        field(e, "value", 2, T.TYPE_MESSAGE, type_name=value_type)

    field(msg("BytesList"), "value", 1, T.TYPE_BYTES, T.LABEL_REPEATED)
    field(msg("FloatList"), "value", 1, T.TYPE_FLOAT, T.LABEL_REPEATED, packed=True)
    field(msg("Int64List"), "value", 1, T.TYPE_INT64, T.LABEL_REPEATED, packed=True)
    feat = msg("Feature")
    feat.oneof_decl.add().name = "kind"
    field(feat, "bytes_list", 1, T.TYPE_MESSAGE, type_name="BytesList", oneof=0)
    field(feat, "float_list", 2, T.TYPE_MESSAGE, type_name="FloatList", oneof=0)
    field(feat, "int64_list", 3, T.TYPE_MESSAGE, type_name="Int64List", oneof=0)
    feats = msg("Features")
    map_entry(feats, "FeatureEntry", "Feature")
    field(feats, "feature", 1, T.TYPE_MESSAGE, T.LABEL_REPEATED, type_name="Features.FeatureEntry")
    field(msg("FeatureList"), "feature", 1, T.TYPE_MESSAGE, T.LABEL_REPEATED, type_name="Feature")
    fls = msg("FeatureLists")
    map_entry(fls, "FeatureListEntry", "FeatureList")
    field(fls, "feature_list", 1, T.TYPE_MESSAGE, T.LABEL_REPEATED, type_name="FeatureLists.FeatureListEntry")
    se = msg("SequenceExample")
    field(se, "context", 1, T.TYPE_MESSAGE, type_name="Features")
    field(se, "feature_lists", 2, T.TYPE_MESSAGE, type_name="FeatureLists")
    pool = descriptor_pool.DescriptorPool()
    pool.Add(f)
    return message_factory.GetMessageClass(pool.FindMessageTypeByName("lc_test_tf.SequenceExample"))


@pytest.fixture(scope="module")
def SequenceExample():
    return _schema()


def _example(cls, x, y, with_context=True):
    ex = cls()
    if with_context:                                  # the reference writes none; a reader must skip one
        ex.context.feature["utt"].bytes_list.value.append(b"011c0201")
        ex.context.feature["dur"].float_list.value.append(1.25)
    fl = ex.feature_lists.feature_list["nnet_input"]
    for row in x:
        fl.feature.add().float_list.value.extend(float(v) for v in row)
    if y is not None:
        fl = ex.feature_lists.feature_list["nnet_target"]
        for v in y:
            fl.feature.add().int64_list.value.append(int(v))
    return ex


@pytest.mark.parametrize("T,D,L", [(7, 5, 4), (1, 40, 1), (33, 120, 17), (3, 1, 0)])
def test_reader_parses_protobuf_serialised_sequence_examples(SequenceExample, T, D, L):
    from lstm_ctc_amd.nnet import tfrecord as tr
    rng = np.random.default_rng(T * 1000 + D)
    x = rng.normal(size=(T, D)).astype(np.float32)
    x[0, 0] = np.float32(-0.0)
    y = rng.integers(0, 1 << 40, size=L).astype(np.int64)
    if L > 1:
        y[1] = -3                                     # 10-byte varint
    payload = _example(SequenceExample, x, y).SerializeToString()
    got = tr.parse_sequence_example(payload)
    assert set(got) == {"nnet_input", "nnet_target"}
    np.testing.assert_array_equal(np.stack(got["nnet_input"]).view(np.uint32), x.view(np.uint32))
    assert [int(v[0]) for v in got["nnet_target"]] == [int(v) for v in y]


def test_reader_handles_no_targets_and_unpacked_encodings(SequenceExample):
    from lstm_ctc_amd.nnet import tfrecord as tr
    x = np.arange(6, dtype=np.float32).reshape(2, 3)
    got = tr.parse_sequence_example(_example(SequenceExample, x, None, with_context=False).SerializeToString())
    assert set(got) == {"nnet_input"}
    np.testing.assert_array_equal(np.stack(got["nnet_input"]), x)
    # a proto2-style writer may emit repeated scalars UNPACKED (one key per element) and map entries value-first:
    # both are valid encodings of the same message; google.protobuf accepts them, so must the reader
    ld = lambda fnum, b: tr._enc_varint((fnum << 3) | 2) + tr._enc_varint(len(b)) + b
    floats = b"".join(tr._enc_varint((1 << 3) | 5) + struct.pack("<f", v) for v in (1.5, -2.0))
    ints = b"".join(tr._enc_varint((1 << 3) | 0) + tr._enc_varint(v) for v in (7, 300))
    entry_in = ld(2, ld(1, ld(2, floats))) + ld(1, b"nnet_input")          # value before key
    entry_tg = ld(1, b"nnet_target") + ld(2, ld(1, ld(3, ints)))
    payload = ld(2, ld(1, entry_in) + ld(1, entry_tg))
    ref = SequenceExample.FromString(payload)
    assert list(ref.feature_lists.feature_list["nnet_input"].feature[0].float_list.value) == [1.5, -2.0]
    assert list(ref.feature_lists.feature_list["nnet_target"].feature[0].int64_list.value) == [7, 300]
    got = tr.parse_sequence_example(payload)
    np.testing.assert_array_equal(got["nnet_input"][0], np.array([1.5, -2.0], np.float32))
    np.testing.assert_array_equal(got["nnet_target"][0], np.array([7, 300], np.int64))


def test_writer_output_is_parsed_by_protobuf(SequenceExample, tmp_path):
    from lstm_ctc_amd.nnet import tfrecord as tr
    rng = np.random.default_rng(5)
    x = rng.normal(size=(11, 40)).astype(np.float32)
    y = np.array([3, 0, 71, 2, 1 << 33], np.int64)
    p = str(tmp_path / "u.tfrecords")
    tr.write_tfrecord(p, x, y)
    (payload,) = tr.read_tfrecord(p)
    ex = SequenceExample.FromString(payload)
    assert sorted(ex.feature_lists.feature_list) == ["nnet_input", "nnet_target"]
    rows = [np.array(f.float_list.value, np.float32) for f in ex.feature_lists.feature_list["nnet_input"].feature]
    np.testing.assert_array_equal(np.stack(rows), x)
    assert [f.int64_list.value[0] for f in ex.feature_lists.feature_list["nnet_target"].feature] == list(y)
    # and the bytes are exactly what protobuf itself would write for the same message (deterministic map order)
    assert _example(SequenceExample, x, y, with_context=False).SerializeToString(deterministic=True) == payload
