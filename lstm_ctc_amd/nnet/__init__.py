"""Host-side mirror of the reference's ``nnet`` package surface (nnet/__init__.py:15-26).

``parse_config`` and ``get_class_prior`` are pure host code; everything that computes goes through
liblstm_ctc_hip.so and is imported lazily so that CPU-only tooling can still read configs.
"""
from .config import parse_config
from .class_prior import get_class_prior

from .pipeline import create_pipeline_sequence_batch, create_pipeline_sequential
from .tfrecord import dataset_from_tfrecords, write_tfrecord

__all__ = ["parse_config", "get_class_prior", "train", "validate", "create_graph_for_inference",
           "create_graph_for_training_ctc", "create_graph_for_validation_ctc", "Session", "get_create_logits",
           "create_logits_blstm", "create_logits_lstm", "create_logits_cudnnlstm",
           "create_pipeline_sequence_batch", "create_pipeline_sequential", "dataset_from_tfrecords",
           "write_tfrecord"]


def __getattr__(name):
    if name in ("train", "validate"):
        from . import funcs
        return getattr(funcs, name)
    if name in ("create_graph_for_inference", "create_graph_for_training_ctc", "create_graph_for_validation_ctc",
                "Session", "OutOfRangeError", "get_create_logits", "create_logits_blstm", "create_logits_lstm",
                "create_logits_cudnnlstm"):
        from . import graph
        return getattr(graph, name)
    raise AttributeError(name)
