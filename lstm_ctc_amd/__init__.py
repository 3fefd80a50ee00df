"""lstm_ctc_amd — MI355X-native implementation of the mobvoi/lstm_ctc training hot path.

The compute lives in ``liblstm_ctc_hip.so`` (hand-written gfx950 HIP kernels behind the C ABI of
``include/lstm_ctc_hip.h``); this package is the Python host that mirrors the reference's ``nnet``
operator surface (``parse_config``, ``create_graph_for_*``, ``train``/``validate``) on PyTorch-ROCm
tensors.  There is no CPU or eager-PyTorch fallback: importing the ops without the built library,
or calling them without a GPU, raises.
"""
__version__ = "0.1.0"
