"""CPU-side tests of the host logic and of the C-ABI surface (no compute calls without a GPU)."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_every_declared_symbol():
    """The .so must load on a GPU-less box and export every function include/lstm_ctc_hip.h declares."""
    import __graft_entry__ as g
    g.build()
    from lstm_ctc_amd import _lib
    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "lstm_ctc_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(lc_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 18
    for name in declared:
        assert hasattr(lib, name), "missing export: " + name
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert lib.lc_version() >= 1


def test_ops_refuse_cpu_tensors():
    import torch
    from lstm_ctc_amd import ops, _lib
    with pytest.raises(_lib.LibraryError):
        ops.gemm(torch.zeros(4, 4), torch.zeros(4, 4))


def test_edit_distance_host_entry(oracle):
    from lstm_ctc_amd import ops
    hyp = np.array([[1, 2, 3, 0], [0, 0, 0, 0], [5, 6, 0, 0]], np.int32)
    flat = np.array([1, 3, 7, 8, 9, 5, 6], np.int32)
    offs = np.array([0, 2, 5, 7], np.int32)
    d = ops.edit_distance_host(hyp, [3, 0, 2], flat, offs)
    assert list(d) == [1, 3, 0]
    assert list(d) == list(oracle.edit_distance(hyp, [3, 0, 2], flat, offs))


WSJ_CONFIG = """nnet_type = blstm
input_dim = 120
left_context = 1
right_context = 1
subsample = 3
num_layers = 4
num_neurons = 320
num_projects = 320
num_targets = 72
use_peepholes = true
use_bn = false
dropout_rate = 0.9
num_experts = 72
moe_temp = 10.0
seed = 777
uniform_label_sm = 0
prior_label_sm =  0
prior_label_path = exp/label.counts
"""


def test_parse_config_recipe(tmp_path):
    """The WSJ recipe's nnet.config (egs/wsj/run_wsj_phn.sh:226-243, incl. the double space)."""
    from lstm_ctc_amd.nnet import parse_config
    f = tmp_path / "nnet.config"
    f.write_text(WSJ_CONFIG + "# comment line\n")
    c = parse_config(str(f))
    assert c == dict(nnet_type="blstm", input_dim=120, left_context=1, right_context=1, subsample=3,
                     num_layers=4, num_neurons=320, num_projects=320, num_targets=72, use_peepholes=True,
                     use_bn=False, dropout_rate=0.9, num_experts=72, moe_temp=10.0, seed=777,
                     uniform_label_sm=0, prior_label_sm=0, prior_label_path="exp/label.counts")
    assert isinstance(c["input_dim"], int) and isinstance(c["moe_temp"], float) and c["use_peepholes"] is True


def test_class_prior(tmp_path):
    """[ 10 5 0 85 ] -> log prior with the blank (index 0) rotated to the end, -1e10 for zero counts."""
    from lstm_ctc_amd.nnet import get_class_prior
    f = tmp_path / "label.counts"
    f.write_text(" [ 10 5 0 85 ]\n")
    p = get_class_prior(str(f))
    assert p.dtype == np.float32
    np.testing.assert_allclose(p, [np.log(0.05), -1e10, np.log(0.85), np.log(0.10)], rtol=1e-6)


def test_param_store_layout_cpu():
    from lstm_ctc_amd.nnet.model import ParamStore, gate_perm
    cfg = dict(nnet_type="blstm", input_dim=5, left_context=1, right_context=1, num_layers=2, num_neurons=16,
               num_projects=8, num_targets=7, use_peepholes=True, dropout_rate=1.0)
    ps = ParamStore(cfg, "cpu")
    ps.init_random(0)
    assert ps.D == 15 and ps.shapes["fd0/frnn0/kernel"] == (15 + 8, 64)
    assert ps.shapes["fd1/frnn1/kernel"] == (16 + 8, 64)
    tf = ps.export_tf()
    ps2 = ParamStore(cfg, "cpu")
    ps2.load_tf(tf)
    assert (ps.flat == ps2.flat).all()
    perm = gate_perm(16)
    assert sorted(perm) == list(range(64))
    # interleaved column (blk=1, gate=2, unit 3 of the block) is TF column 2*N + 8 + 3
    assert perm[1 * 32 + 2 * 8 + 3] == 2 * 16 + 11
    k_int = ps.p("fd0/frnn0/kernel").numpy()
    assert np.array_equal(k_int[:, 1 * 32 + 2 * 8 + 3], tf["fd0/frnn0/kernel"][:, 2 * 16 + 11])
    assert all(("bias" in n) == (ps.offsets[n] >= ps.n_decay) for n in ps.names())


def test_bf16x3_product_rule():
    """Which products the mode sends to its 256 x 256 kernels: c4's, c3's and (round 5: measured, 1.4 - 1.56 x the fp32 kernels
    at 2.44 and 1.46 rounds) c2's big ones, not the K = 40 input layer, not the 44-wide head, not a product of a few dozen
    tiles; weight gradients with few tiles only when K can be sliced."""
    from lstm_ctc_amd.nnet.model import _x3_pays
    assert _x3_pays(64000, 4096, 2048) and _x3_pays(64000, 2048, 4096) and _x3_pays(32000, 2048, 1024)
    assert _x3_pays(32000, 1280, 640) and _x3_pays(32000, 640, 1280)                     # c2: 81 % and 73 % of their rounds
    assert not _x3_pays(64000, 4096, 40) and not _x3_pays(64000, 44, 2048) and not _x3_pays(8000, 640, 1280)
    assert not _x3_pays(30000, 256, 4096)                                                # 118 tiles: under half a round
    assert _x3_pays(2048, 4096, 64000, split_k=True) and _x3_pays(1024, 1024, 64000, split_k=True)
    assert not _x3_pays(1024, 1024, 2000, split_k=True)


def test_split_operand_forward_recurrence_width_rule():
    """Round 5: with its operands requested a step ahead the fp32 forward recurrence is the faster one up to 320 units (1.77
    against 2.11 us per step at c2's width, 1.44 / 1.72 at 256; 2.86 against 2.56 at 512 - `profiles/r5_persist_probe_ahead.txt`),
    (round 6: 1.99 / 2.14 at 384, 2.64 / 2.70 at 448 - `profiles/r6_x3_width_probe.txt`),
    so bf16x3 mode takes the split-operand FORWARD kernel above 448 units only (`LC_X3_FWD_MIN_N` overrides; the BPTT is the
    split-operand kernel at every width it exists for).  The GPU side of the rule: `tests/test_gpu_configs.py::_x3_kind`."""
    import os
    from lstm_ctc_amd.nnet import model as model_mod
    assert model_mod.X3_FWD_MIN_N == int(os.environ.get("LC_X3_FWD_MIN_N", "448"))
    if "LC_X3_FWD_MIN_N" not in os.environ:
        assert [model_mod.x3_forward_recurrence(n) for n in (64, 256, 320, 384, 448, 512)] == [False, False, False, False, False, True]
    # (what Model.forward hands to ops.lstm_fwd at each width: tests/test_gpu_round6.py::test_x3_forward_width_rule_by_behaviour)


def test_gemm_whole_round_rule():
    """The tail rule of the 256 x 256 product kernels (host arithmetic in the library, no GPU): whole rounds of 256 CUs on
    the big kernel, the rows behind them on the 128 x 128 kernel - where that is cheaper by the rule's cost model."""
    from lstm_ctc_amd import _lib
    lib = _lib.load()
    assert lib.lc_debug_gemm_whole_round_row_tiles(250, 16, 256) == 250      # default: off (see csrc/gemm.hip for why)
    assert lib.lc_set_option(b"gemm_tail", 1) == 0
    try:
        _whole_round_rule_cases(lambda r, c: lib.lc_debug_gemm_whole_round_row_tiles(r, c, 256))
    finally:
        lib.lc_set_option(b"gemm_tail", _lib.OPTION_UNSET)


def _whole_round_rule_cases(f):
    assert f(250, 16) == 240            # c4 / c5 zx: 4000 tiles = 15.6 rounds -> 15 whole rounds + 10 row tiles of tail
    assert f(249, 16) == 240            # T = 999
    assert f(125, 5) == 102             # c2 zx: 625 tiles = 2.44 rounds -> 510 tiles (99.6 % of 2 rounds) + tail
    assert f(250, 8) == 250             # c4 dX: 7.8 rounds - the last round is 81 % full, a split does not pay
    assert f(250, 4) == 250 and f(125, 8) == 125 and f(125, 20) == 125
    assert f(256, 16) == 256            # whole rounds already
    assert f(10, 16) == 10 and f(1, 1) == 1 and f(0, 5) == 0      # under two rounds / degenerate: unchanged
    for r in range(1, 400, 7):          # never more than there is, never an empty big part
        for c in (1, 2, 3, 5, 8, 16, 20):
            k = f(r, c)
            assert 1 <= k <= r
            if k < r:
                assert (k * c) % 256 == 0 or (k * c) % 256 >= 0.97 * 256
