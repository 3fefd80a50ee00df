"""The acoustic model: deep (Bi)LSTM-P stack + affine or high-rank (MoE) head, forward and backward.

Host-side mirror of mobvoi/lstm_ctc ``nnet/bilstm.py:25-273`` (``create_logits_blstm``),
``nnet/lstm.py:125-368`` (``create_logits_lstm``, intent) and ``nnet/moe.py:29-72`` (``create_moe``).
All arithmetic is done by ``liblstm_ctc_hip.so`` through :mod:`lstm_ctc_amd.ops`; this file only
owns buffers and sequencing.

MI355X-first design choices:

* activations are time-major ``[T*B, width]`` row-major matrices, so every batched op is one GEMM;
* ``x_t . Kx`` is hoisted over all T, the projection is folded into the recurrent weights
  (``R = proj . Kh``), so each time step is ONE dependent ``[B,N] x [N,4N]`` GEMM + gate math,
  both directions in the same call (``lc_lstm_fwd`` / ``lc_lstm_bwd``: one launch for both, or one chain per
  direction on two streams);
* ``tf.reverse_sequence`` never materialises: the reverse direction just walks t downwards;
* forward/backward layer outputs land in the two column halves of one ``[T*B, 2P]`` buffer;
* every LSTM ``kernel``/``bias`` lives permanently in the gate-interleaved column layout the step
  kernel wants; TF's ``[i|j|f|o]`` layout only exists at checkpoint I/O (:meth:`ParamStore.export_tf`);
* all parameters (and gradients, and optimizer slots) are single flat fp32 buffers: L2-decayed
  tensors first, the LSTM ``bias`` vectors (the only names containing "bias", nnet/graph.py:185) last.
"""
import math
import os

import numpy as np
import torch

from .. import ops


# variable scope of layer i's cell under tf.nn.dynamic_rnn(MultiRNNCell([CudnnCompatibleLSTMCell ...])) (nnet/lstm.py:73-96)
CUDNN_CELL = "rnn/multi_rnn_cell/cell_%d/cudnn_compatible_lstm_cell"


def gate_perm(N):
    """perm[c'] = c: interleaved column c' = (n//8)*32 + g*8 + n%8  <-  TF column c = g*N + n."""
    assert N % 8 == 0
    cp = np.arange(4 * N)
    blk, rem = cp // 32, cp % 32
    g, i = rem // 8, rem % 8
    return g * N + blk * 8 + i


class ParamStore:
    """Flat fp32 parameter / gradient buffers with named views (TF variable names)."""

    def __init__(self, cfg, device):
        self.cfg = cfg
        self.device = device
        self.blstm = cfg.get("nnet_type", "blstm") == "blstm"
        # 'cudnnlstm' (nnet/lstm.py:26-122): a MultiRNNCell of CudnnCompatibleLSTMCell(num_units) - plain LSTM cells: no
        # peepholes, no projection, forget bias 0, no dropout, no residual; the config's num_projects / use_peepholes are
        # read and logged there but never reach the cell
        self.cudnn = cfg.get("nnet_type") == "cudnnlstm"
        D = cfg["input_dim"] * (1 + (cfg.get("left_context") or 0) + (cfg.get("right_context") or 0))
        N, P, V = cfg["num_neurons"], cfg.get("num_projects"), cfg["num_targets"]
        if self.cudnn:
            if P and P != N:
                # lstm.py:99-102 reshapes the [.., num_neurons] cell output to [-1, num_projects]: anything but
                # num_projects == num_neurons scrambles frames there; refuse instead of reproducing that
                raise ValueError("cudnnlstm: num_projects (%d) must be absent or equal num_neurons (%d): the cell has no "
                                 "projection (nnet/lstm.py:74-76,99)" % (P, N))
            P = None
        self.D, self.N, self.P, self.V = D, N, (P or 0), V
        self.Pout = P if P else N
        # lstm.py:240 hard-codes True for 'lstm'; CudnnCompatibleLSTMCell has none
        self.peep = False if self.cudnn else (True if not self.blstm else bool(cfg.get("use_peepholes") or False))
        self.num_layers = cfg["num_layers"]
        self.E = (cfg.get("num_experts") or 0) if self.blstm else 0
        specs = []       # (name, shape, kind)
        for i in range(self.num_layers):
            if self.blstm:
                I = D if i == 0 else 2 * self.Pout
                prefixes = ["fd%d/frnn%d" % (i, i), "bd%d/brnn%d" % (i, i)]       # bilstm.py:135,156,178,187
            elif self.cudnn:
                I = D if i == 0 else N
                prefixes = [CUDNN_CELL % i]                                        # lstm.py:73-88 (MultiRNNCell scopes)
            else:
                I = D if i == 0 else self.Pout
                prefixes = ["drnn%d/lstm_cell" % i]                                # lstm.py:279-287
            for pre in prefixes:
                specs.append((pre + "/kernel", (I + self.Pout, 4 * N), "lstm_kernel"))
                specs.append((pre + "/bias", (4 * N,), "lstm_bias"))
                if self.peep:
                    for nm in ("w_f_diag", "w_i_diag", "w_o_diag"):
                        specs.append((pre + "/" + nm, (N,), "peephole"))
                if P:
                    specs.append((pre + "/projection/kernel", (N, P), "proj"))
        # tf.layers.batch_normalization of the uni-LSTM (lstm.py:271-294): first-layer input + every layer output
        self.use_bn = bool(cfg.get("use_bn") or False) and not self.blstm and not self.cudnn
        self.bn_names = (["drnn_bn_0_0"] + ["drnn_bn%d" % i for i in range(self.num_layers)]) if self.use_bn else []
        self.aux_specs = []      # non-trainable variables (moving averages): saved/restored, never optimised
        for j, bn in enumerate(self.bn_names):
            C = D if j == 0 else self.Pout
            specs += [(bn + "/gamma", (C,), "bn_gamma"), (bn + "/beta", (C,), "bn_beta")]
            self.aux_specs += [(bn + "/moving_mean", (C,), 0.0), (bn + "/moving_variance", (C,), 1.0)]
        H = (2 if self.blstm else 1) * self.Pout
        self.H = H
        if self.E > 0:                                                            # moe.py:33-58
            specs += [("Variable", (H, self.E), "head_w"), ("Variable_1", (self.E,), "head_b"),
                      ("Variable_2", (H, self.E * V), "head_w"), ("Variable_3", (self.E * V,), "head_b")]
        else:                                                                     # bilstm.py:238-248
            specs += [("Variable", (H, V), "head_w"), ("Variable_1", (V,), "head_b")]
        decayed = [s for s in specs if "bias" not in s[0]]                        # graph.py:183-187
        plain = [s for s in specs if "bias" in s[0]]
        self.specs = decayed + plain
        self.kinds = {s[0]: s[2] for s in self.specs}
        self.shapes = {s[0]: s[1] for s in self.specs}
        self.offsets = {}
        off = 0
        for name, shape, _ in self.specs:
            if name == plain[0][0] if plain else False:
                self.n_decay = off
            self.offsets[name] = off
            off += (int(np.prod(shape)) + 3) // 4 * 4       # keep every tensor 16-byte aligned
        if not plain:
            self.n_decay = off
        self.n = off
        self.flat = torch.zeros(off, dtype=torch.float32, device=device)
        self.grad = torch.zeros(off, dtype=torch.float32, device=device)
        self.aux = {name: torch.full(shape, float(init), dtype=torch.float32, device=device)
                    for name, shape, init in self.aux_specs}
        self._perm = gate_perm(N)
        self._inv = np.argsort(self._perm)

    def names(self):
        return [s[0] for s in self.specs]

    def _view(self, buf, name):
        shape = self.shapes[name]
        o = self.offsets[name]
        return buf[o:o + int(np.prod(shape))].view(*shape)

    def p(self, name):
        return self._view(self.flat, name)

    def g(self, name):
        return self._view(self.grad, name)

    def num_params(self):
        return sum(int(np.prod(s[1])) for s in self.specs)

    # --- TF-layout import / export (checkpoint I/O and parity tests) ---------------------------
    def _to_internal(self, name, arr):
        kind = self.kinds[name]
        if kind == "lstm_kernel":
            return arr[:, self._perm]
        if kind == "lstm_bias":
            return arr[self._perm]
        return arr

    def _to_tf(self, name, arr):
        kind = self.kinds[name]
        if kind == "lstm_kernel":
            return arr[:, self._inv]
        if kind == "lstm_bias":
            return arr[self._inv]
        return arr

    def load_tf(self, params):
        """params: dict name -> numpy array in TF layout (tf.trainable_variables shapes)."""
        missing = [n for n in self.names() if n not in params]
        if missing:
            raise KeyError("checkpoint lacks variables: %s" % missing)
        for name in self.names():
            a = np.asarray(params[name], np.float32)
            if tuple(a.shape) != tuple(self.shapes[name]):
                raise ValueError("shape mismatch for %s: %s vs %s" % (name, a.shape, self.shapes[name]))
            self.p(name).copy_(torch.from_numpy(np.ascontiguousarray(self._to_internal(name, a))))
        for name, shape, _ in self.aux_specs:        # moving averages: restored when the checkpoint has them
            if name in params:
                a = np.asarray(params[name], np.float32)
                if tuple(a.shape) != tuple(shape):
                    raise ValueError("shape mismatch for %s: %s vs %s" % (name, a.shape, shape))
                self.aux[name].copy_(torch.from_numpy(np.ascontiguousarray(a)))

    def export_tf(self, grads=False):
        out = {}
        for name in self.names():
            a = (self.g(name) if grads else self.p(name)).detach().cpu().numpy()
            out[name] = np.ascontiguousarray(self._to_tf(name, a))
        if not grads:
            for name in self.aux:
                out[name] = self.aux[name].detach().cpu().numpy().copy()
        return out

    def init_random(self, seed=None):
        """The reference's initialisers: Glorot-uniform LSTM kernels, peepholes
        and projections (TF variable-scope default), zero biases, truncated-normal heads with
        stddev 1/sqrt(num_neurons) (bilstm.py:239, sic), 1/sqrt(output_dim) (moe.py:33,48; lstm.py:333)."""
        rng = np.random.default_rng(seed)
        params = {}
        for name, shape, kind in self.specs:
            if kind in ("lstm_kernel", "proj", "peephole"):
                fan_in, fan_out = (shape[0], shape[1]) if len(shape) == 2 else (shape[0], shape[0])
                lim = math.sqrt(6.0 / (fan_in + fan_out))
                params[name] = rng.uniform(-lim, lim, size=shape).astype(np.float32)
            elif kind == "head_w":
                if self.cudnn:
                    std = 1.0 / math.sqrt(float(self.N))                           # lstm.py:105
                elif self.E > 0 or not self.blstm:
                    std = 1.0 / math.sqrt(float(self.H))
                else:
                    std = 1.0 / math.sqrt(float(self.N))
                a = rng.normal(0.0, std, size=shape)
                bad = np.abs(a) > 2 * std
                while bad.any():                                  # tf.truncated_normal: re-draw outside 2 sigma
                    a[bad] = rng.normal(0.0, std, size=int(bad.sum()))
                    bad = np.abs(a) > 2 * std
                params[name] = a.astype(np.float32)
            elif kind == "bn_gamma":
                params[name] = np.ones(shape, np.float32)
            else:
                params[name] = np.zeros(shape, np.float32)
        self.load_tf(params)


X3_FWD_MIN_N = int(os.environ.get("LC_X3_FWD_MIN_N", "448"))     # split-operand FORWARD recurrence above this width only


def x3_forward_recurrence(N):
    """bf16x3 mode: whether the FORWARD recurrence of an N-unit layer runs the split-operand kernel (the BPTT always does where
    one exists).  Measured, us per step at B = 32, fp32 kernel with its operands a step ahead / split-operand kernel
    (profiles/r6_x3_width_probe.txt): N = 320 1.78 / 2.12, 384 1.99 / 2.14, 448 2.64 / 2.70, 512 2.90 / 2.55 - the crossover
    lies between 448 and 512 (rounds 4-5 had measured 320 and 512 only and drawn the line at 320)."""
    return N > X3_FWD_MIN_N


X3_FORCE = False          # tests: every eligible product on the bf16x3 kernels, whatever its size
X3_MIN_FILL = int(os.environ.get("LC_X3_MIN_FILL", "45"))     # per cent of whole 256-CU rounds (development knob)


def _x3_pays(M, N, K, split_k=False):
    """Whether an [M, K] x [K, N] product goes to the bf16x3 kernels (256 x 256 tiles, one workgroup per CU): big enough to
    amortise the operand splits, and its tiles (x K slices for the weight gradients) fill the whole rounds of 256 CUs they
    take to at least 45 %.  Round 5 measured the rule instead of inheriting the fp32 256 x 256 kernel's 90 %
    (`profiles/r5_x3_fill_sweep.txt`): a split-operand tile takes ~0.65 of an fp32 tile's time, so c2's 32000 x 1280 x 640
    products (2.44 rounds = 81 %) run 323 us against 505 us on the fp32 kernels and its 32000 x 640 x 1280 ones (1.46 rounds of
    tiles that are 5 / 6 useful) 361 against 498; in the step c2x3 goes from 26.5 to 26.0 ms and c3x3 from 64.9 to 64.0.
    Everything else stays on the fp32 kernels."""
    if X3_FORCE:
        return M >= 1 and N >= 1 and K >= 1
    if K < 64 or 2.0 * M * N * K < 2e10:
        return False
    tiles = -(-M // 256) * -(-N // 256)
    if split_k and tiles < 256:
        return K >= 4096                       # lc_gemm_bf16x3_tn slices K to whole rounds
    rounds = -(-tiles // 256)
    return tiles >= 128 and tiles * 100 >= rounds * 256 * X3_MIN_FILL


class Model:
    """Forward / backward of the stack on one batch.  ``x`` is time-major ``[T,B,D]`` on the GPU,
    zero beyond each utterance's length (the pipeline's padding value, nnet/pipeline.py:44)."""

    def __init__(self, cfg, device="cuda", seed=None):
        self.cfg = dict(cfg)
        self.device = torch.device(device)
        self.ps = ParamStore(self.cfg, self.device)
        self.ps.init_random(seed)
        is_training = self.cfg.get("is_training")
        self.is_training = True if is_training is None else bool(is_training)
        dr = self.cfg.get("dropout_rate")
        self.keep = float(dr) if (dr is not None and self.is_training) else 1.0      # bilstm.py:98-101
        if self.ps.cudnn:
            self.keep = 1.0                                                          # lstm.py:26-122: no DropoutWrapper
        mt = self.cfg.get("moe_temp")
        self.tau = 10.0 if mt is None else float(mt)                                 # bilstm.py:74-76
        self.forget_bias = 5.0 if self.ps.blstm else 1.0                             # bilstm.py:133 / TF default
        if self.ps.cudnn:
            self.forget_bias = 0.0                       # CudnnCompatibleLSTMCell: LSTMBlockCell(forget_bias=0)
        # Extension key (not in the reference): compute_dtype = bf16 selects BASELINE config c5 - every product
        # with an activation operand (gate/projection/head GEMMs, their gradients, the recurrent step) rounds
        # its operands to bf16 and accumulates in fp32; weights, state, CTC and the optimizer stay fp32.
        cd = str(self.cfg.get("compute_dtype") or "fp32").lower()
        if cd not in ("fp32", "float32", "f32", "bf16", "bfloat16", "bf16x3"):
            raise ValueError("compute_dtype must be fp32, bf16x3 or bf16, got %r" % cd)
        self.bf16 = cd in ("bf16", "bfloat16")
        # compute_dtype = bf16x3 (extension): fp32 semantics on the bf16 matrix cores - the products with an activation
        # operand split both fp32 operands exactly into three bf16 terms and accumulate the six significant term products
        # in fp32 (lc_split_bf16x3 + lc_gemm_bf16x3_nt; error against float64 no larger than the fp32 MFMA kernels').
        # The recurrence, CTC, weight-only products and the optimizer are the fp32 mode's.
        self.x3 = cd == "bf16x3"
        # bf16 mode, second stage: operands go through bf16 shadow copies (faster loader); off -> converting loader only
        sh = self.cfg.get("bf16_shadows")
        self.use_shadows = self.bf16 and (True if sh is None else bool(sh))
        self._shadows = {}
        self.saved = None
        # Small fp32 models run each layer's BPTT as one persistent launch that is bound by its exchange latency, not by
        # the matrix pipe: the weight-gradient GEMMs of layer i then run on a second (low-priority)
        # stream under the BPTT of layer i-1.  (For the big model the same overlap was measured twice and lost: its
        # step kernels are MFMA-bound.)  LC_OVERLAP_WGRAD=0 switches it off.
        import os
        self.overlap_wgrad = (not self.bf16 and self.ps.N <= 512 and self.ps.N % 16 == 0
                              and os.environ.get("LC_OVERLAP_WGRAD", "1") != "0")
        # (Wide layers - XCD-pair BPTT, nothing runs beside it -: a layer's weight-gradient GEMMs on the second stream BESIDE
        # its dX GEMMs, joined in front of the next BPTT, so that one product's last partial round of tiles fills with the
        # other's first, was measured at c4 in round 4: 187.7 / 187.9 k frames/s against 188.6 / 188.9 k - dropped.)
        self._side = None
        # (bf16x3 mode: a split-operand GEMM workgroup - 128 KB of LDS, two 200-register waves per SIMD - cannot share a CU
        # with a workgroup of the persistent BPTT the way an fp32 one can, and takes 642 instead of 373 us per product beside
        # it; the step is nevertheless shorter with them: c2x3 25.2 vs 25.7 ms, c3x3 65.0 vs 68.8 on one box.
        # LC_X3_SIDE_WGRAD=f32 keeps the side-stream products on the fp32 kernels.)
        self.x3_side_f32 = os.environ.get("LC_X3_SIDE_WGRAD", "x3") == "f32"
        self.fuse_dx = os.environ.get("LC_FUSE_DX", "1") != "0"       # one dX product per bidirectional layer (backward; f32 and bf16)
        # bf16: the recurrences store hs / dz ONLY as the bf16 shadows every product of the step reads (no fp32 copy), where
        # every consumer is known to take the natural shadow (`_shadow_only`); LC_C5_SHADOW_ONLY=0 writes both
        self.shadow_only = os.environ.get("LC_C5_SHADOW_ONLY", "1") != "0"
        # development / bisection knob (tools/x3_truth.py): which recurrences of the bf16x3 mode run the split-operand kernels -
        # "both" (default), "fwd", "bwd" or "none" (round 3's mode: split-operand products around fp32 recurrences)
        rec = os.environ.get("LC_X3_REC", "both")
        self.x3_rec_fwd = self.x3 and rec in ("both", "fwd")
        self.x3_rec_bwd = self.x3 and rec in ("both", "bwd")
        # DropoutWrapper masks (and the bf16 shadows of what they produce) ride in the epilogue of the product that
        # writes the masked matrix (lc_gemm_next_epilogue); LC_FUSE_DROPOUT=0 -> separate lc_dropout_scale passes
        self.fuse_dropout = os.environ.get("LC_FUSE_DROPOUT", "1") != "0"

    # ---- products with an activation operand: follow compute_dtype (weight-only products stay ops.gemm / fp32)
    def _shadow(self, t, tr):
        """bf16 shadow of the fp32 matrix / view ``t`` (transposed copy if ``tr``), made once per step.  The cache
        entry keeps ``t`` alive, so its address cannot be handed to another tensor while the shadow is valid; every
        operand is cast only after its last in-place modification of the step."""
        key = (t.data_ptr(), tuple(t.shape), t.stride(0), bool(tr))
        hit = self._shadows.get(key)
        if hit is None:
            nat, trn = ops.cast_bf16(t, nat=not tr, tr=tr)
            hit = (t, trn if tr else nat)
            self._shadows[key] = hit
        return hit[1]

    def _shadow3(self, t, tr=False):
        """x3 shadow (hi | mid | lo bf16 terms, lc_split_bf16x3) of the fp32 matrix / view ``t`` - of its transpose if ``tr``
        (weights only) -, made once per step under the same rules as ``_shadow``."""
        key = (t.data_ptr(), tuple(t.shape), t.stride(0), bool(tr), "x3")
        hit = self._shadows.get(key)
        if hit is None:
            hit = (t, ops.split_bf16x3(ops.transpose(t) if tr else t))
            self._shadows[key] = hit
        return hit[1]

    def _shadow_pad256(self, t):
        """Natural bf16 shadow of the NARROW fp32 matrix ``t`` [rows, C < 256] in a zero-padded [rows, 256] buffer: a K-major
        operand of the 256 x 256 TN kernel (the 40-wide input's and the 44-wide head's weight gradients), once per step."""
        key = (t.data_ptr(), tuple(t.shape), t.stride(0), False, "pad256")
        hit = self._shadows.get(key)
        if hit is None:
            buf = torch.zeros((t.shape[0], 256), dtype=torch.bfloat16, device=t.device)
            ops.cast_bf16(t, out_nat=buf)
            hit = (t, buf)
            self._shadows[key] = hit
        return hit[1]

    def _shadow_only(self, rows, I):
        """Whether the fp32 hs / dz of a layer with input width I may stay unwritten: bf16 mode with shadows, a plain BiLSTM
        layer, and shapes for which EVERY product that reads hs or dz takes its natural bf16 shadow (projection, dproj, dR, dKx,
        dX - see _mm and backward): layer widths in whole 256-tiles, and for a narrow input (layer 0) the padded K-major route,
        which needs rows >= 4096.  The library honours the request in its full-width kernel (N = 1024) and ignores it elsewhere."""
        ps = self.ps
        return bool(self.shadow_only and self.bf16 and self.use_shadows and ps.blstm and not ps.use_bn and ps.P
                    and ps.N % 256 == 0 and ps.Pout % 256 == 0 and rows % 256 == 0 and rows >= 4096
                    and (I % 256 == 0 or I < 256))

    def _adopt_shadow(self, t, shadow):
        """Registers ``shadow`` (bf16, same orientation, written by the kernel that produced ``t``) as the step's
        shadow of the fp32 matrix ``t``: the next ``_shadow(t, tr=False)`` takes it instead of casting."""
        self._shadows[(t.data_ptr(), tuple(t.shape), t.stride(0), False)] = (t, shadow)

    def _mm(self, A, B, ta=False, tb=False, out=None, alpha=1.0, beta=0.0, bias=None, epilogue=None, x3_ok=True):
        """op(A) @ op(B) (+bias), fp32 or - compute_dtype = bf16 - with bf16 operands: through bf16 shadow copies
        in NT form (lc_cast_bf16 + lc_gemm_bf16_nt) when K allows 16-byte operand rows, else with the converting
        loader (lc_gemm_bf16); both round the same operands the same way."""
        if self.x3 and not ta and A.dim() == 2 and B.dim() == 2 and _x3_pays(A.shape[0], B.shape[0] if tb else B.shape[1], A.shape[1]):
            # activation rows x weight: A's x3 shadow as it lies, the weight's with k contiguous (B itself for op(B) = B^T)
            return ops.gemm_bf16x3_nt(self._shadow3(A), self._shadow3(B, tr=not tb), A.shape[1], out=out, alpha=alpha,
                                      beta=beta, bias=bias, epilogue=epilogue)
        if (self.x3 and x3_ok and ta and not tb and A.dim() == 2 and B.dim() == 2 and epilogue is None
                and _x3_pays(A.shape[1], B.shape[1], A.shape[0], split_k=True)):
            # X^T dZ: both activations K-major - the shadows the forward / dX products already made
            return ops.gemm_bf16x3_tn(self._shadow3(A), self._shadow3(B), A.shape[1], B.shape[1], out=out, alpha=alpha,
                                      beta=beta, bias=bias)
        if not self.bf16:
            return ops.gemm(A, B, ta=ta, tb=tb, out=out, alpha=alpha, beta=beta, bias=bias, epilogue=epilogue)
        K = A.shape[0] if ta else A.shape[1]
        if (self.use_shadows and ta and not tb and A.dim() == 2 and B.dim() == 2 and A.shape[1] % 256 == 0
                and B.shape[1] % 256 == 0 and A.stride(0) % 8 == 0 and B.stride(0) % 8 == 0):
            # X^T dZ with both layer widths in whole 256-tiles: the K-major kernel on the NATURAL shadows (transposing
            # LDS reads) - no transposed copy of either activation is ever made
            return ops.gemm_bf16_tn(self._shadow(A, tr=False), self._shadow(B, tr=False), out=out, alpha=alpha,
                                    beta=beta, bias=bias, epilogue=epilogue)
        if (self.use_shadows and ta and not tb and A.dim() == 2 and B.dim() == 2 and beta == 0.0 and bias is None
                and epilogue is None and K >= 4096
                and ((A.shape[1] % 256 == 0 and B.shape[1] < 256) or (B.shape[1] % 256 == 0 and A.shape[1] < 256))):
            # X^T dZ with ONE narrow side (the 40-wide input layer, the 44-wide head): the K-major kernel on the wide
            # operand's natural shadow and a zero-padded 256-column shadow of the narrow one, the valid block copied out -
            # instead of a TRANSPOSED bf16 copy of the wide activation (0.26 - 0.5 ms each at c5's sizes) for the NT form.
            # Same operand roundings, another summation order.
            if A.shape[1] < 256:
                full = ops.gemm_bf16_tn(self._shadow_pad256(A), self._shadow(B, tr=False), alpha=alpha)
                res = full[:A.shape[1]]
            else:
                full = ops.gemm_bf16_tn(self._shadow(A, tr=False), self._shadow_pad256(B), alpha=alpha)
                res = full[:, :B.shape[1]]
            if out is None:
                return res.contiguous()
            out.copy_(res)
            return out
        if (self.use_shadows and not ta and not tb and A.dim() == 2 and B.dim() == 2 and A.shape[0] % 256 == 0
                and B.shape[1] % 256 == 0 and K % 64 == 0 and A.stride(0) % 8 == 0 and B.stride(0) % 8 == 0):
            # X . W with whole 256-tiles: activation AND weight in their natural layouts (no transposed weight copy)
            return ops.gemm_bf16_nn(self._shadow(A, tr=False), self._shadow(B, tr=False), out=out, alpha=alpha,
                                    beta=beta, bias=bias, epilogue=epilogue)
        if self.use_shadows and K % 8 == 0 and A.dim() == 2 and B.dim() == 2:
            return ops.gemm_bf16_nt(self._shadow(A, tr=ta), self._shadow(B, tr=not tb), out=out, alpha=alpha,
                                    beta=beta, bias=bias, K=K, epilogue=epilogue)
        return ops.gemm(A, B, ta=ta, tb=tb, out=out, alpha=alpha, beta=beta, bias=bias, bf16=True, epilogue=epilogue)

    # ------------------------------------------------------------------------------------ helpers
    def _cell(self, prefix):
        ps = self.ps
        k = ps.p(prefix + "/kernel")
        I = k.shape[0] - ps.Pout
        return dict(Kx=k[:I], Kh=k[I:], bias=ps.p(prefix + "/bias"),
                    w_f=ps.p(prefix + "/w_f_diag") if ps.peep else None,
                    w_i=ps.p(prefix + "/w_i_diag") if ps.peep else None,
                    w_o=ps.p(prefix + "/w_o_diag") if ps.peep else None,
                    proj=ps.p(prefix + "/projection/kernel") if ps.P else None, I=I, prefix=prefix)

    def _prefixes(self, i):
        if self.ps.blstm:
            return ["fd%d/frnn%d" % (i, i), "bd%d/brnn%d" % (i, i)]
        if self.ps.cudnn:
            return [CUDNN_CELL % i]
        return ["drnn%d/lstm_cell" % i]

    # ------------------------------------------------------------------------------------ forward
    def forward(self, x, seq_len, drop_seed=0):
        """x [T,B,D] f32 cuda, seq_len [B] int32 cuda -> logits [T,B,V] (time-major)."""
        ps = self.ps
        self._shadows.clear()                       # parameters moved since the last step; activations are new
        T, B, D = x.shape
        assert D == ps.D, (D, ps.D)
        rows, N, P = T * B, ps.N, ps.Pout
        dev = x.device
        inp = x.reshape(rows, D)
        bn_saved = {}

        def batch_norm(name, t):
            y, mean, var = ops.bn_forward(t, ps.p(name + "/gamma"), ps.p(name + "/beta"), self.is_training,
                                          ps.aux[name + "/moving_mean"], ps.aux[name + "/moving_variance"])
            bn_saved[name] = dict(x=t, mean=mean, var=var)
            return y

        if ps.use_bn:
            inp = batch_norm("drnn_bn_0_0", inp)                                      # lstm.py:271-277
        layers = []
        for i in range(ps.num_layers):
            cells = [self._cell(p) for p in self._prefixes(i)]
            ndir = len(cells)
            Y = torch.empty((rows, ndir * P), dtype=torch.float32, device=dev)
            dirs = []
            # (the two directions' zx products on two streams - so that one's first round of tiles fills the other's partial
            # last one - measured in round 6 at c4: 338.7 / 338.7 against 340.9 / 338.7 ms; not taken, like round 4's attempt
            # with the weight gradients)
            for d, c in enumerate(cells):
                zx = self._mm(inp, c["Kx"], bias=c["bias"])                          # hoisted x_t.Kx + b
                R = ops.gemm(c["proj"], c["Kh"]) if c["proj"] is not None else c["Kh"]
                cs = torch.empty((rows, N), dtype=torch.float32, device=dev)
                hs = torch.empty((rows, N), dtype=torch.float32, device=dev)
                dirs.append(dict(zx=zx, R=R, w_f=c["w_f"], w_i=c["w_i"], w_o=c["w_o"], cs=cs, hs=hs,
                                 reverse=(d == 1)))
                if self.bf16 and self.use_shadows and c["proj"] is not None and N % 8 == 0:
                    # the projection reads hs as a bf16 shadow: let the recurrence write it in the same pass
                    dirs[-1]["hs_bf16"] = torch.empty((rows, N), dtype=torch.bfloat16, device=dev)
                    dirs[-1]["shadow_only"] = self._shadow_only(rows, c["I"])
            # (split-operand mode, widths up to 320: the fp32 forward recurrence is the faster one since its operands are
            # requested a step ahead - 1.77 against 2.11 us per step at N = 320, 1.44 / 1.72 at 256; from 512 up the
            # split-operand kernel leads, 2.56 against 2.86: profiles/r5_persist_probe_ahead.txt)
            ops.lstm_fwd(dirs, seq_len, T, B, N, self.forget_bias, bf16=self.bf16,
                         x3=self.x3_rec_fwd and x3_forward_recurrence(N))
            for dd in dirs:
                if dd.get("hs_bf16") is not None:
                    self._adopt_shadow(dd["hs"], dd["hs_bf16"])
            residual = (i == 0 and D == 2 * P) if ps.blstm else False               # bilstm.py:199
            drop = ps.blstm and self.keep < 1.0                                      # DropoutWrapper on each direction
            # bf16 path: the pass that applies the mask also writes the bf16 shadow the next product reads
            # (round 6: without dropout too - inference, parity runs - when every direction has a projection product to carry it)
            Y16 = (torch.empty((rows, ndir * P), dtype=torch.bfloat16, device=dev)
                   if (drop or (ps.blstm and all(c_["proj"] is not None for c_ in cells)))
                   and self.bf16 and self.use_shadows and not residual and not ps.use_bn and P % 4 == 0 else None)
            for d, c in enumerate(cells):
                half = Y[:, d * P:(d + 1) * P]
                if c["proj"] is not None:
                    ep = None
                    if drop and self.fuse_dropout:          # ... and that pass is the projection's own epilogue
                        # (shadow_only: the fp32 layer output stays unwritten where every reader - the next layer's zx product,
                        # its dKx, the head and its weight gradient - takes the bf16 shadow: _shadow_only)
                        ep = ops.Epilogue(self.keep, drop_seed, 2 * i + d, P,
                                          None if Y16 is None else Y16[:, d * P:(d + 1) * P],
                                          shadow_only=Y16 is not None and ps.E == 0 and self._shadow_only(rows, ndir * P))
                    elif not drop and Y16 is not None:      # no mask: the shadow alone (keep = 1 leaves the values as they are)
                        ep = ops.Epilogue(1.0, 0, 0, 1, Y16[:, d * P:(d + 1) * P],
                                          shadow_only=ps.E == 0 and self._shadow_only(rows, ndir * P))
                    self._mm(dirs[d]["hs"], c["proj"], out=half, epilogue=ep)        # m_t = m'_t . proj, batched
                elif drop and self.fuse_dropout:                                     # the strided copy carries the mask
                    ops.dropout_scale(dirs[d]["hs"], self.keep, drop_seed, 2 * i + d, out=half,
                                      shadow=None if Y16 is None else Y16[:, d * P:(d + 1) * P])
                else:
                    ops.dropout_scale(dirs[d]["hs"], 1.0, 0, 0, out=half)            # plain strided copy
            if ps.blstm:
                if drop:
                    for d, c in enumerate(cells):
                        if not self.fuse_dropout:
                            ops.dropout_scale(Y[:, d * P:(d + 1) * P], self.keep, drop_seed, 2 * i + d,
                                              shadow=None if Y16 is None else Y16[:, d * P:(d + 1) * P])
                if Y16 is not None:
                    self._adopt_shadow(Y, Y16)
                if residual:
                    ops.dropout_scale(inp, 1.0, 0, 0, out=Y, accumulate=True)        # finput + concat
            elif ps.cudnn:
                residual = False                                                     # bare cells (lstm.py:73-76)
            else:
                residual = not (i == 0 and D != P)                                   # lstm.py:236-260
                if residual:
                    ops.dropout_scale(inp, 1.0, 0, 0, out=Y, accumulate=True)        # ResidualWrapper
                    if ps.use_bn:         # a batch-normalised input is non-zero in the padded frames, but
                        ops.length_mask_(Y, seq_len, T, B)    # dynamic_rnn zeroes the wrapped cell's output there
                if self.keep < 1.0:
                    ops.dropout_scale(Y, self.keep, drop_seed, 2 * i)
            layers.append(dict(inp=inp, dirs=dirs, cells=cells, Y=Y, residual=residual))
            inp = batch_norm("drnn_bn%d" % i, Y) if ps.use_bn else Y                  # lstm.py:288-294
        head = {}
        if ps.E > 0:
            a = self._mm(inp, ps.p("Variable"), bias=ps.p("Variable_1"))
            q = self._mm(inp, ps.p("Variable_2"), bias=ps.p("Variable_3"))
            logits, pi = ops.moe_combine_fwd(a, q, ps.E, ps.V, self.tau, self.keep, drop_seed)
            head = dict(q=q, pi=pi)
        else:
            logits = self._mm(inp, ps.p("Variable"), bias=ps.p("Variable_1"))
        self.saved = dict(layers=layers, head=head, T=T, B=B, seq_len=seq_len, drop_seed=drop_seed, bn=bn_saved,
                          top=inp)
        return logits.view(T, B, ps.V)

    def encoder(self):
        """Final (c, m) states of the last layer, concatenated as bilstm.py:206-208.  Computed lazily
        from the saved activations (graph.py never uses it)."""
        sv = self.saved
        B = sv["B"]
        last = sv["layers"][-1]
        sl = sv["seq_len"].long()
        outs = []
        for d, dd in enumerate(last["dirs"]):
            t_last = torch.zeros_like(sl) if dd["reverse"] else (sl - 1).clamp(min=0)
            idx = t_last * B + torch.arange(B, device=sl.device)
            # (bf16 mode may have left the fp32 hs unwritten: its shadow holds the values the projection reads anyway)
            h = (dd["hs_bf16"][idx].float() if dd.get("shadow_only") else dd["hs"][idx]).contiguous()
            proj = last["cells"][d]["proj"]
            outs += [dd["cs"][idx], self._mm(h, proj) if proj is not None else h]
        return torch.cat(outs, dim=1)

    # ------------------------------------------------------------------------------------ backward
    def layer_grad_range(self, i):
        """[lo, hi) of layer i's L2-decayed tensors (kernels, peepholes, projections of its cells) in the flat buffers."""
        ps = self.ps
        lo = ps.offsets[self._prefixes(i)[0] + "/kernel"]
        if i + 1 < ps.num_layers:
            hi = ps.offsets[self._prefixes(i + 1)[0] + "/kernel"]
        else:
            names = [s_[0] for s_ in ps.specs]
            k = max(j for j, nm in enumerate(names) if nm.startswith(self._prefixes(i)[-1] + "/") and "bias" not in nm)
            hi = ps.offsets[names[k + 1]] if k + 1 < len(names) else ps.n
        return lo, hi

    def backward(self, dlogits, buckets=None):
        """dlogits [T,B,V] -> fills ps.grad (overwrites; call once per forward).  ``buckets`` (dp.GradientBuckets, data
        parallelism): layer i + 1's gradient range is handed over once the BPTT of layer i is enqueued, and every
        outstanding bucket is waited for in front of the next recurrence."""
        ps, sv = self.ps, self.saved
        T, B = sv["T"], sv["B"]
        rows, N, P = T * B, ps.N, ps.Pout
        seed = sv["drop_seed"]
        dl = dlogits.reshape(rows, ps.V)
        ps.grad.zero_()
        top = sv["top"]
        keepalive = []

        def batch_norm_bwd(name, d):
            b = sv["bn"][name]
            return ops.bn_backward(b["x"], d, b["mean"], b["var"], ps.p(name + "/gamma"), self.is_training,
                                   ps.g(name + "/gamma"), ps.g(name + "/beta"), dx=d)


        def masked_dY(i, width):
            """The epilogue that finishes layer i's dY inside the product writing it (the DropoutWrapper's backward: the
            forward mask again, per direction), plus the bf16 shadow both products of each half (dh, dproj) read -
            or (None, None) when layer i masks in a pass of its own."""
            if not (self.fuse_dropout and ps.blstm and self.keep < 1.0 and not ps.use_bn):
                return None, None
            d16 = (torch.empty((rows, width), dtype=torch.bfloat16, device=dl.device)
                   if self.bf16 and self.use_shadows and P % 4 == 0 else None)
            # (shadow_only: dY is read through the shadow by both products of each half - dh = half . proj^T, dproj = hs^T half)
            return ops.Epilogue(self.keep, seed, 2 * i, P, d16,
                                shadow_only=d16 is not None and ps.E == 0 and self._shadow_only(rows, width)), d16

        ep, dY16 = masked_dY(ps.num_layers - 1, top.shape[1])
        premasked = ep is not None
        if ps.E > 0:
            q, pi = sv["head"]["q"], sv["head"]["pi"]
            da = ops.moe_combine_bwd(pi, q, dl, ps.E, ps.V, self.tau, self.keep, seed)       # q now holds dq
            dY = self._mm(da, ps.p("Variable"), tb=True)
            self._mm(q, ps.p("Variable_2"), tb=True, out=dY, beta=1.0, epilogue=ep)
            self._mm(top, da, ta=True, out=ps.g("Variable"))
            ops.colsum(da, out=ps.g("Variable_1"))
            self._mm(top, q, ta=True, out=ps.g("Variable_2"))
            ops.colsum(q, out=ps.g("Variable_3"))
        else:
            dY = self._mm(dl, ps.p("Variable"), tb=True, epilogue=ep)
            self._mm(top, dl, ta=True, out=ps.g("Variable"))
            ops.colsum(dl, out=ps.g("Variable_1"))
        for i in reversed(range(ps.num_layers)):
            L = sv["layers"][i]
            cells, dirs, inp = L["cells"], L["dirs"], L["inp"]
            ndir = len(cells)
            need_dinp = i > 0 or ps.use_bn
            dres = None
            if ps.use_bn:
                dY = batch_norm_bwd("drnn_bn%d" % i, dY)
            if ps.blstm:
                if premasked:                    # the product that wrote dY masked it and wrote its shadow
                    if dY16 is not None:
                        for d in range(ndir):
                            self._adopt_shadow(dY[:, d * P:(d + 1) * P], dY16[:, d * P:(d + 1) * P])
                elif self.keep < 1.0:
                    dY16 = (torch.empty((rows, ndir * P), dtype=torch.bfloat16, device=dY.device)
                            if self.bf16 and self.use_shadows and P % 4 == 0 and dY.stride(0) % 4 == 0 else None)
                    for d in range(ndir):
                        half = dY[:, d * P:(d + 1) * P]
                        if dY16 is None:
                            ops.dropout_scale(half, self.keep, seed, 2 * i + d)
                        else:            # mask + the bf16 shadow both products of `half` (dh, dproj) read, one pass
                            ops.dropout_scale(half, self.keep, seed, 2 * i + d, shadow=dY16[:, d * P:(d + 1) * P])
                            self._adopt_shadow(half, dY16[:, d * P:(d + 1) * P])
            else:
                if self.keep < 1.0:
                    ops.dropout_scale(dY, self.keep, seed, 2 * i)
                if L["residual"]:
                    if ps.use_bn:
                        ops.length_mask_(dY, sv["seq_len"], T, B)
                    dres = dY                                                            # d(out+inp)/d inp
            bdirs = []
            for d, c in enumerate(cells):
                half = dY[:, d * P:(d + 1) * P]
                if c["proj"] is not None:
                    dh = self._mm(half, c["proj"], tb=True)                              # [rows,N]
                    RT = ops.transpose(dirs[d]["R"])             # (proj.Kh)^T: the forward's fold, transposed (not redone)
                else:
                    dh = half.contiguous() if ndir > 1 else half
                    RT = ops.transpose(c["Kh"])
                # w_f_diag, w_i_diag, w_o_diag are adjacent in the flat gradient buffer: one [3,N] block
                dpeep = None
                if ps.peep:
                    o = ps.offsets[c["prefix"] + "/w_f_diag"]
                    dpeep = ps.grad[o:o + 3 * N]
                bdirs.append(dict(gates=dirs[d]["zx"], RT=RT, w_f=c["w_f"], w_i=c["w_i"], w_o=c["w_o"],
                                  cs=dirs[d]["cs"], dh=dh, dpeep=dpeep, dbias=ps.g(c["prefix"] + "/bias"),
                                  reverse=dirs[d]["reverse"]))
                if self.bf16 and self.use_shadows and (i > 0 or ps.use_bn or N % 256 == 0):
                    # dX = dz . Kx^T (and, in whole 256-tiles, dKx / dR on the K-major kernel) read dz as a bf16 shadow:
                    # written by the BPTT itself
                    bdirs[-1]["dz_bf16"] = torch.empty((rows, 4 * N), dtype=torch.bfloat16, device=dY.device)
                    bdirs[-1]["shadow_only"] = self._shadow_only(rows, c["I"])
                side_x3 = not (self.overlap_wgrad and i > 0 and self.x3_side_f32)      # this layer's weight gradients on x3?
                if self.x3_rec_bwd and N % 4 == 0 and (_x3_pays(rows, c["I"], 4 * N)
                                                or (side_x3 and _x3_pays(c["I"], 4 * N, rows, split_k=True))
                                                or (side_x3 and T > 1 and _x3_pays(N, 4 * N, rows - B, split_k=True))):
                    # the dX / dKx / dR products read dz as an x3 shadow: the split-operand BPTT's producers write it (they
                    # split dz for the exchange anyway); other schedules get the split pass behind the recurrence
                    bdirs[-1]["dz_x3"] = torch.empty((rows, 12 * N), dtype=torch.bfloat16, device=dY.device)
            if buckets is not None:
                buckets.wait()                   # a persistent recurrence needs every CU: no collective kernel beside it
            ops.lstm_bwd(bdirs, sv["seq_len"], T, B, N, bf16=self.bf16, x3=self.x3_rec_bwd)
            if buckets is not None and i + 1 < ps.num_layers and not self.overlap_wgrad:
                buckets.issue(*self.layer_grad_range(i + 1))      # runs beside this layer's weight-gradient GEMMs
            for bd in bdirs:
                if bd.get("dz_bf16") is not None:
                    self._adopt_shadow(bd["gates"], bd["dz_bf16"])
                if bd.get("dz_x3") is not None:
                    t_ = bd["gates"]
                    self._shadows[(t_.data_ptr(), tuple(t_.shape), t_.stride(0), False, "x3")] = (t_, bd["dz_x3"])
            dinp = torch.empty((rows, inp.shape[1]), dtype=torch.float32, device=dY.device) if need_dinp else None
            ep, next16 = masked_dY(i - 1, inp.shape[1]) if i > 0 else (None, None)    # rides on the LAST product into dinp
            overlap = self.overlap_wgrad and i > 0
            side_x3 = not (overlap and self.x3_side_f32)
            # bf16 shadows, both directions: dinp = dz_f . Kx_f^T + dz_b . Kx_b^T as ONE product whose reduction walks both
            # operand pairs (lc_gemm_bf16_nt2) - dinp is written once instead of written, read back and written again
            fuse_dx = (self.bf16 and self.use_shadows and ndir == 2 and need_dinp and not overlap and N % 2 == 0
                       and self.fuse_dx)
            # ... and the same in float32 (lc_gemm_f32_nt2; not in bf16x3 mode, whose products are split-operand ones)
            fuse_dx32 = (not self.bf16 and not self.x3 and ndir == 2 and need_dinp and not overlap and self.fuse_dx
                         and bdirs[0]["gates"].stride(0) == bdirs[1]["gates"].stride(0)
                         and cells[0]["Kx"].stride(0) == cells[1]["Kx"].stride(0))
            if overlap and need_dinp:            # the next layer's BPTT waits for this only: issue it first
                # (the two-segment product here too was measured in round 6: c3 80.8 / 81.5 -> 80.7 / 81.5 ms, c2 inside its
                # run-to-run spread - not taken)
                for d, c in enumerate(cells):
                    self._mm(bdirs[d]["gates"], c["Kx"], tb=True, out=dinp, beta=(0.0 if d == 0 else 1.0),
                             epilogue=ep if d == ndir - 1 else None)
            main = torch.cuda.current_stream()
            if overlap:
                if self._side is None:
                    self._side = torch.cuda.Stream(device=dY.device, priority=0)
                ev = torch.cuda.Event()
                ev.record(main)
                self._side.wait_event(ev)
                keepalive.append((dY, bdirs))    # read on the side stream after main has dropped its references
            with torch.cuda.stream(self._side if overlap else main):
                for d, c in enumerate(cells):
                    pre = c["prefix"]
                    dz, hs = bdirs[d]["gates"], dirs[d]["hs"]
                    gk = ps.g(pre + "/kernel")
                    I = c["I"]
                    self._mm(inp, dz, ta=True, out=gk[:I], x3_ok=side_x3)                    # dKx = X^T dZ
                    if T > 1:                                                                # dR = M'_{prev}^T dZ
                        if dirs[d]["reverse"]:
                            hprev, dzs = hs[B:], dz[:rows - B]
                        else:
                            hprev, dzs = hs[:rows - B], dz[B:]
                        dR_out = None if c["proj"] is not None else gk[I:]
                        if self.x3 and side_x3 and _x3_pays(N, 4 * N, rows - B, split_k=True):
                            # row windows, one step apart, of the x3 shadows of the WHOLE hs / dz (shared with the
                            # projection, dKx, dproj and dX)
                            hs_3, dz_3 = self._shadow3(hs), self._shadow3(dz)
                            if dirs[d]["reverse"]:
                                a_v, b_v = hs_3[B:rows], dz_3[:rows - B]
                            else:
                                a_v, b_v = hs_3[:rows - B], dz_3[B:rows]
                            dR = ops.gemm_bf16x3_tn(a_v, b_v, N, 4 * N, out=dR_out)
                        elif self.use_shadows and N % 256 == 0:
                            # row windows, one step apart, of the natural shadows of the WHOLE hs / dz (shared with dKx,
                            # dproj, dX and the projection): K-major kernel, nothing transposed
                            hs_n, dz_n = self._shadow(hs, tr=False), self._shadow(dz, tr=False)
                            if dirs[d]["reverse"]:
                                a_v, b_v = hs_n[B:rows], dz_n[:rows - B]
                            else:
                                a_v, b_v = hs_n[:rows - B], dz_n[B:rows]
                            dR = ops.gemm_bf16_tn(a_v, b_v, out=dR_out)
                        elif self.use_shadows and B % 8 == 0:
                            # column windows of the transposed shadows of the WHOLE hs / dz (shared with dKx, dproj)
                            hs_t, dz_t = self._shadow(hs, tr=True), self._shadow(dz, tr=True)
                            if dirs[d]["reverse"]:
                                a_v, b_v = hs_t[:, B:rows], dz_t[:, :rows - B]
                            else:
                                a_v, b_v = hs_t[:, :rows - B], dz_t[:, B:rows]
                            dR = ops.gemm_bf16_nt(a_v, b_v, out=dR_out, K=rows - B)
                        else:
                            dR = self._mm(hprev, dzs, ta=True, out=dR_out, x3_ok=side_x3)
                        if c["proj"] is not None:
                            ops.gemm(c["proj"], dR, ta=True, out=gk[I:])                     # dKh = proj^T dR
                    half = dY[:, d * P:(d + 1) * P]
                    if c["proj"] is not None:
                        gp = ps.g(pre + "/projection/kernel")
                        self._mm(hs, half, ta=True, out=gp, x3_ok=side_x3)                   # from m_t = m'_t.proj
                        if T > 1:
                            ops.gemm(dR, c["Kh"], tb=True, out=gp, beta=1.0)                 # from R = proj.Kh
                    if need_dinp and not overlap and not fuse_dx and not fuse_dx32:
                        self._mm(dz, c["Kx"], tb=True, out=dinp, beta=(0.0 if d == 0 else 1.0),
                                 epilogue=ep if d == ndir - 1 else None)
                if fuse_dx32:
                    ops.gemm_nt2(bdirs[0]["gates"], cells[0]["Kx"], bdirs[1]["gates"], cells[1]["Kx"], out=dinp, epilogue=ep)
                if fuse_dx:
                    ops.gemm_bf16_nt2(self._shadow(bdirs[0]["gates"], tr=False), self._shadow(cells[0]["Kx"], tr=False),
                                      self._shadow(bdirs[1]["gates"], tr=False), self._shadow(cells[1]["Kx"], tr=False),
                                      out=dinp, epilogue=ep)
            if need_dinp:
                if dres is not None:
                    ops.dropout_scale(dres, 1.0, 0, 0, out=dinp, accumulate=True)
                dY, dY16, premasked = dinp, next16, ep is not None
        if ps.use_bn:
            batch_norm_bwd("drnn_bn_0_0", dY)
        if keepalive:
            torch.cuda.current_stream().wait_stream(self._side)     # gradients complete before anyone reads them
            keepalive.clear()
        self.saved = None
        self._shadows.clear()

    def update_moving_averages(self, bn_saved=None):
        """The batch-norm UPDATE_OPS the train op depends on (graph.py:194-196); call once per training step with the
        batch moments of that step's forward (default: the ones still saved, i.e. before backward())."""
        if bn_saved is None:
            bn_saved = (self.saved or {}).get("bn", {})
        for name, b in bn_saved.items():
            if self.is_training:
                ops.bn_update_moving(self.ps.aux[name + "/moving_mean"], self.ps.aux[name + "/moving_variance"],
                                     b["mean"], b["var"])
