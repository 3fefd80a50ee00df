"""Tensor-level wrappers over the C ABI (include/lstm_ctc_hip.h).

PyTorch is used here for device memory and streams only: every function takes CUDA(ROCm) float32 /
int32 tensors, passes raw ``data_ptr()``s plus the current HIP stream to ``liblstm_ctc_hip.so``, and
returns tensors.  No arithmetic happens in torch on the product path, and nothing falls back to the
CPU: a missing library or a CPU tensor raises.
"""
import ctypes
import threading

import numpy as np
import torch

from . import _lib

_workspaces = {}

# Optional live profiling (bench.py): when PROFILE is a list, every GEMM / CTC call is bracketed by HIP
# events on the launch stream and (kind, work, start_event, end_event) is appended; work = flops or bytes.
PROFILE = None


def _prof_begin():
    if PROFILE is None:
        return None
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    return e


def _prof_end(kind, work, start):
    if start is not None:
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        PROFILE.append((kind, work, start, e))


def _require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise _lib.LibraryError("lstm_ctc_amd ops need GPU tensors (got a CPU tensor); there is no CPU path")


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def workspace(name, nbytes, device):
    """A cached, grow-only scratch buffer (the library never allocates)."""
    device = torch.device(device)
    if device.type == "cuda" and device.index is None:                    # "cuda" and "cuda:0" are the same buffer
        device = torch.device("cuda", torch.cuda.current_device())
    key = (name, str(device), torch.cuda.current_stream(device).cuda_stream)      # one scratch buffer per stream
    buf = _workspaces.get(key)
    if buf is None or buf.numel() < nbytes:
        new = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
        if name == "lstm":
            # the first 256 bytes are the control block with the sticky status word (lstm_ctc_hip.h): it survives growth
            if buf is None:
                new[:256].zero_()
            else:
                new[:256].copy_(buf[:256])
        buf = new
        _workspaces[key] = buf
    return buf


def lstm_status(device):
    """int32 view [1] of the sticky status word of this stream's LSTM workspace (LC_LSTM_STATUS_OFFSET): non-zero after
    a persistent-recurrence launch could not complete (its outputs are NaN).  Zero it at the start of a step, read it
    at the step's sync point; get the view AFTER the step's last lstm call (the workspace may have grown)."""
    buf = workspace("lstm", 256, device)
    o = _lib.LSTM_STATUS_OFFSET
    return buf[o:o + 4].view(torch.int32)


def debug_spin(blocks, microseconds, lds_bytes=96 * 1024):
    """Development hook (lc_debug_spin): `blocks` idle workgroups, each holding `lds_bytes` of LDS, resident for `microseconds`
    on the CURRENT stream - a foreign kernel for the co-residency tests (96 KB: no persistent recurrence fits beside one)."""
    _lib.check(_lib.load().lc_debug_spin(int(blocks), int(microseconds), int(lds_bytes), _stream()), "lc_debug_spin")


def set_option(name, value=None):
    """Per-thread override of a library development switch (lc_set_option; None clears it)."""
    _lib.check(_lib.load().lc_set_option(name.encode(), _lib.OPTION_UNSET if value is None else int(value)),
               "lc_set_option")


def get_option(name):
    """Effective value of a switch (override, else its LC_* environment variable), or None for the library default."""
    v = ctypes.c_long(0)
    _lib.check(_lib.load().lc_get_option(name.encode(), ctypes.byref(v)), "lc_get_option")
    return None if v.value == _lib.OPTION_UNSET else int(v.value)


_tls = threading.local()


class force_launch_train:
    """Context: every lstm_fwd / lstm_bwd issued by THIS thread inside runs the per-step launch train - the checked
    fallback of the persistent schedules.  A per-thread library override (lc_set_option "lstm_persistent"), not an
    environment mutation: loader threads that read os.environ are never raced."""

    def __enter__(self):
        self._outer = getattr(_tls, "launch_train_depth", 0)
        _tls.launch_train_depth = self._outer + 1
        set_option("lstm_persistent", 0)

    def __exit__(self, *exc):
        _tls.launch_train_depth = self._outer
        if self._outer == 0:
            set_option("lstm_persistent", None)


def last_lstm_schedule():
    """Decoded lc_debug_last_lstm_schedule(): dict(kind, mt, bf16, backward)."""
    v = _lib.load().lc_debug_last_lstm_schedule()
    kinds = {0: "none", 1: "persistent_f32", 2: "persistent_bf16", 3: "two_stream_train", 4: "launch_train",
             5: "persistent_f32_xcd_pair", 6: "persistent_x3_xcd_pair", 7: "persistent_x3"}
    return dict(kind=kinds.get(v & 0xff, "?"), mt=(v >> 8) & 0xff, bf16=bool(v >> 16 & 1), backward=bool(v >> 17 & 1),
                dz_shadow_in_kernel=bool(v >> 18 & 1))


def _f32c(t):
    assert t.dtype == torch.float32, t.dtype
    return t if t.is_contiguous() else t.contiguous()


def _rowmajor2d(t):
    assert t.dim() == 2 and t.dtype == torch.float32
    if t.stride(1) != 1:
        t = t.contiguous()
    return t, (t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1]))


# ------------------------------------------------------------------------------------------ GEMM
class Epilogue:
    """Fused finish of ONE product (lc_gemm_next_epilogue): the dropout mask of stream ``stream0 + col // width`` applied
    to the output, and / or its bf16 shadow ``shadow`` ([M,N] bf16 view, last stride 1) written in the same pass."""

    def __init__(self, keep=1.0, seed=0, stream0=0, width=1, shadow=None, shadow_only=False):
        self.keep, self.seed, self.stream0, self.width, self.shadow = float(keep), int(seed), int(stream0), int(width), shadow
        self.shadow_only = bool(shadow_only) and shadow is not None      # the fp32 output is left untouched (nobody reads it)


def _arm(lib, epilogue, out):
    """Arms the calling thread's next lc_gemm_* call; call it RIGHT before that call (nothing between may fail)."""
    if epilogue is None:
        return
    sh = epilogue.shadow
    if sh is not None:
        _require_cuda(sh)
        assert sh.dtype == torch.bfloat16 and sh.shape == out.shape and sh.stride(1) == 1
    e = _lib.GemmEpilogue(epilogue.keep, epilogue.seed & 0xFFFFFFFF, epilogue.stream0, epilogue.width,
                          _ptr(sh), (sh.stride(0) if sh.shape[0] > 1 else max(sh.stride(0), sh.shape[1])) if sh is not None else 0,
                          int(getattr(epilogue, "shadow_only", False)))
    _lib.check(lib.lc_gemm_next_epilogue(ctypes.byref(e)), "lc_gemm_next_epilogue")


def gemm(A, B, ta=False, tb=False, out=None, alpha=1.0, beta=0.0, bias=None, bf16=False, epilogue=None):
    """out[M,N] = alpha * op(A) @ op(B) + beta*out (+ bias).  A/B/out are 2-D row-major views whose
    last stride is 1 (row stride free, so column slices of wider buffers are fine).
    bf16=True: operands rounded to bf16 on load, fp32 accumulate (lc_gemm_bf16; config c5)."""
    lib = _lib.load()
    _require_cuda(A, B, out, bias)
    A, lda = _rowmajor2d(A)
    B, ldb = _rowmajor2d(B)
    M, K = (A.shape[1], A.shape[0]) if ta else (A.shape[0], A.shape[1])
    K2, N = (B.shape[1], B.shape[0]) if tb else (B.shape[0], B.shape[1])
    assert K == K2, (A.shape, B.shape, ta, tb)
    if out is None:
        assert beta == 0.0
        out = torch.empty((M, N), dtype=torch.float32, device=A.device)
    assert out.shape == (M, N) and out.stride(1) == 1 and out.dtype == torch.float32
    ldc = out.stride(0) if M > 1 else max(out.stride(0), N)
    nbytes = lib.lc_gemm_workspace_bytes(M, N, K)
    ws = workspace("gemm", nbytes, A.device) if nbytes else None
    ev = _prof_begin()
    fn, who = (lib.lc_gemm_bf16, "lc_gemm_bf16") if bf16 else (lib.lc_gemm_f32, "lc_gemm_f32")
    _arm(lib, epilogue, out)
    _lib.check(fn(int(ta), int(tb), M, N, K, alpha, _ptr(A), lda, _ptr(B), ldb, beta, _ptr(out), ldc,
                  _ptr(bias), _ptr(ws), nbytes, _stream()), who)
    _prof_end("gemm_bf16" if bf16 else "gemm", 2.0 * M * N * K, ev)
    return out


def transpose(x):
    lib = _lib.load()
    _require_cuda(x)
    x = _f32c(x)
    rows, cols = x.shape
    out = torch.empty((cols, rows), dtype=torch.float32, device=x.device)
    _lib.check(lib.lc_transpose(_ptr(x), rows, cols, _ptr(out), _stream()), "lc_transpose")
    return out


def colsum(x, out=None, accumulate=False):
    lib = _lib.load()
    _require_cuda(x, out)
    x, ldx = _rowmajor2d(x)
    rows, N = x.shape
    if out is None:
        out = torch.empty(N, dtype=torch.float32, device=x.device)
        accumulate = False
    nbytes = lib.lc_colsum_workspace_bytes(N)
    ws = workspace("colsum", nbytes, x.device)
    _lib.check(lib.lc_colsum(_ptr(x), rows, N, ldx, _ptr(out), int(accumulate), _ptr(ws), nbytes, _stream()),
               "lc_colsum")
    return out


def dropout_scale(x, keep, seed, stream_id, out=None, accumulate=False, shadow=None):
    """out (+)= x * Bernoulli(keep)/keep with the counter-based mask; x, out: 2-D [rows,P] views.  shadow: optional bf16
    [rows,P] view that receives the rounded result in the same pass (lc_dropout_scale_bf16)."""
    lib = _lib.load()
    _require_cuda(x, out, shadow)
    assert x.dim() == 2 and x.stride(1) == 1
    if out is None:
        out = x
    rows, P = x.shape
    if shadow is not None:
        assert shadow.dtype == torch.bfloat16 and shadow.shape == x.shape and shadow.stride(1) == 1
        ev = _prof_begin()
        _lib.check(lib.lc_dropout_scale_bf16(_ptr(x), rows, P, x.stride(0), float(keep), int(seed) & 0xFFFFFFFF,
                                             int(stream_id), _ptr(out), out.stride(0), int(accumulate), _ptr(shadow),
                                             shadow.stride(0), _stream()), "lc_dropout_scale_bf16")
        _prof_end("cast_bf16", float(rows) * P * 10, ev)       # bytes moved: 4 read + 4 + 2 written
        return out
    _lib.check(lib.lc_dropout_scale(_ptr(x), rows, P, x.stride(0), float(keep), int(seed) & 0xFFFFFFFF,
                                    int(stream_id), _ptr(out), out.stride(0), int(accumulate), _stream()),
               "lc_dropout_scale")
    return out


# ------------------------------------------------------------------------------------------ CTC
def ctc_loss(logits, labels, offsets, seq_len, max_label_len, want_grad=True):
    """logits [T,B,V] f32; labels flat int32; offsets [B+1] int32; seq_len [B] int32 (all on device).
    Returns (loss[B], grad[T,B,V] or None)."""
    lib = _lib.load()
    _require_cuda(logits, labels, offsets, seq_len)
    logits = _f32c(logits)
    T, B, V = logits.shape
    assert labels.dtype == torch.int32 and offsets.dtype == torch.int32 and seq_len.dtype == torch.int32
    loss = torch.empty(B, dtype=torch.float32, device=logits.device)
    grad = torch.empty_like(logits) if want_grad else None
    nbytes = lib.lc_ctc_workspace_bytes(T, B, V, int(max_label_len))
    ws = workspace("ctc", nbytes, logits.device)
    if labels.numel() == 0:
        labels = torch.zeros(1, dtype=torch.int32, device=logits.device)
    ev = _prof_begin()
    _lib.check(lib.lc_ctc_loss(_ptr(logits), T, B, V, _ptr(labels), _ptr(offsets), _ptr(seq_len),
                               int(max_label_len), _ptr(loss), _ptr(grad), _ptr(ws), nbytes, _stream()),
               "lc_ctc_loss")
    _prof_end("ctc", (T, B, V), ev)
    return loss, grad


def ctc_greedy(logits, seq_len):
    """Returns (tokens [B,T] int32, out_len [B] int32) on device."""
    lib = _lib.load()
    _require_cuda(logits, seq_len)
    logits = _f32c(logits)
    T, B, V = logits.shape
    tokens = torch.empty((B, T), dtype=torch.int32, device=logits.device)
    out_len = torch.empty(B, dtype=torch.int32, device=logits.device)
    ws = workspace("greedy", 4 * T * B, logits.device)
    _lib.check(lib.lc_ctc_greedy(_ptr(logits), T, B, V, _ptr(seq_len), _ptr(tokens), _ptr(out_len), _ptr(ws),
                                 _stream()), "lc_ctc_greedy")
    return tokens, out_len


def edit_distance_host(tokens, token_len, truth_flat, truth_offsets):
    """Host-side Levenshtein per utterance (numpy int32 in, numpy int32 out)."""
    lib = _lib.load()
    tokens = np.ascontiguousarray(tokens, np.int32)
    token_len = np.ascontiguousarray(token_len, np.int32)
    truth_flat = np.ascontiguousarray(truth_flat, np.int32)
    truth_offsets = np.ascontiguousarray(truth_offsets, np.int32)
    B = tokens.shape[0]
    dist = np.zeros(B, np.int32)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    _lib.check(lib.lc_edit_distance_host(p(tokens), tokens.shape[1], p(token_len), p(truth_flat), p(truth_offsets),
                                         B, p(dist)), "lc_edit_distance_host")
    return dist


# ------------------------------------------------------------------------------------------ LSTM
def lstm_fwd(dirs, seq_len, T, B, N, forget_bias, bf16=False, x3=False):
    """dirs: list (1 or 2) of dict(zx, R, w_f, w_i, w_o, cs, hs, reverse).  Runs the recurrence in place.
    bf16=True: the step GEMM's operands (m'_{t-1}, R) are rounded to bf16 (lc_lstm_fwd_bf16; config c5).
    x3=True: the step product as fp32-on-bf16x3 - both operands split exactly into three bf16 terms, six term products in
    fp32 (lc_lstm_fwd_x3; fp32-grade results) - where a split-operand kernel exists for the shape, fp32 kernels elsewhere."""
    assert not (bf16 and x3)
    lib = _lib.load()
    arr = (_lib.LstmFwdDir * len(dirs))()
    for i, d in enumerate(dirs):
        _require_cuda(d["zx"], d["R"], d["cs"], d["hs"])
        arr[i].zx, arr[i].R = d["zx"].data_ptr(), d["R"].data_ptr()
        arr[i].w_f = d["w_f"].data_ptr() if d.get("w_f") is not None else None
        arr[i].w_i = d["w_i"].data_ptr() if d.get("w_i") is not None else None
        arr[i].w_o = d["w_o"].data_ptr() if d.get("w_o") is not None else None
        arr[i].cs, arr[i].hs = d["cs"].data_ptr(), d["hs"].data_ptr()
        arr[i].reverse = int(d["reverse"])
        h16 = d.get("hs_bf16") if bf16 else None          # optional fused bf16 copy of hs (bf16 entry point only)
        if h16 is not None:
            _require_cuda(h16)
            assert h16.dtype == torch.bfloat16 and h16.is_contiguous() and h16.numel() == d["hs"].numel()
        arr[i].hs_bf16 = h16.data_ptr() if h16 is not None else None
        arr[i].shadow_only = int(bool(d.get("shadow_only")) and h16 is not None)      # fp32 hs unspecified afterwards
    # one workspace for both passes (it carries the sticky status word): sized for the larger (backward) one at once
    nbytes = max(lib.lc_lstm_fwd_workspace_bytes(B, N, len(dirs)), lib.lc_lstm_bwd_workspace_bytes(B, N, len(dirs)))
    ws = workspace("lstm", nbytes, dirs[0]["zx"].device)
    ev = _prof_begin()
    fn, who = ((lib.lc_lstm_fwd_bf16, "lc_lstm_fwd_bf16") if bf16 else
               (lib.lc_lstm_fwd_x3, "lc_lstm_fwd_x3") if x3 else (lib.lc_lstm_fwd, "lc_lstm_fwd"))
    _lib.check(fn(ctypes.cast(arr, ctypes.c_void_p), len(dirs), _ptr(seq_len), T, B, N,
                  float(forget_bias), _ptr(ws), nbytes, _stream()), who)
    _prof_end("lstm_fwd", 2.0 * len(dirs) * T * B * N * 4 * N, ev)


def lstm_bwd(dirs, seq_len, T, B, N, bf16=False, x3=False):
    """dirs: list of dict(gates, RT, w_f, w_i, w_o, cs, dh, dpeep, dbias, reverse).  gates -> dz in place;
    dpeep [3,N] and dbias [4N] are accumulated into (+=).
    bf16=True: the step GEMM's operands (dz_{t'}, R^T) are rounded to bf16 (lc_lstm_bwd_bf16).
    x3=True: split-operand step product (lc_lstm_bwd_x3), as in lstm_fwd."""
    assert not (bf16 and x3)
    lib = _lib.load()
    arr = (_lib.LstmBwdDir * len(dirs))()
    for i, d in enumerate(dirs):
        _require_cuda(d["gates"], d["RT"], d["cs"], d["dh"])
        arr[i].gates, arr[i].RT = d["gates"].data_ptr(), d["RT"].data_ptr()
        arr[i].w_f = d["w_f"].data_ptr() if d.get("w_f") is not None else None
        arr[i].w_i = d["w_i"].data_ptr() if d.get("w_i") is not None else None
        arr[i].w_o = d["w_o"].data_ptr() if d.get("w_o") is not None else None
        arr[i].cs, arr[i].dh = d["cs"].data_ptr(), d["dh"].data_ptr()
        arr[i].dpeep = d["dpeep"].data_ptr() if d.get("dpeep") is not None else None
        arr[i].dbias = d["dbias"].data_ptr() if d.get("dbias") is not None else None
        arr[i].reverse = int(d["reverse"])
        z16 = d.get("dz_bf16") if bf16 else None          # optional fused bf16 copy of dz (bf16 entry point only)
        if z16 is not None:
            _require_cuda(z16)
            assert z16.dtype == torch.bfloat16 and z16.is_contiguous() and z16.numel() == d["gates"].numel()
        if x3 and d.get("dz_x3") is not None:             # split-operand entry point: the x3 shadow of dz ([rows, 12 N])
            z16 = d["dz_x3"]
            _require_cuda(z16)
            assert z16.dtype == torch.bfloat16 and z16.is_contiguous() and z16.numel() == 3 * d["gates"].numel()
        arr[i].dz_bf16 = z16.data_ptr() if z16 is not None else None
        arr[i].shadow_only = int(bool(d.get("shadow_only")) and bf16 and z16 is not None)      # fp32 dz unspecified afterwards
    nbytes = max(lib.lc_lstm_fwd_workspace_bytes(B, N, len(dirs)), lib.lc_lstm_bwd_workspace_bytes(B, N, len(dirs)))
    ws = workspace("lstm", nbytes, dirs[0]["gates"].device)
    ev = _prof_begin()
    fn, who = ((lib.lc_lstm_bwd_bf16, "lc_lstm_bwd_bf16") if bf16 else
               (lib.lc_lstm_bwd_x3, "lc_lstm_bwd_x3") if x3 else (lib.lc_lstm_bwd, "lc_lstm_bwd"))
    _lib.check(fn(ctypes.cast(arr, ctypes.c_void_p), len(dirs), _ptr(seq_len), T, B, N, _ptr(ws),
                  nbytes, _stream()), who)
    _prof_end("lstm_bwd", 2.0 * len(dirs) * T * B * N * 4 * N, ev)


# ------------------------------------------------------------------------------------------ MoE head
def moe_combine_fwd(a, q, E, V, tau, keep, seed):
    lib = _lib.load()
    _require_cuda(a, q)
    R = a.shape[0]
    logits = torch.empty((R, V), dtype=torch.float32, device=a.device)
    pi = torch.empty((R, E), dtype=torch.float32, device=a.device)
    _lib.check(lib.lc_moe_combine_fwd(_ptr(a), _ptr(q), R, E, V, float(tau), float(keep), int(seed) & 0xFFFFFFFF,
                                      _ptr(logits), _ptr(pi), _stream()), "lc_moe_combine_fwd")
    return logits, pi


def moe_combine_bwd(pi, q, dlogits, E, V, tau, keep, seed):
    lib = _lib.load()
    _require_cuda(pi, q, dlogits)
    R = pi.shape[0]
    da = torch.empty((R, E), dtype=torch.float32, device=pi.device)
    _lib.check(lib.lc_moe_combine_bwd(_ptr(pi), _ptr(q), _ptr(_f32c(dlogits)), R, E, V, float(tau), float(keep),
                                      int(seed) & 0xFFFFFFFF, _ptr(da), _stream()), "lc_moe_combine_bwd")
    return da


# ------------------------------------------------------------------------------------------ optimizer
OPTIMIZERS = {"sgd": 0, "momentum": 1, "adam": 2}


def optimizer_step(params, grads, n_decay, l2, clip_norm, optimizer, lr, step, state, norm_out, guard=None):
    """guard: optional device int32 tensor; the update is skipped on the device when it is non-zero (lstm_status)."""
    lib = _lib.load()
    _require_cuda(params, grads, state, norm_out, guard)
    n = params.numel()
    nbytes = lib.lc_optimizer_workspace_bytes(n)
    ws = workspace("optim", nbytes, params.device)
    _lib.check(lib.lc_optimizer_step(_ptr(params), _ptr(grads), n, int(n_decay), float(l2), float(clip_norm),
                                     OPTIMIZERS[optimizer], float(lr), int(step), _ptr(state), _ptr(norm_out),
                                     _ptr(guard), _ptr(ws), nbytes, _stream()), "lc_optimizer_step")


def label_smoothing(logits, weight, log_q=None, dlogits=None):
    """Returns the device double scalar w*sum p(log p - log q) over all rows; accumulates the gradient into dlogits."""
    lib = _lib.load()
    _require_cuda(logits, log_q, dlogits)
    logits = _f32c(logits)
    rows, V = logits.shape
    acc = torch.zeros(1, dtype=torch.float64, device=logits.device)
    _lib.check(lib.lc_label_smoothing(_ptr(logits), rows, V, _ptr(log_q), float(weight), _ptr(acc), _ptr(dlogits),
                                      _stream()), "lc_label_smoothing")
    return acc


def posteriors(logits, smooth=1.0, apply_softmax=True, apply_log=True, log_prior=None):
    lib = _lib.load()
    _require_cuda(logits, log_prior)
    logits = _f32c(logits)
    rows, V = logits.shape
    out = torch.empty_like(logits)
    _lib.check(lib.lc_posteriors(_ptr(logits), rows, V, float(smooth), int(apply_softmax), int(apply_log),
                                 _ptr(log_prior), _ptr(out), _stream()), "lc_posteriors")
    return out


# ------------------------------------------------------------------------------------------ batch normalisation
BN_EPS, BN_MOMENTUM = 1e-3, 0.99          # tf.layers.batch_normalization defaults (nnet/lstm.py:273,290)


def bn_forward(x, gamma, beta, training, moving_mean, moving_var, out=None):
    """tf.layers.batch_normalization on x [rows, C] (nnet/lstm.py:271-294).  training: batch moments over all rows
    (returned for the backward and the moving-average update); else the moving averages.  -> (y, mean, var)."""
    lib = _lib.load()
    _require_cuda(x, gamma, beta, moving_mean, moving_var)
    x, ldx = _rowmajor2d(x)
    rows, C = x.shape
    if out is None:
        out = torch.empty((rows, C), dtype=torch.float32, device=x.device)
    if training:
        mean = torch.empty(C, dtype=torch.float32, device=x.device)
        var = torch.empty(C, dtype=torch.float32, device=x.device)
        nbytes = lib.lc_bn_workspace_bytes(C)
        ws = workspace("bn", nbytes, x.device)
        _lib.check(lib.lc_bn_moments(_ptr(x), rows, C, ldx, _ptr(mean), _ptr(var), _ptr(ws), nbytes, _stream()),
                   "lc_bn_moments")
    else:
        mean, var = moving_mean, moving_var
    _lib.check(lib.lc_bn_apply(_ptr(x), rows, C, ldx, _ptr(mean), _ptr(var), _ptr(gamma), _ptr(beta), BN_EPS,
                               _ptr(out), out.stride(0), _stream()), "lc_bn_apply")
    return out, mean, var


def bn_backward(x, dy, mean, var, gamma, training, dgamma, dbeta, dx=None):
    """Gradient of bn_forward; dgamma / dbeta are overwritten; dx may be dy (in place)."""
    lib = _lib.load()
    _require_cuda(x, dy, mean, var, gamma, dgamma, dbeta)
    x, ldx = _rowmajor2d(x)
    dy, lddy = _rowmajor2d(dy)
    rows, C = x.shape
    if dx is None:
        dx = torch.empty((rows, C), dtype=torch.float32, device=x.device)
    nbytes = lib.lc_bn_workspace_bytes(C)
    ws = workspace("bn", nbytes, x.device)
    _lib.check(lib.lc_bn_bwd(_ptr(x), _ptr(dy), rows, C, ldx, lddy, _ptr(mean), _ptr(var), _ptr(gamma), BN_EPS,
                             int(bool(training)), _ptr(dx), dx.stride(0), _ptr(dgamma), _ptr(dbeta), _ptr(ws), nbytes,
                             _stream()), "lc_bn_bwd")
    return dx


def bn_update_moving(moving_mean, moving_var, mean, var):
    lib = _lib.load()
    _require_cuda(moving_mean, moving_var, mean, var)
    _lib.check(lib.lc_bn_update_moving(_ptr(moving_mean), _ptr(moving_var), _ptr(mean), _ptr(var),
                                       moving_mean.numel(), BN_MOMENTUM, _stream()), "lc_bn_update_moving")


def length_mask_(x, seq_len, T, B):
    """Zero the rows of the time-major x [T*B, C] that lie beyond each utterance's length (in place)."""
    lib = _lib.load()
    _require_cuda(x, seq_len)
    assert x.dim() == 2 and x.stride(1) == 1 and x.shape[0] == T * B
    _lib.check(lib.lc_length_mask(_ptr(x), T, B, x.shape[1], x.stride(0), _ptr(seq_len), _stream()), "lc_length_mask")
    return x


# ------------------------------------------------------------------------------------------ bf16 shadow operands (c5)
def cast_bf16(x, nat=True, tr=False, out_nat=None):
    """bf16 copies of the float32 matrix x [rows, C]: (nat [rows, C] or None, tr [C, rows8] or None), rows8 = rows
    rounded up to a multiple of 8 (the transposed copy's row pitch; its pad columns are zero).  out_nat: an existing
    bf16 [rows, >= C] tensor (last stride 1) to receive the natural copy in its first C columns (the rest is left alone:
    a zero-padded operand of a 256-wide tile)."""
    lib = _lib.load()
    _require_cuda(x, out_nat)
    x, ldx = _rowmajor2d(x)
    rows, C = x.shape
    ldn = C
    if out_nat is not None:
        assert (out_nat.dtype == torch.bfloat16 and out_nat.dim() == 2 and out_nat.shape[0] == rows and out_nat.shape[1] >= C
                and out_nat.stride(1) == 1)
        nat = True
        ldn = out_nat.stride(0) if rows > 1 else max(out_nat.stride(0), C)
    n = out_nat if out_nat is not None else (torch.empty((rows, C), dtype=torch.bfloat16, device=x.device) if nat else None)
    t = None
    if tr:
        rows8 = (rows + 7) // 8 * 8
        t = (torch.zeros if rows8 != rows else torch.empty)((C, rows8), dtype=torch.bfloat16, device=x.device)
    ev = _prof_begin()
    _lib.check(lib.lc_cast_bf16(_ptr(x), rows, C, ldx, _ptr(n), ldn, _ptr(t), t.shape[1] if t is not None else 0,
                                _stream()), "lc_cast_bf16")
    _prof_end("cast_bf16", float(rows) * C * (4 + 2 * (int(nat) + int(tr))), ev)      # bytes moved
    return n, t


def gemm_bf16_nt(A, B, out=None, alpha=1.0, beta=0.0, bias=None, K=None, epilogue=None):
    """out[M,N] = alpha * A[M,K] @ B[N,K]^T + beta*out (+ bias) on bf16 shadow operands (k contiguous in both).
    K defaults to A.shape[1] (pass the true K when the operands carry zero pad columns)."""
    lib = _lib.load()
    _require_cuda(A, B, out, bias)
    assert A.dtype == torch.bfloat16 and B.dtype == torch.bfloat16 and A.stride(1) == 1 and B.stride(1) == 1
    M, N = A.shape[0], B.shape[0]
    K = A.shape[1] if K is None else K
    assert B.shape[1] >= K and A.shape[1] >= K
    if out is None:
        assert beta == 0.0
        out = torch.empty((M, N), dtype=torch.float32, device=A.device)
    assert out.shape == (M, N) and out.stride(1) == 1 and out.dtype == torch.float32
    lda = A.stride(0) if M > 1 else max(A.stride(0), A.shape[1])
    ldb = B.stride(0) if N > 1 else max(B.stride(0), B.shape[1])
    ldc = out.stride(0) if M > 1 else max(out.stride(0), N)
    nbytes = lib.lc_gemm_workspace_bytes(M, N, K)
    ws = workspace("gemm", nbytes, A.device) if nbytes else None
    ev = _prof_begin()
    _arm(lib, epilogue, out)
    _lib.check(lib.lc_gemm_bf16_nt(M, N, K, alpha, _ptr(A), lda, _ptr(B), ldb, beta, _ptr(out), ldc, _ptr(bias),
                                   _ptr(ws), nbytes, _stream()), "lc_gemm_bf16_nt")
    _prof_end("gemm_bf16", 2.0 * M * N * K, ev)
    return out


def gemm_nt2(A1, B1, A2, B2, out=None, alpha=1.0, beta=0.0, bias=None, epilogue=None):
    """out[M,N] = alpha * (A1 @ B1^T + A2 @ B2^T) + beta*out (+ bias) in one pass over out, float32 (lc_gemm_f32_nt2): A [M,K],
    B [N,K] row-major views with last stride 1, both pairs with the same row strides - the dX of a bidirectional layer."""
    lib = _lib.load()
    _require_cuda(A1, B1, A2, B2, out, bias)
    for t in (A1, B1, A2, B2):
        assert t.dtype == torch.float32 and t.dim() == 2 and t.stride(1) == 1
    M, N, K1, K2 = A1.shape[0], B1.shape[0], A1.shape[1], A2.shape[1]
    assert A2.shape[0] == M and B2.shape[0] == N and B1.shape[1] == K1 and B2.shape[1] == K2
    assert A1.stride(0) == A2.stride(0) and B1.stride(0) == B2.stride(0) and M > 1 and N > 1
    if out is None:
        assert beta == 0.0
        out = torch.empty((M, N), dtype=torch.float32, device=A1.device)
    assert out.shape == (M, N) and out.stride(1) == 1 and out.dtype == torch.float32
    ev = _prof_begin()
    _arm(lib, epilogue, out)
    _lib.check(lib.lc_gemm_f32_nt2(M, N, K1, K2, alpha, _ptr(A1), _ptr(A2), A1.stride(0), _ptr(B1), _ptr(B2), B1.stride(0),
                                   beta, _ptr(out), out.stride(0), _ptr(bias), _stream()), "lc_gemm_f32_nt2")
    _prof_end("gemm", 2.0 * M * N * (K1 + K2), ev)
    return out


def gemm_bf16_nt2(A1, B1, A2, B2, out=None, alpha=1.0, beta=0.0, bias=None, epilogue=None):
    """out[M,N] = alpha * (A1 @ B1^T + A2 @ B2^T) + beta*out (+ bias) in one pass over out (lc_gemm_bf16_nt2): bf16 shadow
    operands, k contiguous, both pairs with the same row strides - the dX of a bidirectional layer."""
    lib = _lib.load()
    _require_cuda(A1, B1, A2, B2, out, bias)
    for t in (A1, B1, A2, B2):
        assert t.dtype == torch.bfloat16 and t.dim() == 2 and t.stride(1) == 1
    M, N, K1, K2 = A1.shape[0], B1.shape[0], A1.shape[1], A2.shape[1]
    assert A2.shape[0] == M and B2.shape[0] == N and B1.shape[1] == K1 and B2.shape[1] == K2
    assert A1.stride(0) == A2.stride(0) and B1.stride(0) == B2.stride(0) and M > 1 and N > 1
    if out is None:
        assert beta == 0.0
        out = torch.empty((M, N), dtype=torch.float32, device=A1.device)
    assert out.shape == (M, N) and out.stride(1) == 1 and out.dtype == torch.float32
    ev = _prof_begin()
    _arm(lib, epilogue, out)
    _lib.check(lib.lc_gemm_bf16_nt2(M, N, K1, K2, alpha, _ptr(A1), _ptr(A2), A1.stride(0), _ptr(B1), _ptr(B2), B1.stride(0),
                                    beta, _ptr(out), out.stride(0), _ptr(bias), _stream()), "lc_gemm_bf16_nt2")
    _prof_end("gemm_bf16", 2.0 * M * N * (K1 + K2), ev)
    return out


def gemm_bf16_tn(A, B, out=None, alpha=1.0, beta=0.0, bias=None, epilogue=None):
    """out[M,N] = alpha * A^T @ B + beta*out (+ bias) on bf16 operands that are both K-MAJOR: A [K,M], B [K,N] (row
    windows of natural-layout shadows are fine: only the last stride must be 1).  M, N multiples of 256."""
    lib = _lib.load()
    _require_cuda(A, B, out, bias)
    assert A.dtype == torch.bfloat16 and B.dtype == torch.bfloat16 and A.stride(1) == 1 and B.stride(1) == 1
    K, M = A.shape
    assert B.shape[0] == K
    N = B.shape[1]
    if out is None:
        assert beta == 0.0
        out = torch.empty((M, N), dtype=torch.float32, device=A.device)
    assert out.shape == (M, N) and out.stride(1) == 1 and out.dtype == torch.float32
    lda = A.stride(0) if K > 1 else max(A.stride(0), M)
    ldb = B.stride(0) if K > 1 else max(B.stride(0), N)
    ldc = out.stride(0) if M > 1 else max(out.stride(0), N)
    nbytes = lib.lc_gemm_workspace_bytes(M, N, K)
    ws = workspace("gemm", nbytes, A.device) if nbytes else None
    ev = _prof_begin()
    _arm(lib, epilogue, out)
    _lib.check(lib.lc_gemm_bf16_tn(M, N, K, alpha, _ptr(A), lda, _ptr(B), ldb, beta, _ptr(out), ldc, _ptr(bias),
                                   _ptr(ws), nbytes, _stream()), "lc_gemm_bf16_tn")
    _prof_end("gemm_bf16", 2.0 * M * N * K, ev)
    return out


def gemm_bf16_nn(A, B, out=None, alpha=1.0, beta=0.0, bias=None, epilogue=None):
    """out[M,N] = alpha * A @ B + beta*out (+ bias) on bf16 operands in their NATURAL layouts: A [M,K] (k contiguous), B [K,N]
    (K-major).  M, N multiples of 256, K of 64."""
    lib = _lib.load()
    _require_cuda(A, B, out, bias)
    assert A.dtype == torch.bfloat16 and B.dtype == torch.bfloat16 and A.stride(1) == 1 and B.stride(1) == 1
    M, K = A.shape
    assert B.shape[0] == K
    N = B.shape[1]
    if out is None:
        assert beta == 0.0
        out = torch.empty((M, N), dtype=torch.float32, device=A.device)
    assert out.shape == (M, N) and out.stride(1) == 1 and out.dtype == torch.float32
    lda = A.stride(0) if M > 1 else max(A.stride(0), K)
    ldb = B.stride(0) if K > 1 else max(B.stride(0), N)
    ldc = out.stride(0) if M > 1 else max(out.stride(0), N)
    nbytes = lib.lc_gemm_workspace_bytes(M, N, K)
    ws = workspace("gemm", nbytes, A.device) if nbytes else None
    ev = _prof_begin()
    _arm(lib, epilogue, out)
    _lib.check(lib.lc_gemm_bf16_nn(M, N, K, alpha, _ptr(A), lda, _ptr(B), ldb, beta, _ptr(out), ldc, _ptr(bias),
                                   _ptr(ws), nbytes, _stream()), "lc_gemm_bf16_nn")
    _prof_end("gemm_bf16", 2.0 * M * N * K, ev)
    return out


# ------------------------------------------------------------------------------------------ fp32 products as bf16 x 3
def split_bf16x3(x):
    """x [rows, K] f32 (last stride 1) -> its x3 shadow [rows, 3 * roundup(K, 16)] bf16 (lc_split_bf16x3)."""
    lib = _lib.load()
    _require_cuda(x)
    assert x.dim() == 2 and x.dtype == torch.float32 and x.stride(1) == 1
    rows, K = x.shape
    ldo = 3 * ((K + 15) // 16 * 16)
    out = torch.empty((rows, ldo), dtype=torch.bfloat16, device=x.device)
    ldx = x.stride(0) if rows > 1 else max(x.stride(0), K)
    ev = _prof_begin()
    _lib.check(lib.lc_split_bf16x3(_ptr(x), rows, K, ldx, _ptr(out), ldo, _stream()), "lc_split_bf16x3")
    _prof_end("cast_bf16", float(rows) * K * 10, ev)
    return out


def gemm_bf16x3_nt(A3, B3, K, out=None, alpha=1.0, beta=0.0, bias=None, epilogue=None):
    """out[M,N] = alpha * A @ B^T + beta*out (+ bias) from the x3 shadows of A [M,K] and B [N,K]: fp32-grade result on the
    bf16 matrix cores (six bf16 term products per element product)."""
    lib = _lib.load()
    _require_cuda(A3, B3, out, bias)
    assert A3.dtype == torch.bfloat16 and B3.dtype == torch.bfloat16 and A3.stride(1) == 1 and B3.stride(1) == 1
    M, N = A3.shape[0], B3.shape[0]
    if out is None:
        assert beta == 0.0
        out = torch.empty((M, N), dtype=torch.float32, device=A3.device)
    assert out.shape == (M, N) and out.stride(1) == 1 and out.dtype == torch.float32
    lda = A3.stride(0) if M > 1 else max(A3.stride(0), A3.shape[1])
    ldb = B3.stride(0) if N > 1 else max(B3.stride(0), B3.shape[1])
    ldc = out.stride(0) if M > 1 else max(out.stride(0), N)
    ev = _prof_begin()
    _arm(lib, epilogue, out)
    _lib.check(lib.lc_gemm_bf16x3_nt(M, N, K, alpha, _ptr(A3), lda, _ptr(B3), ldb, beta, _ptr(out), ldc, _ptr(bias),
                                     _stream()), "lc_gemm_bf16x3_nt")
    _prof_end("gemm_x3", 2.0 * M * N * K, ev)
    return out


def gemm_bf16x3_tn(A3, B3, M, N, out=None, alpha=1.0, beta=0.0, bias=None):
    """out[M,N] = alpha * A^T @ B + beta*out (+ bias) from the x3 shadows of A [K,M] and B [K,N] (both K-major; row windows
    of larger shadows are fine): the weight gradients, fp32-grade, on the bf16 matrix cores."""
    lib = _lib.load()
    _require_cuda(A3, B3, out, bias)
    assert A3.dtype == torch.bfloat16 and B3.dtype == torch.bfloat16 and A3.stride(1) == 1 and B3.stride(1) == 1
    K = A3.shape[0]
    assert B3.shape[0] == K and A3.shape[1] >= 3 * ((M + 15) // 16 * 16) and B3.shape[1] >= 3 * ((N + 15) // 16 * 16)
    if out is None:
        assert beta == 0.0
        out = torch.empty((M, N), dtype=torch.float32, device=A3.device)
    assert out.shape == (M, N) and out.stride(1) == 1 and out.dtype == torch.float32
    lda = A3.stride(0) if K > 1 else max(A3.stride(0), A3.shape[1])
    ldb = B3.stride(0) if K > 1 else max(B3.stride(0), B3.shape[1])
    ldc = out.stride(0) if M > 1 else max(out.stride(0), N)
    nbytes = lib.lc_gemm_bf16x3_tn_workspace_bytes_ld(M, N, K, lda, ldb)     # K slices: CU fill, and the 2 GB descriptor reach
    ws = workspace("gemm_x3", nbytes, A3.device) if nbytes else None
    ev = _prof_begin()
    _lib.check(lib.lc_gemm_bf16x3_tn(M, N, K, alpha, _ptr(A3), lda, _ptr(B3), ldb, beta, _ptr(out), ldc, _ptr(bias),
                                     _ptr(ws), nbytes, _stream()), "lc_gemm_bf16x3_tn")
    _prof_end("gemm_x3", 2.0 * M * N * K, ev)
    return out
