"""CPU oracle for the mobvoi/lstm_ctc hot path — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module, and only as the checker / the reported CPU baseline.  The product
package ``lstm_ctc_amd`` never imports it.

PARITY UNPINNED BY THE REFERENCE: mobvoi/lstm_ctc ships no tests or golden vectors and its
arithmetic lives in the un-vendored dependency tensorflow==1.8.0 (reference README.md:23),
which cannot be installed here.  The restatement is pinned by TF-upstream CTC known answers,
torch-CPU cross-checks and fp64 finite-difference checks (tests/test_oracle_*.py).

The heavy arithmetic is in ``oracle.c`` / ``oracle_core.inc`` (plain C, float32 and float64
instantiations); this file is the ctypes binding plus the numpy glue that follows the
reference's graph assembly:

* BiLSTM stack ............ nnet/bilstm.py:107-273
* uni-LSTM stack .......... nnet/lstm.py:125-368 (intent; the shipped code cannot run)
* MoE head ................ nnet/moe.py:29-72
* loss / clip / optimizer . nnet/graph.py:51-209
* running means / logs .... nnet/funcs.py:23-152

All tensors are batch-major ``[B, T, *]`` like the reference's.
"""
import ctypes
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    """Compile liboracle.so with gcc (the oracle's own Makefile)."""
    subprocess.check_call(["make", "-s", "-C", _HERE])


def usable_cpus():
    """CPUs this process may actually use: its affinity mask, capped by the container's CPU quota (cgroup v2
    ``cpu.max`` / v1 ``cpu.cfs_quota_us``) - the GPU boxes show 256 logical CPUs behind a 16-CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(math.ceil(int(quota) / int(period)))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and per > 0:
                n = min(n, max(1, int(math.ceil(q / per))))
        except (OSError, ValueError):
            pass
    env = os.environ.get("OMP_NUM_THREADS")
    if env and env.isdigit() and int(env) > 0:
        n = int(env)
    return max(1, n)


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        _LIB = ctypes.CDLL(path)
        if not hasattr(_LIB, "orc_set_num_threads") or not hasattr(_LIB, "orc_lstmp_fwd_init_f64"):      # a stale build
            build()
            _LIB = ctypes.CDLL(path)
        _LIB.orc_set_num_threads(usable_cpus())
    return _LIB


def _sfx(dtype):
    return "_f64" if np.dtype(dtype) == np.float64 else "_f32"


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _c(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


def _real(dtype):
    return ctypes.c_double if np.dtype(dtype) == np.float64 else ctypes.c_float


def num_threads():
    return int(lib().orc_num_threads())


# ----------------------------------------------------------------------------- primitives
def reverse_sequence(x, seq_len):
    """tf.reverse_sequence(x, len, seq_axis=1, batch_axis=0) — nnet/bilstm.py:112."""
    x = np.ascontiguousarray(x)
    B, T, D = x.shape
    y = np.empty_like(x)
    sl = _c(seq_len, np.int32)
    getattr(lib(), "orc_reverse_sequence" + _sfx(x.dtype))(_p(x), _p(sl), B, T, D, _p(y))
    return y


def lstmp_fwd(x, seq_len, kernel, bias, w_f, w_i, w_o, proj, forget_bias, init_c=None, init_m=None):
    """One dynamic_rnn(LSTMCell) — nnet/bilstm.py:125-188.  Returns (out, saved).  The reference always starts from the zero
    state (bilstm.py:140-144); `init_c` / `init_m` exist for the known-answer vector of TF's own cell test only."""
    dt = x.dtype
    x = _c(x, dt)
    B, T, I = x.shape
    N = bias.shape[0] // 4
    P = proj.shape[1] if proj is not None else 0
    Pout = P if proj is not None else N
    assert kernel.shape == (I + Pout, 4 * N), (kernel.shape, I, Pout, N)
    out = np.empty((B, T, Pout), dt)
    gates = np.empty((B, T, 4 * N), dt)
    cs = np.empty((B, T, N), dt)
    mp = np.empty((B, T, N), dt)
    fc = np.empty((B, N), dt)
    fm = np.empty((B, Pout), dt)
    sl = _c(seq_len, np.int32)
    args = [_c(a, dt) if a is not None else None for a in (kernel, bias, w_f, w_i, w_o, proj)]
    ic = _c(init_c, dt) if init_c is not None else None
    im = _c(init_m, dt) if init_m is not None else None
    getattr(lib(), "orc_lstmp_fwd_init" + _sfx(dt))(
        _p(x), _p(sl), B, T, I, N, P, *[_p(a) for a in args], _real(dt)(forget_bias),
        _p(out), _p(gates), _p(cs), _p(mp), _p(fc), _p(fm), _p(ic), _p(im))
    saved = dict(x=x, seq_len=sl, out=out, gates=gates, cs=cs, mp=mp, final_c=fc, final_m=fm)
    return out, saved


def lstmp_bwd(saved, kernel, w_f, w_i, w_o, proj, d_out):
    """BPTT through lstmp_fwd.  Returns (dx, grads dict)."""
    x = saved["x"]
    dt = x.dtype
    B, T, I = x.shape
    N = saved["cs"].shape[2]
    P = proj.shape[1] if proj is not None else 0
    dx = np.empty_like(x)
    g = dict(kernel=np.zeros_like(kernel, dtype=dt), bias=np.zeros(4 * N, dt))
    for name, w in (("w_f_diag", w_f), ("w_i_diag", w_i), ("w_o_diag", w_o)):
        g[name] = np.zeros(N, dt) if w is not None else None
    g["proj"] = np.zeros_like(proj, dtype=dt) if proj is not None else None
    args = [_c(a, dt) if a is not None else None for a in (kernel, w_f, w_i, w_o, proj)]
    d_out = _c(d_out, dt)
    getattr(lib(), "orc_lstmp_bwd" + _sfx(dt))(
        _p(x), _p(saved["seq_len"]), B, T, I, N, P, *[_p(a) for a in args],
        _p(saved["out"]), _p(saved["gates"]), _p(saved["cs"]), _p(saved["mp"]), _p(d_out),
        _p(dx), _p(g["kernel"]), _p(g["bias"]), _p(g["w_f_diag"]), _p(g["w_i_diag"]),
        _p(g["w_o_diag"]), _p(g["proj"]))
    return dx, g


def moe_fwd(h, Wp, bp, W, b, tau, drop_pi=None, drop_z=None):
    """create_moe — nnet/moe.py:29-72.  h [R,H] → y [R,V]."""
    dt = h.dtype
    h = _c(h, dt)
    R, H = h.shape
    E = Wp.shape[1]
    V = W.shape[1] // E
    y = np.empty((R, V), dt)
    pi_s = np.empty((R, E), dt)
    zt = np.empty((R, E * V), dt)
    a = [_c(v, dt) if v is not None else None for v in (Wp, bp, W, b, drop_pi, drop_z)]
    getattr(lib(), "orc_moe_fwd" + _sfx(dt))(
        _p(h), R, H, E, V, _p(a[0]), _p(a[1]), _p(a[2]), _p(a[3]), _real(dt)(tau),
        _p(a[4]), _p(a[5]), _p(y), _p(pi_s), _p(zt))
    return y, dict(h=h, pi_s=pi_s, zt=zt, drop_pi=a[4], drop_z=a[5])


def moe_bwd(saved, Wp, W, tau, dy):
    h = saved["h"]
    dt = h.dtype
    R, H = h.shape
    E = Wp.shape[1]
    V = W.shape[1] // E
    dh = np.empty((R, H), dt)
    g = dict(Wp=np.zeros((H, E), dt), bp=np.zeros(E, dt), W=np.zeros((H, E * V), dt), b=np.zeros(E * V, dt))
    getattr(lib(), "orc_moe_bwd" + _sfx(dt))(
        _p(h), R, H, E, V, _p(_c(Wp, dt)), _p(_c(W, dt)), _real(dt)(tau), _p(saved["drop_pi"]),
        _p(saved["drop_z"]), _p(saved["pi_s"]), _p(saved["zt"]), _p(_c(dy, dt)), _p(dh),
        _p(g["Wp"]), _p(g["bp"]), _p(g["W"]), _p(g["b"]))
    return dh, g


def gemm(A, B, ta=False, tb=False):
    dt = A.dtype
    A = _c(A, dt)
    B = _c(B, dt)
    M, K = (A.shape[1], A.shape[0]) if ta else A.shape
    N = B.shape[0] if tb else B.shape[1]
    C = np.empty((M, N), dt)
    r = _real(dt)
    getattr(lib(), "orc_gemm" + _sfx(dt))(int(ta), int(tb), M, N, K, r(1), _p(A), A.shape[1],
                                          _p(B), B.shape[1], r(0), _p(C), N)
    return C


def bf16_round(x):
    """float32 -> nearest bf16 (round-to-nearest-even), returned as float32: the operand rounding of the
    c5 configuration ("bf16 MFMA gate GEMMs"); what v_cvt_pk_bf16_f32 does for finite values."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    r = (u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) & np.uint32(0xFFFF0000)
    return r.view(np.float32)


def flatten_labels(dense):
    """Dense [B,Lmax] int64 labels padded with -1 → (flat int32, offsets[B+1]) — the
    tf.where/gather_nd/SparseTensor conversion of nnet/graph.py:76-104."""
    dense = np.asarray(dense)
    flat, offs = [], [0]
    for row in dense:
        keep = row[row != -1]
        flat.extend(int(v) for v in keep)
        offs.append(len(flat))
    return np.asarray(flat, np.int32), np.asarray(offs, np.int32)


def ctc_loss(logits_tbv, labels_flat, offsets, seq_len, want_grad=True):
    """tf.nn.ctc_loss as called at nnet/graph.py:109-114 (time-major logits [T,B,V])."""
    dt = logits_tbv.dtype
    x = _c(logits_tbv, dt)
    T, B, V = x.shape
    loss = np.empty(B, dt)
    grad = np.empty_like(x) if want_grad else None
    n_bad = getattr(lib(), "orc_ctc_loss" + _sfx(dt))(
        _p(x), T, B, V, _p(_c(labels_flat, np.int32)), _p(_c(offsets, np.int32)),
        _p(_c(seq_len, np.int32)), _p(loss), _p(grad))
    return loss, grad, int(n_bad)


def ctc_greedy(logits_tbv, seq_len):
    """tf.nn.ctc_greedy_decoder(merge_repeated=True) — nnet/graph.py:138-142."""
    dt = logits_tbv.dtype
    x = _c(logits_tbv, dt)
    T, B, V = x.shape
    tokens = np.zeros((B, T), np.int32)
    out_len = np.zeros(B, np.int32)
    nsl = np.zeros(B, dt)
    getattr(lib(), "orc_ctc_greedy" + _sfx(dt))(_p(x), T, B, V, _p(_c(seq_len, np.int32)),
                                                _p(tokens), _p(out_len), _p(nsl))
    return tokens, out_len, nsl


def edit_distance(hyp, hyp_len, truth_flat, offsets):
    """tf.edit_distance(normalize=False) per utterance — nnet/graph.py:143-149."""
    hyp = _c(hyp, np.int32)
    B = hyp.shape[0]
    dist = np.zeros(B, np.int32)
    lib().orc_edit_distance(_p(hyp), hyp.shape[1], _p(_c(hyp_len, np.int32)),
                            _p(_c(truth_flat, np.int32)), _p(_c(offsets, np.int32)), B, _p(dist))
    return dist


# ----------------------------------------------------------------------------- dropout mask
def _fmix32(h):
    h = h.copy()
    h ^= h >> np.uint32(16)
    h *= np.uint32(0x85EBCA6B)
    h ^= h >> np.uint32(13)
    h *= np.uint32(0xC2B2AE35)
    h ^= h >> np.uint32(16)
    return h


def dropout_mask(seed, stream, shape_tbp, keep, dtype=np.float32):
    """Counter-based Bernoulli(keep)/keep mask shared bit-for-bit with the device kernels
    (lstm_ctc_amd/csrc/common.h: lc_dropout_scale).  Index = linear offset in the
    time-major [T,B,P] tensor.  Returns the mask in time-major shape."""
    n = int(np.prod(shape_tbp))
    if keep >= 1.0:
        return np.ones(shape_tbp, dtype)
    idx = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        h = np.uint32((seed * 0x9E3779B1) & 0xFFFFFFFF) ^ np.uint32((stream * 0x85EBCA77 + 0x165667B1) & 0xFFFFFFFF)
        lo = (idx & np.uint64(0xFFFFFFFF)).astype(np.uint32)
        hi = (idx >> np.uint64(32)).astype(np.uint32)
        v = _fmix32(lo ^ h)
        v = _fmix32(v ^ (hi + np.uint32(0x27D4EB2F)))
    u = (v >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    m = np.where(u < np.float32(keep), np.float32(1.0) / np.float32(keep), np.float32(0.0))
    return m.astype(dtype).reshape(shape_tbp)


# ----------------------------------------------------------------------------- model glue
def lstm_param_names(prefix):
    return {k: prefix + "/" + k for k in
            ("kernel", "bias", "w_f_diag", "w_i_diag", "w_o_diag", "projection/kernel")}


def _cellp(params, prefix):
    g = lambda k: params.get(prefix + "/" + k)
    return g("kernel"), g("bias"), g("w_f_diag"), g("w_i_diag"), g("w_o_diag"), g("projection/kernel")


def blstm_forward(params, cfg, x, seq_len, drop_seed=0):
    """create_logits_blstm — nnet/bilstm.py:25-273.  x [B,T,D] → logits [B,T,V], saved."""
    dt = x.dtype
    nl = cfg["num_layers"]
    keep = 1.0 if not cfg.get("is_training", True) else float(cfg.get("dropout_rate", 1.0))
    B, T, D = x.shape
    finput = x
    binput = reverse_sequence(x, seq_len)                                   # bilstm.py:112
    layers = []
    for i in range(nl):
        kf = _cellp(params, "fd%d/frnn%d" % (i, i))
        kb = _cellp(params, "bd%d/brnn%d" % (i, i))
        fo, sf = lstmp_fwd(finput, seq_len, *kf, forget_bias=5.0)            # bilstm.py:171-179
        bo, sb = lstmp_fwd(binput, seq_len, *kb, forget_bias=5.0)            # bilstm.py:180-188
        P = fo.shape[2]
        mf = mb = None
        rbo = reverse_sequence(bo, seq_len)                                  # bilstm.py:190
        if keep < 1.0:                                                       # DropoutWrapper, bilstm.py:128,149
            mf = dropout_mask(drop_seed, 2 * i, (T, B, P), keep, dt).transpose(1, 0, 2)
            mb = dropout_mask(drop_seed, 2 * i + 1, (T, B, P), keep, dt).transpose(1, 0, 2)
            fo_d, rbo = fo * mf, rbo * mb
        else:
            fo_d = fo
        cat = np.concatenate([fo_d, rbo], axis=2)
        residual = (i == 0 and D == 2 * P)                                   # bilstm.py:199
        finput = finput + cat if residual else cat
        binput = reverse_sequence(finput, seq_len)                           # bilstm.py:203
        layers.append(dict(sf=sf, sb=sb, mf=mf, mb=mb, residual=residual, P=P))
    H = finput.reshape(B * T, -1)
    E = cfg.get("num_experts") or 0
    head = None
    if E > 0:                                                                # bilstm.py:229-236
        tau = cfg.get("moe_temp")
        tau = 10.0 if tau is None else tau
        V = cfg["num_targets"]
        dpi = dz = None
        if keep < 1.0:
            # rows are (b,t) batch-major here; masks are defined on time-major rows
            dpi = dropout_mask(drop_seed, 1000, (T, B, E), keep, dt).transpose(1, 0, 2).reshape(B * T, E)
            dz = dropout_mask(drop_seed, 1001, (T, B, E * V), keep, dt).transpose(1, 0, 2).reshape(B * T, E * V)
        y, head = moe_fwd(H, params["Variable"], params["Variable_1"], params["Variable_2"],
                          params["Variable_3"], tau, dpi, dz)
        head["tau"] = tau
    else:                                                                    # bilstm.py:237-249
        y = gemm(H, params["Variable"].astype(dt)) + params["Variable_1"].astype(dt)
    logits = y.reshape(B, T, -1)
    enc = np.concatenate([layers[-1]["sf"]["final_c"], layers[-1]["sf"]["final_m"],
                          layers[-1]["sb"]["final_c"], layers[-1]["sb"]["final_m"]], axis=1)  # bilstm.py:206-208
    return logits, dict(layers=layers, H=H, head=head, seq_len=np.asarray(seq_len, np.int32),
                        shape=(B, T, D), encoder=enc)


def blstm_backward(params, cfg, saved, dlogits):
    """Gradient of blstm_forward w.r.t. every parameter (tf.gradients at nnet/graph.py:190)."""
    B, T, D = saved["shape"]
    dt = dlogits.dtype
    seq_len = saved["seq_len"]
    grads = {}
    dy = dlogits.reshape(B * T, -1)
    E = cfg.get("num_experts") or 0
    if E > 0:
        dH, g = moe_bwd(saved["head"], params["Variable"], params["Variable_2"], saved["head"]["tau"], dy)
        grads.update({"Variable": g["Wp"], "Variable_1": g["bp"], "Variable_2": g["W"], "Variable_3": g["b"]})
    else:
        dH = gemm(dy, params["Variable"].astype(dt), tb=True)
        grads["Variable"] = gemm(saved["H"], dy, ta=True)
        grads["Variable_1"] = dy.sum(axis=0)
    dfin = dH.reshape(B, T, -1)
    for i in reversed(range(cfg["num_layers"])):
        L = saved["layers"][i]
        P = L["P"]
        dcat = dfin
        dfo, drbo = dcat[..., :P], dcat[..., P:]
        if L["mf"] is not None:
            dfo, drbo = dfo * L["mf"], drbo * L["mb"]
        dbo = reverse_sequence(np.ascontiguousarray(drbo), seq_len)
        pf, pb = "fd%d/frnn%d" % (i, i), "bd%d/brnn%d" % (i, i)
        outs = []
        for prefix, sv, d in ((pf, L["sf"], dfo), (pb, L["sb"], dbo)):
            k, _, wf, wi, wo, pj = _cellp(params, prefix)
            dx, g = lstmp_bwd(sv, k, wf, wi, wo, pj, np.ascontiguousarray(d))
            grads[prefix + "/kernel"] = g["kernel"]
            grads[prefix + "/bias"] = g["bias"]
            for nm in ("w_f_diag", "w_i_diag", "w_o_diag"):
                if g[nm] is not None:
                    grads[prefix + "/" + nm] = g[nm]
            if g["proj"] is not None:
                grads[prefix + "/projection/kernel"] = g["proj"]
            outs.append(dx)
        dprev = outs[0] + reverse_sequence(outs[1], seq_len)
        if L["residual"]:
            dprev = dprev + dcat
        dfin = dprev
    return grads, dfin


BN_EPS, BN_MOMENTUM = 1e-3, 0.99      # tf.layers.batch_normalization defaults (lstm.py:273,290 pass neither)


def _bn_names(cfg):
    """BatchNormalization layers of create_logits_lstm, in call order (lstm.py:271-294)."""
    if not cfg.get("use_bn"):
        return []
    return ["drnn_bn_0_0"] + ["drnn_bn%d" % i for i in range(cfg["num_layers"])]


def bn_forward(x2d, params, name, training):
    """tf.layers.batch_normalization on a rank-3 input (non-fused path): moments over every [B,T] position -
    padded frames included - population variance, y = (x - mean) * rsqrt(var + 1e-3) * gamma + beta; the
    moving averages replace the batch moments when training is False."""
    g, b = params[name + "/gamma"].astype(x2d.dtype), params[name + "/beta"].astype(x2d.dtype)
    if training:
        mean, var = x2d.mean(axis=0), x2d.var(axis=0)
    else:
        mean, var = params[name + "/moving_mean"].astype(x2d.dtype), params[name + "/moving_variance"].astype(x2d.dtype)
    inv = 1.0 / np.sqrt(var + x2d.dtype.type(BN_EPS))
    xhat = (x2d - mean) * inv
    return xhat * g + b, dict(xhat=xhat, inv=inv, mean=mean, var=var, training=training)


def bn_backward(sv, params, name, dy):
    g = params[name + "/gamma"].astype(dy.dtype)
    dgamma, dbeta = (dy * sv["xhat"]).sum(axis=0), dy.sum(axis=0)
    if sv["training"]:
        dx = g * sv["inv"] * (dy - dy.mean(axis=0) - sv["xhat"] * (dy * sv["xhat"]).mean(axis=0))
    else:
        dx = dy * g * sv["inv"]
    return dx, dgamma, dbeta


def bn_update_moving(params, saved):
    """The UPDATE_OPS the train op depends on (graph.py:194-196): assign_moving_average with momentum 0.99,
    variable -= (variable - batch_value) * (1 - momentum)."""
    for name, sv in saved.get("bn", {}).items():
        for key, val in (("/moving_mean", sv["mean"]), ("/moving_variance", sv["var"])):
            v = params[name + key]
            params[name + key] = (v - (v - val.astype(v.dtype)) * v.dtype.type(1.0 - BN_MOMENTUM)).astype(v.dtype)


def lstm_forward(params, cfg, x, seq_len, drop_seed=0):
    """create_logits_lstm (uni-LSTM, intent) — nnet/lstm.py:125-368: per layer
    DropoutWrapper(ResidualWrapper?(LSTMCell(N, P, use_peepholes=True, forget_bias=1.0)));
    residual on every layer except (i == 0 and input_dim != num_projects) (lstm.py:236-260);
    optional batch normalisation of the first layer's input and of every layer's output (lstm.py:271-294);
    affine head sigma = 1/sqrt(out_dim) (lstm.py:332-342)."""
    dt = x.dtype
    training = bool(cfg.get("is_training", True))
    keep = 1.0 if not training else float(cfg.get("dropout_rate", 1.0))
    use_bn = bool(cfg.get("use_bn"))
    B, T, D = x.shape
    inp = x
    bn = {}
    if use_bn:
        y2, bn["drnn_bn_0_0"] = bn_forward(inp.reshape(B * T, D), params, "drnn_bn_0_0", training)
        inp = y2.reshape(B, T, D)
    layers = []
    for i in range(cfg["num_layers"]):
        kp = _cellp(params, "drnn%d/lstm_cell" % i)
        o, sv = lstmp_fwd(inp, seq_len, *kp, forget_bias=1.0)
        P = o.shape[2]
        residual = not (i == 0 and D != P)
        mask = (np.arange(T)[None, :] < np.asarray(seq_len)[:, None]).astype(dt)[..., None]
        if residual:
            o = (o + inp) * mask            # ResidualWrapper adds the input before dynamic_rnn's length mask
        m = None
        if keep < 1.0:
            m = dropout_mask(drop_seed, 2 * i, (T, B, P), keep, dt).transpose(1, 0, 2)
            o = o * m
        if use_bn:
            y2, bn["drnn_bn%d" % i] = bn_forward(o.reshape(B * T, P), params, "drnn_bn%d" % i, training)
            o = y2.reshape(B, T, P)
        layers.append(dict(sv=sv, m=m, residual=residual, mask=mask))
        inp = o
    H = inp.reshape(B * T, -1)
    y = gemm(H, params["Variable"].astype(dt)) + params["Variable_1"].astype(dt)
    return y.reshape(B, T, -1), dict(layers=layers, H=H, seq_len=np.asarray(seq_len, np.int32), shape=(B, T, D),
                                     bn=bn)


def lstm_backward(params, cfg, saved, dlogits):
    B, T, D = saved["shape"]
    dt = dlogits.dtype
    dy = dlogits.reshape(B * T, -1)
    grads = {"Variable": gemm(saved["H"], dy, ta=True), "Variable_1": dy.sum(axis=0)}
    d = gemm(dy, params["Variable"].astype(dt), tb=True).reshape(B, T, -1)
    bn = saved.get("bn", {})

    def through_bn(name, d):
        dx, dg, db = bn_backward(bn[name], params, name, d.reshape(B * T, -1))
        grads[name + "/gamma"], grads[name + "/beta"] = dg, db
        return dx.reshape(d.shape)

    for i in reversed(range(cfg["num_layers"])):
        L = saved["layers"][i]
        if bn:
            d = through_bn("drnn_bn%d" % i, d)
        if L["m"] is not None:
            d = d * L["m"]
        prefix = "drnn%d/lstm_cell" % i
        k, _, wf, wi, wo, pj = _cellp(params, prefix)
        if L["residual"]:
            d = d * L["mask"]
        dx, g = lstmp_bwd(L["sv"], k, wf, wi, wo, pj, np.ascontiguousarray(d))
        grads[prefix + "/kernel"] = g["kernel"]
        grads[prefix + "/bias"] = g["bias"]
        for nm in ("w_f_diag", "w_i_diag", "w_o_diag"):
            if g[nm] is not None:
                grads[prefix + "/" + nm] = g[nm]
        if g["proj"] is not None:
            grads[prefix + "/projection/kernel"] = g["proj"]
        d = dx + d if L["residual"] else dx
    if bn:
        d = through_bn("drnn_bn_0_0", d)
    return grads, d


CUDNN_CELL = "rnn/multi_rnn_cell/cell_%d/cudnn_compatible_lstm_cell"      # MultiRNNCell / dynamic_rnn scopes, lstm.py:73-96


def cudnnlstm_forward(params, cfg, x, seq_len, drop_seed=0):
    """create_logits_cudnnlstm (intent) - nnet/lstm.py:26-122: num_layers x CudnnCompatibleLSTMCell(num_neurons) under one
    MultiRNNCell / dynamic_rnn - plain LSTM cells (LSTMBlockCell: gate order i, j, f, o like LSTMCell; forget_bias 0; no
    peepholes, no projection, no clipping), no dropout, no residual; affine head sigma = 1/sqrt(num_neurons) (lstm.py:105).
    dynamic_rnn's length masking of a MultiRNNCell (zero output, every layer's state carried through) equals masking layer
    by layer: a masked frame's output never reaches an unmasked frame of the same utterance."""
    dt = x.dtype
    B, T, D = x.shape
    inp = x
    layers = []
    for i in range(cfg["num_layers"]):
        k, b = params[CUDNN_CELL % i + "/kernel"], params[CUDNN_CELL % i + "/bias"]
        inp, sv = lstmp_fwd(inp, seq_len, k, b, None, None, None, None, forget_bias=0.0)
        layers.append(sv)
    H = inp.reshape(B * T, -1)
    y = gemm(H, params["Variable"].astype(dt)) + params["Variable_1"].astype(dt)
    return y.reshape(B, T, -1), dict(layers=layers, H=H, seq_len=np.asarray(seq_len, np.int32), shape=(B, T, D))


def cudnnlstm_backward(params, cfg, saved, dlogits):
    B, T, D = saved["shape"]
    dt = dlogits.dtype
    dy = dlogits.reshape(B * T, -1)
    grads = {"Variable": gemm(saved["H"], dy, ta=True), "Variable_1": dy.sum(axis=0)}
    d = gemm(dy, params["Variable"].astype(dt), tb=True).reshape(B, T, -1)
    for i in reversed(range(cfg["num_layers"])):
        prefix = CUDNN_CELL % i
        d, g = lstmp_bwd(saved["layers"][i], params[prefix + "/kernel"], None, None, None, None, np.ascontiguousarray(d))
        grads[prefix + "/kernel"], grads[prefix + "/bias"] = g["kernel"], g["bias"]
    return grads, d


def forward(params, cfg, x, seq_len, drop_seed=0):
    t = cfg.get("nnet_type", "blstm")
    if t == "blstm":
        return blstm_forward(params, cfg, x, seq_len, drop_seed)
    if t == "lstm":
        return lstm_forward(params, cfg, x, seq_len, drop_seed)
    if t == "cudnnlstm":
        return cudnnlstm_forward(params, cfg, x, seq_len, drop_seed)
    raise ValueError("unsupported nnet_type: %s" % t)


def backward(params, cfg, saved, dlogits):
    t = cfg.get("nnet_type", "blstm")
    fn = {"blstm": blstm_backward, "lstm": lstm_backward, "cudnnlstm": cudnnlstm_backward}[t]
    return fn(params, cfg, saved, dlogits)


def label_smoothing(logits, cfg, class_prior=None):
    """KL label-smoothing regulariser — nnet/bilstm.py:255-269 (sum over ALL [B,T,V], padded
    frames included).  Returns (loss, dlogits) or (None, None)."""
    u = cfg.get("uniform_label_sm")
    pw = cfg.get("prior_label_sm")
    V = logits.shape[-1]
    if u is not None and u > 0:
        w, logq = u, np.full(V, -math.log(V), logits.dtype)
    elif pw is not None and pw > 0 and class_prior is not None:
        w, logq = pw, np.asarray(class_prior, logits.dtype)
    else:
        return None, None
    mx = logits.max(axis=-1, keepdims=True)
    lp = logits - mx - np.log(np.exp(logits - mx).sum(axis=-1, keepdims=True))
    p = np.exp(lp)
    kl = p * (lp - logq)
    loss = w * kl.sum()
    s = kl.sum(axis=-1, keepdims=True)
    return loss, w * (kl - p * s)          # d/dx sum_k p_k (log p_k - log q_k)


def validation_graph(params, cfg, x, seq_len, dense_labels, drop_seed=0, want_grad=False, class_prior=None):
    """create_graph_for_validation_ctc — nnet/graph.py:51-162.  Returns dict with
    logits, size, eval_loss, loss, eval (+ dlogits/saved when want_grad)."""
    logits, saved = forward(params, cfg, x, seq_len, drop_seed)
    flat, offs = flatten_labels(dense_labels)
    tbv = np.ascontiguousarray(logits.transpose(1, 0, 2))                     # graph.py:72
    loss_b, grad, _ = ctc_loss(tbv, flat, offs, seq_len, want_grad)
    eval_loss = loss_b.sum()                                                  # graph.py:116
    loss = eval_loss
    dlogits = grad.transpose(1, 0, 2) if want_grad else None
    if cfg.get("nnet_type", "blstm") == "blstm":
        rl, rg = label_smoothing(logits, cfg, class_prior)
        if rl is not None:                                                    # graph.py:120-133
            loss = loss + rl
            if want_grad:
                dlogits = dlogits + rg
    tokens, out_len, _ = ctc_greedy(tbv, seq_len)
    dist = edit_distance(tokens, out_len, flat, offs)
    return dict(logits=logits, size=int(len(flat)), eval_loss=float(eval_loss), loss=float(loss),
                eval=float(dist.sum()), loss_per_utt=loss_b, tokens=tokens, token_len=out_len,
                dlogits=dlogits, saved=saved)


def _per_tensor(fn, keys):
    """Runs fn(key) for every parameter tensor on a thread pool (numpy releases the GIL inside large array
    operations); results come back in key order, so every reduction over tensors keeps its order."""
    keys = list(keys)
    nthr = min(len(keys), num_threads())
    if nthr <= 1:
        return [fn(k) for k in keys]
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(nthr) as ex:
        return list(ex.map(fn, keys))


def l2_and_clip(params, grads, clip_norm=5.0, l2=1e-5):
    """L2 on every trainable whose name lacks 'bias' + tf.clip_by_global_norm — graph.py:183-192."""
    g2 = {}

    def add_l2(k):
        g = grads[k]
        if "bias" in k:
            t = g.copy()
        else:
            t = params[k].astype(g.dtype) * l2          # g + l2 * p  (the sum is commutative: same bits)
            t += g
        g2[k] = t
        return float(np.square(t, dtype=np.float64).sum())

    norm = math.sqrt(sum(_per_tensor(add_l2, grads.keys())))
    scale = clip_norm / max(norm, clip_norm)

    def rescale(k):
        g2[k] *= g2[k].dtype.type(scale)

    _per_tensor(rescale, g2.keys())
    return g2, norm


def apply_optimizer(name, params, grads, state, lr):
    """tf.train.{Adam,GradientDescent,Momentum}Optimizer.apply_gradients — graph.py:37-48,197-200."""
    if name == "sgd":
        for k in grads:
            params[k] = (params[k] - lr * grads[k]).astype(params[k].dtype)
    elif name == "momentum":
        for k in grads:
            a = state.setdefault(k, np.zeros_like(params[k]))
            a[...] = 0.9 * a + grads[k]
            params[k] = (params[k] - lr * a).astype(params[k].dtype)
    elif name == "adam":
        t = state["__t"] = state.get("__t", 0) + 1
        b1, b2, eps = 0.9, 0.999, 1e-8
        lr_t = lr * math.sqrt(1 - b2 ** t) / (1 - b1 ** t)
        for k in grads:
            state.setdefault(k, [np.zeros_like(params[k]), np.zeros_like(params[k])])

        def adam(k):                       # same operations in the same order as the one-line formulas, in place
            m, v = state[k]
            g = grads[k]
            m *= b1                        # m = b1*m + (1-b1)*g
            m += (1 - b1) * g
            v *= b2                        # v = b2*v + (1-b2)*g^2
            g2 = g ** 2
            g2 *= (1 - b2)
            v += g2
            d = np.sqrt(v)                 # p = p - lr_t*m / (sqrt(v) + eps)
            d += eps
            u = lr_t * m
            u /= d
            params[k] = (params[k] - u).astype(params[k].dtype)

        _per_tensor(adam, grads.keys())
    else:
        raise ValueError(name)


def train_step(params, cfg, x, seq_len, dense_labels, opt_state, optimizer="adam", lr=1e-4,
               clip_norm=5.0, l2=1e-5, drop_seed=0):
    """One sess.run of create_graph_for_training_ctc — nnet/graph.py:165-209."""
    out = validation_graph(params, cfg, x, seq_len, dense_labels, drop_seed, want_grad=True)
    grads, _ = backward(params, cfg, out["saved"], np.ascontiguousarray(out["dlogits"]))
    clipped, norm = l2_and_clip(params, grads, clip_norm, l2)
    apply_optimizer(optimizer, params, clipped, opt_state, lr)
    bn_update_moving(params, out["saved"])
    out["grad_norm"] = norm
    out["grads"] = grads
    out.pop("saved")
    return out


class RunningStats:
    """The label-weighted running means of nnet.train / nnet.validate — nnet/funcs.py:36-65."""

    def __init__(self):
        self.step = 0
        self.processed = 0
        self.loss = 0.0
        self.acc = 0.0

    def update(self, size, eval_loss, batch_eval=None):
        if size > 0:
            self.processed += size
            bl = eval_loss / size
            self.loss += (bl - self.loss) * size / self.processed
            if batch_eval is not None:
                be = batch_eval / size
                self.acc += (be - self.acc) * size / self.processed
        self.step += 1
        return math.isnan(self.loss)


# ----------------------------------------------------------------------------- init helper
def init_params(cfg, seed=0, dtype=np.float32):
    """Random parameters with the reference's shapes, names and initialisers
    (Glorot-uniform LSTM kernels/peepholes/projection, zero biases,
    truncated-normal heads: bilstm.py:238-248, moe.py:33-57, lstm.py:332-342)."""
    rng = np.random.default_rng(seed)
    D = cfg["input_dim"] * (1 + cfg.get("left_context", 0) + cfg.get("right_context", 0))
    N, P, V = cfg["num_neurons"], cfg.get("num_projects"), cfg["num_targets"]
    Pout = P if P else N
    peep = bool(cfg.get("use_peepholes", False))
    blstm = cfg.get("nnet_type", "blstm") == "blstm"
    cudnn = cfg.get("nnet_type") == "cudnnlstm"
    if cudnn:
        P, Pout = None, N                                                 # the cell has no projection (lstm.py:73-76)
    params = {}

    def glorot(shape):
        fan_in, fan_out = (shape[0], shape[1]) if len(shape) == 2 else (shape[0], shape[0])
        lim = math.sqrt(6.0 / (fan_in + fan_out))
        return rng.uniform(-lim, lim, size=shape).astype(dtype)

    def trunc_normal(shape, std):
        a = rng.normal(0, std, size=shape)
        bad = np.abs(a) > 2 * std
        while bad.any():
            a[bad] = rng.normal(0, std, size=int(bad.sum()))
            bad = np.abs(a) > 2 * std
        return a.astype(dtype)

    def cell(prefix, I, peepholes):
        params[prefix + "/kernel"] = glorot((I + Pout, 4 * N))
        params[prefix + "/bias"] = np.zeros(4 * N, dtype)
        if peepholes:
            for nm in ("w_f_diag", "w_i_diag", "w_o_diag"):
                params[prefix + "/" + nm] = glorot((N,))
        if P:
            params[prefix + "/projection/kernel"] = glorot((N, P))

    for i in range(cfg["num_layers"]):
        if blstm:
            I = D if i == 0 else 2 * Pout
            cell("fd%d/frnn%d" % (i, i), I, peep)
            cell("bd%d/brnn%d" % (i, i), I, peep)
        elif cudnn:
            cell(CUDNN_CELL % i, D if i == 0 else N, False)
        else:
            cell("drnn%d/lstm_cell" % i, D if i == 0 else Pout, True)     # lstm.py:240 use_peepholes=True
    if not blstm and not cudnn:
        for j, name in enumerate(_bn_names(cfg)):                         # gamma 1, beta 0, moving mean 0 / var 1
            C = D if j == 0 else Pout
            params[name + "/gamma"], params[name + "/beta"] = np.ones(C, dtype), np.zeros(C, dtype)
            params[name + "/moving_mean"], params[name + "/moving_variance"] = np.zeros(C, dtype), np.ones(C, dtype)
    H = 2 * Pout if blstm else Pout
    E = (cfg.get("num_experts") or 0) if blstm else 0
    if E > 0:
        std = 1.0 / math.sqrt(H)
        params["Variable"] = trunc_normal((H, E), std)
        params["Variable_1"] = np.zeros(E, dtype)
        params["Variable_2"] = trunc_normal((H, E * V), std)
        params["Variable_3"] = np.zeros(E * V, dtype)
    else:
        std = 1.0 / math.sqrt(N if (blstm or cudnn) else H)               # bilstm.py:239, lstm.py:105 use num_neurons
        params["Variable"] = trunc_normal((H, V), std)
        params["Variable_1"] = np.zeros(V, dtype)
    return params
