"""GPU parity tests, kernel level: every C-ABI entry point of liblstm_ctc_hip.so against the CPU oracle
on the same seeded inputs.  Integer results (greedy tokens, edit distance) must be bit-exact; floating
point within the tolerances written below (north star: loss <= 1e-4 relative)."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from lstm_ctc_amd import ops as o, _lib
    _lib.load()
    return o


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


# ------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("ta,tb", [(0, 0), (1, 0), (0, 1), (1, 1)])
@pytest.mark.parametrize("M,N,K", [(128, 128, 16), (257, 131, 70), (64, 44, 2048), (1000, 1280, 40), (5, 3, 7),
                                   (300, 200, 64), (256, 200, 32), (300, 256, 48), (200, 300, 8192)])   # peeled edges
def test_gemm(ops, ta, tb, M, N, K):
    rng = np.random.default_rng(M * 7 + N * 3 + K + ta * 2 + tb)
    A = rng.normal(size=(K, M) if ta else (M, K)).astype(np.float32)
    B = rng.normal(size=(N, K) if tb else (K, N)).astype(np.float32)
    bias = rng.normal(size=N).astype(np.float32)
    C0 = rng.normal(size=(M, N)).astype(np.float32)
    ref = 0.5 * ((A.T if ta else A).astype(np.float64) @ (B.T if tb else B).astype(np.float64)) + 2.0 * C0 + bias
    out = dev(C0)
    ops.gemm(dev(A), dev(B), ta=bool(ta), tb=bool(tb), out=out, alpha=0.5, beta=2.0, bias=dev(bias))
    err = np.abs(out.cpu().numpy() - ref).max()
    assert err < 2e-6 * K * 4 + 1e-5, err


@pytest.mark.parametrize("ta,tb", [(0, 0), (1, 0), (0, 1), (1, 1)])
@pytest.mark.parametrize("M,N,K", [(256, 256, 32), (256, 512, 64), (512, 768, 192), (768, 256, 4160), (1024, 1024, 128),
                                   (256, 256, 8192),
                                   # ragged edges: interior on the 256 x 256 kernel, right / bottom strips on the 128 x 128 one
                                   (300, 520, 64), (513, 256, 96), (704, 300, 160)])
def test_gemm_f32_lds_dma_tiles(ops, ta, tb, M, N, K, monkeypatch):
    """The 256 x 256 x 32 LDS-DMA kernel (gemm_f32g_kernel) forced onto small eligible shapes: one and several k tiles, odd
    and even tile counts, split K, all four operand forms (row-form operands swizzled on the source side, k-major operands
    staged linearly), alpha / beta / bias; against float64, and bit-compared with nothing - the k walk is permuted."""
    monkeypatch.setenv("LC_GEMM_F32_BIG", "2")
    rng = np.random.default_rng(M * 5 + N * 3 + K + ta * 2 + tb)
    A = rng.normal(size=(K, M) if ta else (M, K)).astype(np.float32)
    B = rng.normal(size=(N, K) if tb else (K, N)).astype(np.float32)
    bias = rng.normal(size=N).astype(np.float32)
    C0 = rng.normal(size=(M, N)).astype(np.float32)
    ref = 0.5 * ((A.T if ta else A).astype(np.float64) @ (B.T if tb else B).astype(np.float64)) + 2.0 * C0 + bias
    out = dev(C0)
    ops.gemm(dev(A), dev(B), ta=bool(ta), tb=bool(tb), out=out, alpha=0.5, beta=2.0, bias=dev(bias))
    err = np.abs(out.cpu().numpy() - ref).max()
    assert err < 2e-6 * K * 4 + 1e-5, err
    monkeypatch.setenv("LC_GEMM_F32_BIG", "0")                     # and the 128 x 128 kernel on the same inputs
    out2 = dev(C0)
    ops.gemm(dev(A), dev(B), ta=bool(ta), tb=bool(tb), out=out2, alpha=0.5, beta=2.0, bias=dev(bias))
    assert np.abs(out.cpu().numpy() - out2.cpu().numpy()).max() < 2e-6 * K * 4 + 1e-5


@pytest.mark.parametrize("ta,tb", [(0, 0), (1, 0), (0, 1)])
def test_gemm_f32_lds_dma_tiles_on_views(ops, ta, tb, monkeypatch):
    """The 256 x 256 kernel on the operand shapes the model hands it: column windows of wider matrices (leading dimension >
    extent, 16-byte-aligned offsets), output written into a column window of a wider buffer with beta = 1."""
    monkeypatch.setenv("LC_GEMM_F32_BIG", "2")
    M, N, K = 512, 256, 96
    rng = np.random.default_rng(17 + 2 * ta + tb)
    Abig = rng.normal(size=(K, M + 64) if ta else (M, K + 32)).astype(np.float32)
    Bbig = rng.normal(size=(N, K + 64) if tb else (K, N + 128)).astype(np.float32)
    Cbig = rng.normal(size=(M, N + 256)).astype(np.float32)
    A = Abig[:, 32:32 + M] if ta else Abig[:, 16:16 + K]
    B = Bbig[:, 64:64 + K] if tb else Bbig[:, 128:128 + N]
    ref = Cbig.astype(np.float64).copy()
    ref[:, 256:] += (A.T if ta else A).astype(np.float64) @ (B.T if tb else B).astype(np.float64)
    a_d, b_d, c_d = dev(Abig), dev(Bbig), dev(Cbig)
    a_v = a_d[:, 32:32 + M] if ta else a_d[:, 16:16 + K]
    b_v = b_d[:, 64:64 + K] if tb else b_d[:, 128:128 + N]
    ops.gemm(a_v, b_v, ta=bool(ta), tb=bool(tb), out=c_d[:, 256:], beta=1.0)
    got = c_d.cpu().numpy()
    assert np.array_equal(got[:, :256], Cbig[:, :256])               # nothing written outside the window
    assert np.abs(got - ref).max() < 2e-6 * K * 4 + 1e-5


@pytest.mark.parametrize("ta,tb", [(0, 0), (1, 0), (0, 1), (1, 1)])
@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (256, 384, 160), (257, 131, 70), (64, 44, 2048), (1000, 1280, 40),
                                   (5, 3, 7), (256, 256, 8192), (300, 200, 64), (256, 200, 32), (200, 300, 8192)])
def test_gemm_bf16(ops, oracle, ta, tb, M, N, K):
    """c5 operand mode: C = alpha * bf16(A).bf16(B) + beta*C + bias with fp32 accumulation; checked against the
    float64 product of the SAME rounded operands (so only the accumulation order differs)."""
    rng = np.random.default_rng(M * 5 + N * 3 + K + ta * 2 + tb)
    A = rng.normal(size=(K, M) if ta else (M, K)).astype(np.float32)
    B = rng.normal(size=(N, K) if tb else (K, N)).astype(np.float32)
    bias = rng.normal(size=N).astype(np.float32)
    C0 = rng.normal(size=(M, N)).astype(np.float32)
    Ar, Br = oracle.bf16_round(A), oracle.bf16_round(B)
    assert np.abs(Ar - A).max() <= np.abs(A).max() * 2.0 ** -8
    ref = 0.5 * ((Ar.T if ta else Ar).astype(np.float64) @ (Br.T if tb else Br).astype(np.float64)) + 2.0 * C0 + bias
    out = dev(C0)
    ops.gemm(dev(A), dev(B), ta=bool(ta), tb=bool(tb), out=out, alpha=0.5, beta=2.0, bias=dev(bias), bf16=True)
    err = np.abs(out.cpu().numpy() - ref).max()
    assert err < 2e-6 * K * 4 + 1e-5, err


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 192), (300, 200, 128), (257, 131, 72), (64, 44, 2048),
                                   (1000, 1280, 40), (5, 3, 8), (200, 300, 8192), (256, 256, 8192),
                                   # whole 256 x 256 tiles: the LDS-DMA kernel (one k tile, odd / even tile counts, split K)
                                   (256, 512, 64), (512, 768, 192), (768, 256, 4160), (1024, 1024, 128),
                                   # ragged edges around an interior of whole 256 x 256 tiles
                                   (300, 520, 64), (513, 256, 128), (704, 300, 192)])
def test_gemm_bf16_shadow_operands(ops, oracle, M, N, K):
    """lc_cast_bf16 + lc_gemm_bf16_nt: bf16 shadows (natural and transposed) of fp32 tensors, product in NT form.
    The shadows must be exactly the RNE rounding; the product is checked against float64 on the rounded operands;
    and it must agree with lc_gemm_bf16 (which rounds the same operands in its loader) to accumulation-order level."""
    rng = np.random.default_rng(M + 3 * N + K)
    A = rng.normal(size=(M, K)).astype(np.float32)            # A[M,K]
    Bt = rng.normal(size=(K, N)).astype(np.float32)           # stored [K,N]: its TRANSPOSED shadow is B[N,K]
    bias = rng.normal(size=N).astype(np.float32)
    C0 = rng.normal(size=(M, N)).astype(np.float32)
    a_nat, _ = ops.cast_bf16(dev(A), nat=True, tr=False)
    b_nat, b_tr = ops.cast_bf16(dev(Bt), nat=True, tr=True)
    Ar, Br = oracle.bf16_round(A), oracle.bf16_round(Bt)
    assert np.array_equal(a_nat.float().cpu().numpy(), Ar)
    assert np.array_equal(b_nat.float().cpu().numpy(), Br)
    assert np.array_equal(b_tr.float().cpu().numpy()[:, :K], Br.T)
    ref = 0.5 * (Ar.astype(np.float64) @ Br.astype(np.float64)) + 2.0 * C0 + bias
    out = dev(C0)
    ops.gemm_bf16_nt(a_nat, b_tr, out=out, alpha=0.5, beta=2.0, bias=dev(bias), K=K)
    err = np.abs(out.cpu().numpy() - ref).max()
    assert err < 2e-6 * K * 4 + 1e-5, err
    out2 = dev(C0)
    ops.gemm(dev(A), dev(Bt), out=out2, alpha=0.5, beta=2.0, bias=dev(bias), bf16=True)
    assert np.abs(out.cpu().numpy() - out2.cpu().numpy()).max() < 2e-6 * K * 4 + 1e-5


@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (256, 512, 16), (512, 256, 1), (768, 512, 200), (256, 1024, 4100),
                                   (1024, 256, 8192), (2048, 4096, 1000)])
def test_gemm_bf16_tn_on_natural_shadows(ops, oracle, M, N, K):
    """lc_gemm_bf16_tn: C = alpha A^T B + beta C + bias with both bf16 operands K-MAJOR (A [K,M], B [K,N]) - the weight
    gradients X^T dZ on the natural shadows, transposing LDS reads (ds_read_b64_tr_b16) instead of transposed copies.
    Non-symmetric random operands (a transposed or permuted fragment cannot pass), ragged K (zero-filled tails, also
    inside a split-K chunk), against float64 on the same rounded operands, and against the NT kernel on transposed shadows."""
    rng = np.random.default_rng(7 * M + 3 * N + K)
    A = rng.normal(size=(K, M)).astype(np.float32)
    B = rng.normal(size=(K, N)).astype(np.float32)
    bias = rng.normal(size=N).astype(np.float32)
    C0 = rng.normal(size=(M, N)).astype(np.float32)
    a_nat, a_tr = ops.cast_bf16(dev(A), nat=True, tr=True)
    b_nat, b_tr = ops.cast_bf16(dev(B), nat=True, tr=True)
    Ar, Br = oracle.bf16_round(A), oracle.bf16_round(B)
    ref = 0.5 * (Ar.astype(np.float64).T @ Br.astype(np.float64)) + 2.0 * C0 + bias
    out = dev(C0)
    ops.gemm_bf16_tn(a_nat, b_nat, out=out, alpha=0.5, beta=2.0, bias=dev(bias))
    got = out.cpu().numpy()
    assert np.abs(got - ref).max() < 2e-6 * K * 4 + 1e-5
    if K % 8 == 0:
        out2 = dev(C0)
        ops.gemm_bf16_nt(a_tr, b_tr, out=out2, alpha=0.5, beta=2.0, bias=dev(bias), K=K)
        assert np.abs(got - out2.cpu().numpy()).max() < 2e-6 * K * 4 + 1e-5


@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (512, 256, 128), (256, 768, 2048), (1024, 512, 8192), (768, 1024, 320)])
def test_gemm_bf16_nn_on_natural_shadows(ops, oracle, M, N, K):
    """lc_gemm_bf16_nn: A k-contiguous [M,K], B K-major [K,N] - the forward products on the natural shadows of activation and
    weight (A's fragments by ds_read_b128, B's by transposing reads).  Against float64 on the rounded operands and against
    the NT kernel on B's transposed shadow; strided windows of wider buffers; alpha / beta / bias."""
    rng = np.random.default_rng(5 * M + N + 3 * K)
    A = rng.normal(size=(M, K + 64)).astype(np.float32)
    B = rng.normal(size=(K, N + 256)).astype(np.float32)
    bias = rng.normal(size=N).astype(np.float32)
    C0 = rng.normal(size=(M, N)).astype(np.float32)
    a_nat, _ = ops.cast_bf16(dev(A), nat=True, tr=False)
    b_nat, _ = ops.cast_bf16(dev(B), nat=True, tr=False)
    Ar, Br = oracle.bf16_round(A)[:, 64:], oracle.bf16_round(B)[:, 256:]
    ref = 0.5 * (Ar.astype(np.float64) @ Br.astype(np.float64)) + 2.0 * C0 + bias
    out = dev(C0)
    ops.gemm_bf16_nn(a_nat[:, 64:], b_nat[:, 256:], out=out, alpha=0.5, beta=2.0, bias=dev(bias))
    got = out.cpu().numpy()
    assert np.abs(got - ref).max() < 2e-6 * K * 4 + 1e-5
    _, b_tr = ops.cast_bf16(dev(np.ascontiguousarray(B[:, 256:])), nat=False, tr=True)
    out2 = dev(C0)
    ops.gemm_bf16_nt(a_nat[:, 64:], b_tr, out=out2, alpha=0.5, beta=2.0, bias=dev(bias), K=K)
    assert np.abs(got - out2.cpu().numpy()).max() < 2e-6 * K * 4 + 1e-5
    with pytest.raises(Exception):
        ops.gemm_bf16_nn(a_nat[:, 64:64 + K - 8], b_nat[:K - 8, 256:])    # K not a multiple of 64: refused


def test_gemm_bf16_tn_on_row_windows(ops, oracle):
    """dR = hs_prev^T dZ: both operands are ROW windows of wider natural shadows, one step (B rows) apart, beta = 1."""
    M, N, K, shift = 256, 512, 1000, 64
    rng = np.random.default_rng(29)
    A = rng.normal(size=(K + shift, M + 256)).astype(np.float32)
    B = rng.normal(size=(K + shift, N)).astype(np.float32)
    C0 = rng.normal(size=(M, N)).astype(np.float32)
    a_nat, _ = ops.cast_bf16(dev(A), nat=True, tr=False)
    b_nat, _ = ops.cast_bf16(dev(B), nat=True, tr=False)
    Ar, Br = oracle.bf16_round(A), oracle.bf16_round(B)
    ref = C0 + Ar[:K, 256:256 + M].astype(np.float64).T @ Br[shift:shift + K].astype(np.float64)
    out = dev(C0)
    ops.gemm_bf16_tn(a_nat[:K, 256:256 + M], b_nat[shift:shift + K], out=out, beta=1.0)
    assert np.abs(out.cpu().numpy() - ref).max() < 2e-6 * K * 4 + 1e-5
    with pytest.raises(Exception):
        ops.gemm_bf16_tn(a_nat[:K, :300], b_nat[:K], out=None)          # M not a multiple of 256: refused, not mangled


def test_gemm_bf16_lds_dma_tiles_on_views(ops, oracle):
    """lc_gemm_bf16_nt's 256 x 256 kernel on column windows of wider shadows (what dR = hs^T dz reads: the transposed
    shadows of the whole tensors, shifted by one step of B columns against each other) and with beta = 1."""
    M, N, K, shift = 256, 512, 192, 64
    rng = np.random.default_rng(23)
    A = rng.normal(size=(M, K + shift)).astype(np.float32)
    B = rng.normal(size=(N, K + shift)).astype(np.float32)
    C0 = rng.normal(size=(M, N)).astype(np.float32)
    a_nat, _ = ops.cast_bf16(dev(A), nat=True, tr=False)
    b_nat, _ = ops.cast_bf16(dev(B), nat=True, tr=False)
    Ar, Br = oracle.bf16_round(A), oracle.bf16_round(B)
    ref = C0 + Ar[:, :K].astype(np.float64) @ Br[:, shift:shift + K].astype(np.float64).T
    out = dev(C0)
    ops.gemm_bf16_nt(a_nat[:, :K], b_nat[:, shift:shift + K], out=out, beta=1.0, K=K)
    assert np.abs(out.cpu().numpy() - ref).max() < 2e-6 * K * 4 + 1e-5


@pytest.mark.parametrize("rows,C,ld,off", [(192, 128, 256, 64), (64, 64, 64, 0), (128, 320, 320, 0), (256, 192, 1024, 512),
                                            (100, 128, 128, 0), (128, 72, 72, 0), (64, 64, 68, 4), (64, 64, 66, 2)])
@pytest.mark.parametrize("nat,tr", [(True, True), (True, False), (False, True)])
def test_cast_bf16_exact(ops, oracle, rows, C, ld, off, nat, tr):
    """lc_cast_bf16 on whole 64 x 64 tiles (the 16-byte kernel), on ragged shapes and on unaligned windows (the
    element-wise kernel): column windows of a wider matrix, every combination of outputs; both copies must be exactly
    the round-to-nearest-even bf16 of the input, the transposed one with its zero pad columns."""
    rng = np.random.default_rng(rows * 7 + C + ld + off)
    big = rng.normal(size=(rows, ld)).astype(np.float32)
    big[0, off] = 1.0 + 2.0 ** -8                                   # a tie: rounds to even (1.0)
    big[1, off] = 1.0 + 3 * 2.0 ** -8                               # a tie: rounds to even (1.0 + 2^-6)
    x = dev(big)[:, off:off + C]
    n, t = ops.cast_bf16(x, nat=nat, tr=tr)
    ref = oracle.bf16_round(big[:, off:off + C])
    if nat:
        assert np.array_equal(n.float().cpu().numpy(), ref)
    else:
        assert n is None
    if tr:
        got = t.float().cpu().numpy()
        assert got.shape == (C, (rows + 7) // 8 * 8)
        assert np.array_equal(got[:, :rows], ref.T) and not got[:, rows:].any()
    else:
        assert t is None


@pytest.mark.parametrize("name,ta,tb,M,N,K", [("zx", 0, 0, 64000, 4096, 2048), ("dX", 0, 1, 64000, 2048, 4096),
                                              ("dKx", 1, 0, 2048, 4096, 64000), ("dR", 1, 0, 1024, 4096, 63936),
                                              ("proj", 0, 0, 64000, 1024, 1024), ("head", 0, 0, 64000, 44, 2048)])
def test_gemm_full_size_c4_shapes(ops, name, ta, tb, M, N, K):
    """The c4 products at full size (fast path, split-K, peeled edges) against torch.mm (rocBLAS / hipBLASLt fp32) on
    the same inputs: two fp32 accumulations of K terms in different orders (this kernel accumulates a whole K = 64000
    column sequentially in the MFMA accumulator: rms error 1.1e-3 against float64, torch 0.7e-3; the tail is on the
    largest |values|), hence a mixed tolerance; and, in bf16 mode, the shadow-operand route against torch.mm on the
    same bf16-rounded values."""
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    A = torch.randn((K, M) if ta else (M, K), device="cuda", generator=g)
    B = torch.randn((N, K) if tb else (K, N), device="cuda", generator=g)
    At, Bt = (A.t() if ta else A), (B.t() if tb else B)
    ref = torch.mm(At, Bt)
    out = ops.gemm(A, B, ta=bool(ta), tb=bool(tb))
    scale = float(K) ** 0.5                                   # entries are sums of K unit-variance products
    assert torch.allclose(out, ref, rtol=5e-5, atol=5e-5 * scale), (name, float((out - ref).abs().max()))
    a_s = ops.cast_bf16(A, nat=not ta, tr=bool(ta))
    b_s = ops.cast_bf16(B, nat=bool(tb), tr=not tb)
    As, Bs = (a_s[1] if ta else a_s[0]), (b_s[0] if tb else b_s[1])
    out16 = ops.gemm_bf16_nt(As, Bs, K=K)
    ref16 = torch.mm(At.to(torch.bfloat16).float(), Bt.to(torch.bfloat16).float())
    assert torch.allclose(out16, ref16, rtol=5e-5, atol=5e-5 * scale), (name, float((out16 - ref16).abs().max()))


def test_gemm_strided_views(ops):
    """Column-slice outputs / inputs (the concat buffer halves) and 4-byte-aligned-only pointers."""
    rng = np.random.default_rng(1)
    A = rng.normal(size=(300, 96)).astype(np.float32)
    W = rng.normal(size=(48, 50)).astype(np.float32)
    Y = torch.zeros((300, 100), device="cuda")
    a = dev(A)
    ops.gemm(a[:, 48:], dev(W), out=Y[:, 50:])
    ops.gemm(a[:, :48], dev(W), out=Y[:, :50])
    ref = np.concatenate([A[:, :48] @ W, A[:, 48:] @ W], axis=1)
    np.testing.assert_allclose(Y.cpu().numpy(), ref, atol=1e-4)
    odd = a[1:, 1:45]                      # pointer only 4-byte aligned, row stride 96
    got = ops.gemm(odd, dev(W[:44]))
    np.testing.assert_allclose(got.cpu().numpy(), A[1:, 1:45] @ W[:44], atol=1e-4)


# ------------------------------------------------------------------------------------------ CTC
@pytest.fixture(params=["by_batch_size", "in_phase_2"])
def ctc_frame_stats(request, ops):
    """Where the two-launch CTC takes the per-frame log-sum-exp: the library's choice (phase 1 below 512 utterances), or
    forced into phase 2's frame waves - what calls with >= 512 utterances and a gradient do (lc_set_option "ctc_lse2":
    no frame statistics in phase 1, the loss folded by two commuting float atomic adds)."""
    if request.param == "in_phase_2":
        ops.set_option("ctc_lse2", 1)
    yield request.param
    ops.set_option("ctc_lse2", None)


def _ragged(rng, B, T, V, Lmin, Lmax):
    seq_len = np.sort(rng.integers(max(2, int(T * 0.8)), T + 1, size=B)).astype(np.int32)
    seq_len[-1] = T
    labels = []
    for b in range(B):
        L = int(rng.integers(Lmin, min(Lmax, seq_len[b] // 2) + 1))
        lab = rng.integers(0, V - 1, size=L)
        if b % 3 == 0 and L >= 3:
            lab[1] = lab[0]
            lab[-1] = lab[-2]
        labels.append(lab)
    flat = np.concatenate(labels).astype(np.int32)
    offs = np.concatenate([[0], np.cumsum([len(l) for l in labels])]).astype(np.int32)
    return seq_len, flat, offs, max(len(l) for l in labels)


def test_ctc_tf_known_answers(ops):
    kat = json.load(open(os.path.join(GOLD, "ctc_tf_known_answers.json")))
    T, V, B = kat["T"], kat["V"], len(kat["utts"])
    logits = np.zeros((T, B, V), np.float32)
    flat, offs = [], [0]
    for b, u in enumerate(kat["utts"]):
        logits[:, b] = np.log(np.asarray(u["probs"]))
        flat += u["labels"]
        offs.append(len(flat))
    loss, grad = ops.ctc_loss(dev(logits), dev(np.array(flat, np.int32)), dev(np.array(offs, np.int32)),
                              dev(np.array([T] * B, np.int32)), 5)
    for b, u in enumerate(kat["utts"]):
        assert abs(loss[b].item() - u["loss"]) / u["loss"] < 1e-5
        assert np.abs(grad[:, b].cpu().numpy() - np.asarray(u["grad"])).max() < 3e-6      # TF's gradient_log_prob_b


def test_greedy_tf_known_answers(ops):
    kat = json.load(open(os.path.join(GOLD, "ctc_tf_greedy_known_answers.json")))
    with np.errstate(divide="ignore"):
        logits = np.stack([np.log(np.asarray(u["probs"], np.float32)) for u in kat["utts"]], axis=1)   # -inf logits
    tok, n = ops.ctc_greedy(dev(logits), dev(np.array([u["seq_len"] for u in kat["utts"]], np.int32)))
    tok, n = tok.cpu().numpy(), n.cpu().numpy()
    for b, u in enumerate(kat["utts"]):
        assert list(tok[b, :n[b]]) == u["decoded"]


@pytest.mark.parametrize("T,B,V,Lmin,Lmax", [
    (30, 5, 6, 1, 8),          # PPL 1
    (120, 9, 44, 10, 50),      # PPL 2
    (300, 8, 72, 60, 120),     # PPL 4
    (600, 4, 44, 150, 250),    # PPL 8
    (1100, 3, 30, 300, 500),   # PPL 16
    (1400, 2, 20, 520, 560),   # PPL 32
    # more than 128 utterances: the many-utterance geometries of the two-launch path (B <= 128 with more than 256 lattice
    # positions, and every lattice beyond 1024 positions, take the three-kernel path: the cases above from PPL 8 on)
    (60, 130, 9, 3, 14),       # one scan wave, 4 positions per lane (the B = 512 bench geometry)
    (300, 131, 44, 90, 126),   # the same at its widest (253 positions)
    (340, 129, 20, 130, 160),  # four scan waves x 2 positions per lane, per-class LDS atomics
    (700, 129, 12, 270, 300),  # four scan waves x 4 positions per lane
])
def test_ctc_vs_oracle(ops, oracle, T, B, V, Lmin, Lmax, ctc_frame_stats):
    rng = np.random.default_rng(T + B)
    seq_len, flat, offs, maxL = _ragged(rng, B, T, V, Lmin, Lmax)
    logits = rng.normal(0, 1.5, size=(T, B, V)).astype(np.float32)
    ref_loss, ref_grad, bad = oracle.ctc_loss(logits.astype(np.float64), flat, offs, seq_len)
    assert bad == 0
    loss, grad = ops.ctc_loss(dev(logits), dev(flat), dev(offs), dev(seq_len), maxL)
    loss, grad = loss.cpu().numpy(), grad.cpu().numpy()
    np.testing.assert_allclose(loss, ref_loss, rtol=1e-5)                  # north star: <= 1e-4 relative
    # gradient entries are probabilities in [-1,1]; fp32 log-space lattice => absolute tolerance
    assert np.abs(grad - ref_grad).max() < 2e-3 * max(1.0, T / 300)
    assert np.abs(grad - ref_grad).mean() < 2e-5
    for b in range(B):
        assert np.all(grad[seq_len[b]:, b] == 0)
    # vs the float32 oracle (the TF-like arithmetic) the two fp32 results must be equally close to fp64
    f32_loss, f32_grad, _ = oracle.ctc_loss(logits, flat, offs, seq_len)
    assert np.abs(loss - ref_loss).max() <= 4 * np.abs(f32_loss - ref_loss).max() + 1e-3


def test_ctc_loss_precision_at_bench_length_both_variants(ops, oracle):
    """with the frame statistics in phase 2 (the default from 512 utterances) the two workgroups of an
    utterance used to round their halves of sum_t lse_t (thousands at T = 1000) to fp32 before the atomic add - 2.4e-4
    absolute on a loss of a few hundred, batch-size dependent precision.  Each now subtracts half of ln p~ in double first, so
    both variants must land within 1.5 fp32 ulps of the float64 loss (T = 1000, V = 44, L = 100: the bench shape; before the
    change the phase-2 variant carried three roundings at the magnitude of the halves)."""
    rng = np.random.default_rng(5)
    T, B, V, L = 1000, 6, 44, 100
    logits = rng.normal(0, 1.0, size=(T, B, V)).astype(np.float32)
    flat = rng.integers(0, V - 1, size=B * L).astype(np.int32)
    offs = (np.arange(B + 1) * L).astype(np.int32)
    seq_len = np.full(B, T, np.int32)
    ref_loss, _, bad = oracle.ctc_loss(logits.astype(np.float64), flat, offs, seq_len)
    assert bad == 0 and ref_loss.min() > 100
    errs = {}
    for variant, opt in (("phase_1", 0), ("phase_2", 1)):
        ops.set_option("ctc_lse2", opt)
        try:
            loss, _ = ops.ctc_loss(dev(logits), dev(flat), dev(offs), dev(seq_len), L)
        finally:
            ops.set_option("ctc_lse2", None)
        errs[variant] = float(np.abs(loss.cpu().numpy().astype(np.float64) - ref_loss).max())
    ulp = float(np.spacing(np.float32(ref_loss.max())))
    # random logits: the loss is in the thousands here, so one fp32 ulp of it is 2.4e-4; both variants must be within ~1 ulp
    # (the final rounding + the fp32 log-space lattice), the phase-2 variant no further out than the phase-1 one by another
    assert errs["phase_1"] <= 1.5 * ulp and errs["phase_2"] <= 1.5 * ulp, (errs, ulp)


def test_ctc_wide_alphabet_legacy_path(ops, oracle):
    """V > 128 takes the three-kernel path (row statistics, alpha / beta scan with both lattices stored, gradient pass);
    the meet-in-the-middle path covers the alphabets of the recipes (V <= 128)."""
    rng = np.random.default_rng(77)
    T, B, V = 90, 5, 200
    seq_len, flat, offs, maxL = _ragged(rng, B, T, V, 5, 30)
    logits = rng.normal(0, 1.5, size=(T, B, V)).astype(np.float32)
    ref_loss, ref_grad, _ = oracle.ctc_loss(logits.astype(np.float64), flat, offs, seq_len)
    loss, grad = ops.ctc_loss(dev(logits), dev(flat), dev(offs), dev(seq_len), maxL)
    np.testing.assert_allclose(loss.cpu().numpy(), ref_loss, rtol=1e-5)
    assert np.abs(grad.cpu().numpy() - ref_grad).max() < 2e-3


def test_ctc_edge_cases(ops, oracle, ctc_frame_stats):
    """L > T (skipped: loss 0 / grad 0), infeasible repeats (loss inf, grad = softmax), L = 1, T = L."""
    V, T = 7, 12
    rng = np.random.default_rng(5)
    logits = rng.normal(size=(T, 5, V)).astype(np.float32)
    labels = [list(range(6)) * 3, [1, 1, 1], [2], [0, 1, 2, 3], [3, 3]]
    seq_len = np.array([12, 4, 12, 4, 3], np.int32)       # utt0 L=18>12 skip; utt1 needs 5 frames, has 4; utt3 T==L
    flat = np.concatenate(labels).astype(np.int32)
    offs = np.concatenate([[0], np.cumsum([len(l) for l in labels])]).astype(np.int32)
    ref_loss, ref_grad, bad = oracle.ctc_loss(logits, flat, offs, seq_len)
    loss, grad = ops.ctc_loss(dev(logits), dev(flat), dev(offs), dev(seq_len), 18)
    loss, grad = loss.cpu().numpy(), grad.cpu().numpy()
    assert loss[0] == 0 and np.all(grad[:, 0] == 0)
    assert np.isinf(loss[1]) and np.isinf(ref_loss[1]) and bad == 1
    np.testing.assert_allclose(grad[:, 1], ref_grad[:, 1], atol=1e-6)
    np.testing.assert_allclose(loss[2:], ref_loss[2:], rtol=1e-5)
    np.testing.assert_allclose(grad[:, 2:], ref_grad[:, 2:], atol=1e-5)


def test_ctc_full_size_properties(ops):
    """BASELINE sizes (T=1000, B=64, V=44, L=100): size-independent properties — gradient rows sum
    to 0, blank+label posteriors in [0,1], loss finite and positive, grad 0 beyond seq_len."""
    rng = np.random.default_rng(9)
    T, B, V, L = 1000, 64, 44, 100
    logits = dev(rng.normal(size=(T, B, V)).astype(np.float32))
    flat = dev(rng.integers(0, V - 1, size=B * L).astype(np.int32))
    offs = dev((np.arange(B + 1) * L).astype(np.int32))
    seq_len = np.full(B, T, np.int32)
    seq_len[:8] = 700
    loss, grad = ops.ctc_loss(logits, flat, offs, dev(seq_len), L)
    assert torch.isfinite(loss).all() and (loss > 0).all()
    assert grad.sum(dim=2).abs().max().item() < 1e-3
    post = torch.softmax(logits, 2) - grad
    assert post.min().item() > -1e-3 and post.max().item() < 1 + 1e-3
    assert grad[700:, :8].abs().max().item() == 0


# ------------------------------------------------------------------------------------------ greedy / edit distance
def test_greedy_bit_exact(ops, oracle):
    rng = np.random.default_rng(11)
    T, B, V = 333, 7, 44
    logits = rng.normal(size=(T, B, V)).astype(np.float32)
    logits[:, :, V - 1] += 2.0                                  # blank-heavy like a CTC model
    logits[5:40, 2, :] = 0.25                                   # exact ties -> lowest index
    logits[rng.integers(0, T, 50), rng.integers(0, B, 50), :] = np.float32(1.0)
    seq_len = np.array([333, 300, 280, 1, 2, 200, 333], np.int32)
    rt, rn, _ = oracle.ctc_greedy(logits, seq_len)
    tok, n = ops.ctc_greedy(dev(logits), dev(seq_len))
    tok, n = tok.cpu().numpy(), n.cpu().numpy()
    assert np.array_equal(n, rn)
    for b in range(B):
        assert np.array_equal(tok[b, :n[b]], rt[b, :rn[b]])
    truth = [rng.integers(0, V - 1, size=rng.integers(0, 60)) for _ in range(B)]
    flat = np.concatenate(truth).astype(np.int32)
    offs = np.concatenate([[0], np.cumsum([len(t) for t in truth])]).astype(np.int32)
    assert np.array_equal(ops.edit_distance_host(tok, n, flat, offs), oracle.edit_distance(rt, rn, flat, offs))


# ------------------------------------------------------------------------------------------ MoE combine, misc
@pytest.mark.parametrize("keep", [1.0, 0.8])
def test_moe_combine(ops, oracle, keep):
    rng = np.random.default_rng(13)
    T, B, H, E, V = 6, 5, 24, 7, 11
    R = T * B
    h = rng.normal(size=(R, H)).astype(np.float32)
    Wp, bp = rng.normal(0, .3, (H, E)).astype(np.float32), rng.normal(0, .1, E).astype(np.float32)
    W, b = rng.normal(0, .3, (H, E * V)).astype(np.float32), rng.normal(0, .1, E * V).astype(np.float32)
    dy = rng.normal(size=(R, V)).astype(np.float32)
    dpi = oracle.dropout_mask(3, 1000, (T, B, E), keep).reshape(R, E) if keep < 1 else None
    dz = oracle.dropout_mask(3, 1001, (T, B, E * V), keep).reshape(R, E * V) if keep < 1 else None
    y_ref, sv = oracle.moe_fwd(h.astype(np.float64), Wp, bp, W, b, 10.0, dpi, dz)
    dh_ref, g_ref = oracle.moe_bwd(sv, Wp.astype(np.float64), W.astype(np.float64), 10.0, dy.astype(np.float64))
    a = ops.gemm(dev(h), dev(Wp), bias=dev(bp))
    q = ops.gemm(dev(h), dev(W), bias=dev(b))
    logits, pi = ops.moe_combine_fwd(a, q, E, V, 10.0, keep, 3)
    np.testing.assert_allclose(logits.cpu().numpy(), y_ref, atol=2e-4)
    da = ops.moe_combine_bwd(pi, q, dev(dy), E, V, 10.0, keep, 3)
    dh = ops.gemm(da, dev(Wp), tb=True)
    ops.gemm(q, dev(W), tb=True, out=dh, beta=1.0)
    np.testing.assert_allclose(dh.cpu().numpy(), dh_ref, atol=2e-3, rtol=1e-3)
    np.testing.assert_allclose(ops.gemm(dev(h), da, ta=True).cpu().numpy(), g_ref["Wp"], atol=2e-3, rtol=1e-3)
    np.testing.assert_allclose(ops.colsum(q).cpu().numpy(), g_ref["b"], atol=2e-3, rtol=1e-3)


def test_dropout_mask_matches_oracle(ops, oracle):
    T, B, P = 7, 3, 10
    x = torch.ones((T * B, 2 * P), device="cuda")
    ops.dropout_scale(x[:, P:], 0.75, 1234, 5)
    m = oracle.dropout_mask(1234, 5, (T, B, P), 0.75).reshape(T * B, P)
    assert np.array_equal(x[:, P:].cpu().numpy(), m)
    assert np.all(x[:, :P].cpu().numpy() == 1)
    frac = (m > 0).mean()
    assert 0.6 < frac < 0.9


@pytest.mark.parametrize("rows,P", [(45, 12), (1000, 320), (777, 1024)])
def test_dropout_with_bf16_shadow(ops, oracle, rows, P):
    """lc_dropout_scale_bf16: the masked tensor and, in the same pass, its bf16 shadow (exactly the RNE rounding
    lc_cast_bf16 makes) - on column windows of wider buffers, in place; widths that leave lanes of the flat walk ragged."""
    rng = np.random.default_rng(rows + P)
    xh = rng.normal(size=(rows, 2 * P)).astype(np.float32)
    x = dev(xh)
    sh = torch.zeros((rows, 2 * P), dtype=torch.bfloat16, device="cuda")
    ops.dropout_scale(x[:, P:], 0.9, 31, 4, shadow=sh[:, P:])
    m = oracle.dropout_mask(31, 4, (rows, 1, P), 0.9).reshape(rows, P)
    want = xh[:, P:] * m
    got = x.cpu().numpy()
    assert np.array_equal(got[:, P:], want) and np.array_equal(got[:, :P], xh[:, :P])
    assert np.array_equal(sh[:, P:].float().cpu().numpy(), oracle.bf16_round(want))
    assert not sh[:, :P].float().cpu().numpy().any()
    nat, _ = ops.cast_bf16(x[:, P:].contiguous(), nat=True, tr=False)
    assert torch.equal(nat, sh[:, P:].contiguous())


@pytest.mark.parametrize("P,accumulate", [(12, False), (64, True), (1024, False)])
def test_dropout_vectorised_kernel_matches_oracle(ops, oracle, P, accumulate):
    """The 16-byte dropout kernel (P, leading dimensions multiples of 4, aligned windows): the same counter-based mask as
    the oracle's, on a column window of a wider buffer, in place and accumulating into another buffer."""
    T, B = 9, 5
    rng = np.random.default_rng(P)
    xh = rng.normal(size=(T * B, 2 * P)).astype(np.float32)
    m = oracle.dropout_mask(77, 3, (T, B, P), 0.9).reshape(T * B, P)
    x = dev(xh)
    if accumulate:
        yh = rng.normal(size=(T * B, P)).astype(np.float32)
        y = dev(yh)
        ops.dropout_scale(x[:, P:], 0.9, 77, 3, out=y, accumulate=True)
        np.testing.assert_allclose(y.cpu().numpy(), yh + xh[:, P:] * m, rtol=1e-6, atol=1e-6)
        assert np.array_equal(x.cpu().numpy(), xh)
    else:
        ops.dropout_scale(x[:, P:], 0.9, 77, 3)
        got = x.cpu().numpy()
        np.testing.assert_allclose(got[:, P:], xh[:, P:] * m, rtol=1e-6, atol=0)
        assert np.array_equal(got[:, :P], xh[:, :P])


@pytest.mark.parametrize("n,n_decay", [(10007, 9000), (10008, 9000)])       # element-wise and 16-byte paths of the kernels
@pytest.mark.parametrize("opt", ["sgd", "momentum", "adam"])
def test_optimizer_step(ops, oracle, opt, n, n_decay):
    rng = np.random.default_rng(17)
    p0 = rng.normal(size=n).astype(np.float32)
    g0 = (rng.normal(size=n) * 3).astype(np.float32)
    params = {"w": p0[:n_decay].copy(), "x/bias": p0[n_decay:].copy()}
    grads = {"w": g0[:n_decay].copy(), "x/bias": g0[n_decay:].copy()}
    P, G = dev(p0), dev(g0)
    state = torch.zeros(2 * n, device="cuda")
    norm = torch.zeros(2, device="cuda")
    ost = {}
    for step in (1, 2, 3):
        ops.optimizer_step(P, G, n_decay, 1e-5, 5.0, opt, 1e-2, step, state, norm)
        cl, nrm = oracle.l2_and_clip(params, grads, 5.0, 1e-5)
        oracle.apply_optimizer(opt, params, cl, ost, 1e-2)
        assert abs(norm[0].item() - nrm) / nrm < 1e-5
        ref = np.concatenate([params["w"], params["x/bias"]])
        np.testing.assert_allclose(P.cpu().numpy(), ref, atol=2e-6)
        G.copy_(dev(g0))       # grads buffer was modified in place (L2 added): restore for the next step


def test_posteriors_and_colsum_transpose(ops):
    rng = np.random.default_rng(19)
    x = rng.normal(size=(37, 44)).astype(np.float32)
    prior = rng.normal(size=44).astype(np.float32)
    got = ops.posteriors(dev(x), 0.7, True, True, dev(prior)).cpu().numpy()
    z = 0.7 * x.astype(np.float64)
    ref = z - z.max(1, keepdims=True) - np.log(np.exp(z - z.max(1, keepdims=True)).sum(1, keepdims=True)) - prior
    np.testing.assert_allclose(got, ref, atol=1e-5)
    np.testing.assert_allclose(ops.colsum(dev(x)).cpu().numpy(), x.sum(0), atol=1e-4)
    assert np.array_equal(ops.transpose(dev(x)).cpu().numpy(), x.T)


# ------------------------------------------------------------------------------------------ LSTM step kernels, bf16 operands
def _sig(x):
    return 1.0 / (1.0 + np.exp(-x))


def _interleave_cols(N):
    """column of gate g (TF order i,j,f,o) of unit n in the product's gate-interleaved layout"""
    n = np.arange(N)
    return [(n // 8) * 32 + g * 8 + (n % 8) for g in range(4)]


@pytest.mark.parametrize("T,B,N", [(9, 5, 32), (7, 40, 64), (5, 64, 96)])
def test_lstm_step_kernels_bf16(ops, oracle, T, B, N):
    """c5 recurrence: z_t = zx_t + bf16(m'_{t-1}) . bf16(R) (fp32 accumulate), fp32 gates/state; BPTT
    dm' = dh_t + bf16(dz_{t'}) . bf16(R^T).  Emulated in float64 with the SAME operand roundings.  Tolerance: when a
    recurrent value sits on a bf16 rounding boundary, the kernel's fp32 gate math (hardware exp2/rcp, ~1.5e-7) and
    the float64 emulation may round it to neighbouring bf16 values (2^-9 apart relative), which moves a
    pre-activation by up to |R| * 2^-9 ~ 1e-3; everything else agrees to ~1e-6."""
    rng = np.random.default_rng(T * 100 + B + N)
    cols = _interleave_cols(N)
    seq_len = np.sort(rng.integers(max(1, T // 2), T + 1, size=B))[::-1].astype(np.int32).copy()
    seq_len[0] = T
    fb = 5.0
    dirs_np = []
    for d in range(2):
        dirs_np.append(dict(zx=rng.normal(0, 1.0, size=(T, B, 4 * N)).astype(np.float32),
                            R=rng.normal(0, 0.15, size=(N, 4 * N)).astype(np.float32),
                            w_f=rng.normal(0, 0.3, size=N).astype(np.float32),
                            w_i=rng.normal(0, 0.3, size=N).astype(np.float32),
                            w_o=rng.normal(0, 0.3, size=N).astype(np.float32),
                            dh=rng.normal(0, 0.1, size=(T, B, N)).astype(np.float32), reverse=d))
    # ---- emulation
    exp = []
    for dd in dirs_np:
        Rr = oracle.bf16_round(dd["R"]).astype(np.float64)
        zx = dd["zx"].astype(np.float64)
        gates = np.zeros((T, B, 4 * N)); cs = np.zeros((T, B, N)); hs = np.zeros((T, B, N))
        hq = np.zeros((B, N)); cp = np.zeros((B, N))
        order = range(T - 1, -1, -1) if dd["reverse"] else range(T)
        for t in order:
            z = zx[t] + hq @ Rr
            zi, zj, zf, zo = (z[:, c] for c in cols)
            ia = _sig(zi + dd["w_i"] * cp); fa = _sig(zf + fb + dd["w_f"] * cp); ja = np.tanh(zj)
            cn = fa * cp + ia * ja
            oa = _sig(zo + dd["w_o"] * cn)
            h = oa * np.tanh(cn)
            act = (t < seq_len)[:, None]
            for g, v in zip(cols, (ia, ja, fa, oa)):
                gates[t][:, g] = np.where(act, v, 0.0)
            cs[t] = np.where(act, cn, 0.0); hs[t] = np.where(act, h, 0.0)
            cp = cs[t]
            hq = oracle.bf16_round(hs[t].astype(np.float32)).astype(np.float64)
        exp.append(dict(gates=gates, cs=cs, hs=hs))
    # ---- kernel forward
    sl = dev(seq_len)
    fd = []
    for dd in dirs_np:
        fd.append(dict(zx=dev(dd["zx"].reshape(T * B, 4 * N)), R=dev(dd["R"]), w_f=dev(dd["w_f"]), w_i=dev(dd["w_i"]),
                       w_o=dev(dd["w_o"]), cs=torch.empty((T * B, N), device="cuda"),
                       hs=torch.empty((T * B, N), device="cuda"), reverse=dd["reverse"]))
    ops.lstm_fwd(fd, sl, T, B, N, fb, bf16=True)
    for d in range(2):
        for got, want in ((fd[d]["cs"], exp[d]["cs"]), (fd[d]["hs"], exp[d]["hs"]), (fd[d]["zx"], exp[d]["gates"])):
            err = np.abs(got.cpu().numpy().reshape(want.shape) - want)
            assert err.max() < 1.5e-3 and np.median(err) < 2e-6, (err.max(), np.median(err))
    # ---- backward: emulate from the kernel's own saved gates / cells
    bd = []
    for d, dd in enumerate(dirs_np):
        RT = np.ascontiguousarray(dd["R"].T)
        gates = fd[d]["zx"].cpu().numpy().reshape(T, B, 4 * N).astype(np.float64)
        cs = fd[d]["cs"].cpu().numpy().reshape(T, B, N).astype(np.float64)
        RTr = oracle.bf16_round(RT).astype(np.float64)
        dz = np.zeros((T, B, 4 * N)); dzq = np.zeros((B, 4 * N)); dc = np.zeros((B, N))
        order = range(T) if dd["reverse"] else range(T - 1, -1, -1)
        for t in order:
            tprev = t + 1 if dd["reverse"] else t - 1
            cp = cs[tprev] if 0 <= tprev < T else np.zeros((B, N))
            dh = dd["dh"][t].astype(np.float64) + dzq @ RTr
            ia, ja, fa, oa = (gates[t][:, c] for c in cols)
            cn = cs[t]; tc = np.tanh(cn)
            do_pre = dh * tc * oa * (1 - oa)
            dcn = dc + dh * oa * (1 - tc * tc) + do_pre * dd["w_o"]
            di_pre = dcn * ja * ia * (1 - ia); dj_pre = dcn * ia * (1 - ja * ja); df_pre = dcn * cp * fa * (1 - fa)
            act = (t < seq_len)[:, None]
            dc = np.where(act, dcn * fa + di_pre * dd["w_i"] + df_pre * dd["w_f"], dc)
            for g, v in zip(cols, (di_pre, dj_pre, df_pre, do_pre)):
                dz[t][:, g] = np.where(act, v, 0.0)
            dzq = oracle.bf16_round(dz[t].astype(np.float32)).astype(np.float64)
        exp[d]["dz"] = dz
        bd.append(dict(gates=fd[d]["zx"], RT=dev(RT), w_f=fd[d]["w_f"], w_i=fd[d]["w_i"], w_o=fd[d]["w_o"], cs=fd[d]["cs"],
                       dh=dev(dd["dh"].reshape(T * B, N)), dpeep=torch.zeros((3, N), device="cuda"),
                       reverse=dd["reverse"]))
    ops.lstm_bwd(bd, sl, T, B, N, bf16=True)
    for d in range(2):
        got = bd[d]["gates"].cpu().numpy().reshape(T, B, 4 * N)
        scale = np.abs(exp[d]["dz"]).max()
        assert np.abs(got - exp[d]["dz"]).max() < 2e-3 * scale + 1e-6, (np.abs(got - exp[d]["dz"]).max(), scale)


@pytest.mark.parametrize("persistent", [True, False])
@pytest.mark.parametrize("T,B,N", [(9, 40, 1024), (7, 19, 96)])
def test_lstm_bf16_fused_shadows_are_the_exact_rounding(ops, oracle, T, B, N, persistent, monkeypatch):
    """lc_lstm_fwd_bf16 / lc_lstm_bwd_bf16 with hs_bf16 / dz_bf16: the optional bf16 copies of hs and dz that the persistent
    kernels write in the same pass (and every other schedule produces with a cast behind the recurrence) must be exactly
    the round-to-nearest-even bf16 of the float32 outputs of the same call."""
    if not persistent:
        monkeypatch.setenv("LC_LSTM_PERSISTENT", "0")
    rng = np.random.default_rng(T + B + N)
    G = 4 * N
    seq = np.sort(rng.integers(2, T + 1, size=B))[::-1].astype(np.int32).copy()
    seq[0] = T
    fdirs, keep = [], []
    for d in range(2):
        zx = dev(rng.normal(0, 1.0, size=(T * B, G)).astype(np.float32))
        R = dev((rng.normal(0, 1.0, size=(N, G)) / np.sqrt(N)).astype(np.float32))
        fdirs.append(dict(zx=zx, R=R, w_f=dev(rng.normal(0, .3, N).astype(np.float32)), w_i=dev(rng.normal(0, .3, N).astype(np.float32)),
                          w_o=dev(rng.normal(0, .3, N).astype(np.float32)), cs=torch.empty((T * B, N), device="cuda"),
                          hs=torch.empty((T * B, N), device="cuda"), reverse=(d == 1),
                          hs_bf16=torch.empty((T * B, N), dtype=torch.bfloat16, device="cuda")))
    ops.lstm_fwd(fdirs, dev(seq), T, B, N, 1.0, bf16=True)
    sched = ops.last_lstm_schedule()
    assert (sched["kind"] == "persistent_bf16") == persistent, sched
    for d in fdirs:
        hs = d["hs"].cpu().numpy()
        assert np.isfinite(hs).all() and np.abs(hs).max() > 0.01
        assert np.array_equal(d["hs_bf16"].float().cpu().numpy(), oracle.bf16_round(hs))
    bdirs = []
    for d in fdirs:
        RT = ops.transpose(d["R"])
        bdirs.append(dict(gates=d["zx"], RT=RT, w_f=d["w_f"], w_i=d["w_i"], w_o=d["w_o"], cs=d["cs"],
                          dh=dev(rng.normal(0, 1.0, size=(T * B, N)).astype(np.float32)), dpeep=torch.zeros((3, N), device="cuda"),
                          dbias=torch.zeros(G, device="cuda"), reverse=d["reverse"],
                          dz_bf16=torch.empty((T * B, G), dtype=torch.bfloat16, device="cuda")))
    ops.lstm_bwd(bdirs, dev(seq), T, B, N, bf16=True)
    sched = ops.last_lstm_schedule()
    assert (sched["kind"] == "persistent_bf16") == persistent and sched["backward"], sched
    for d in bdirs:
        dz = d["gates"].cpu().numpy()
        assert np.isfinite(dz).all() and np.abs(dz).max() > 1e-4
        assert np.array_equal(d["dz_bf16"].float().cpu().numpy(), oracle.bf16_round(dz))


# ------------------------------------------------------------------------------------------ persistent recurrence
@pytest.mark.parametrize("T,B,N,ndir,bf16", [
    (1500, 64, 320, 2, False),      # the recipes' layer size, 16 rows per XCD
    (1200, 32, 512, 2, False),      # largest fp32 slice (128 VGPRs of R per wave), 8 rows per XCD
    (700, 37, 48, 2, False),        # ragged K split, ragged row groups
    (900, 100, 256, 1, False),      # uni-directional: 8 row groups of 13 rows
    (600, 5, 16, 2, False),         # a single column tile, empty row groups
    (1000, 64, 1024, 2, True),      # config c5: bf16 operands, 256 VGPRs of R per wave, two (row, unit) pairs per thread
    (700, 40, 768, 2, True),        # bf16, ragged chunks
    (300, 33, 960, 2, True),        # bf16, the full-width (AGPR-resident) instantiations with a ragged last block
    (500, 19, 96, 1, True),         # bf16, small and ragged, uni-directional
])
def test_persistent_recurrence_equals_launch_train(ops, T, B, N, ndir, bf16, monkeypatch):
    """The one-launch schedule for small models (one XCD per direction and row group, state exchanged as tagged data
    through the XCD's L2) against the per-step launch train on the same inputs, forward and BPTT, over sequences long
    enough that a single stale or torn exchange would show: both evaluate the same recurrence and differ only in the
    summation order of the step GEMM (and <= 1 ulp on the exchanged dz), so the saved activations agree to ~1e-5."""
    g = torch.Generator().manual_seed(T + B + N)
    rows = T * B
    seq = torch.randint(T // 2, T + 1, (B,), generator=g, dtype=torch.int32)
    seq[0] = T
    seq = seq.cuda()

    def run():
        gg = torch.Generator().manual_seed(7 * N + B)
        fd, bd = [], []
        for d in range(ndir):
            fd.append(dict(zx=(torch.randn(rows, 4 * N, generator=gg) * 0.5).cuda(),
                           R=(torch.randn(N, 4 * N, generator=gg) * (0.5 / N ** 0.5)).cuda(),
                           w_f=(torch.randn(N, generator=gg) * 0.2).cuda(), w_i=(torch.randn(N, generator=gg) * 0.2).cuda(),
                           w_o=(torch.randn(N, generator=gg) * 0.2).cuda(),
                           cs=torch.empty(rows, N, device="cuda"), hs=torch.empty(rows, N, device="cuda"), reverse=d))
        ops.lstm_fwd(fd, seq, T, B, N, 1.0, bf16=bf16)
        for d in range(ndir):
            bd.append(dict(gates=fd[d]["zx"].clone(), RT=fd[d]["R"].t().contiguous(), w_f=fd[d]["w_f"], w_i=fd[d]["w_i"],
                           w_o=fd[d]["w_o"], cs=fd[d]["cs"], dh=(torch.randn(rows, N, generator=gg) * 0.1).cuda(),
                           dpeep=torch.zeros(3, N, device="cuda"), dbias=torch.zeros(4 * N, device="cuda"), reverse=d))
        ops.lstm_bwd(bd, seq, T, B, N, bf16=bf16)
        torch.cuda.synchronize()
        return fd, bd

    monkeypatch.setenv("LC_LSTM_PERSISTENT", "1")
    pf, pb = run()
    monkeypatch.setenv("LC_LSTM_PERSISTENT", "0")
    lf, lb = run()
    # bf16 operands: both schedules round the same fp32 state to bf16, but a state value that differs in its last fp32
    # bits (summation order) can fall on the other side of a bf16 rounding boundary (2^-9 relative), which moves a
    # pre-activation by ~|R| * 2^-9: the comparison is then statistical (mean) with a loose bound on the maximum
    tol = 2e-5 if not bf16 else 3e-2
    for d in range(ndir):
        for k in ("zx", "cs", "hs"):
            a, b = pf[d][k], lf[d][k]
            assert torch.isfinite(a).all()
            assert (a - b).abs().max().item() < tol, (d, k, (a - b).abs().max().item())
            if bf16:
                assert (a - b).abs().mean().item() < 2e-4, (d, k, (a - b).abs().mean().item())
        a, b = pb[d]["gates"], lb[d]["gates"]
        assert torch.isfinite(a).all()
        scale = b.abs().max().item()
        assert (a - b).abs().max().item() < tol * max(scale, 1.0), (d, "dz", (a - b).abs().max().item(), scale)
        if bf16:
            assert (a - b).abs().mean().item() < 2e-4 * max(scale, 1.0)
        for k in ("dpeep", "dbias"):
            a, b = pb[d][k], lb[d][k]
            assert (a - b).abs().max().item() < (1e-4 if not bf16 else 2e-2) * max(b.abs().max().item(), 1.0), (d, k)


# ------------------------------------------------------------------------------------------ fused GEMM epilogue
def _unfused(ops, out, keep, seed, stream0, P):
    """What the epilogue must reproduce: lc_dropout_scale on each column window + lc_cast_bf16 of the result."""
    for d in range(out.shape[1] // P):
        ops.dropout_scale(out[:, d * P:(d + 1) * P], keep, seed, stream0 + d)
    nat, _ = ops.cast_bf16(out, nat=True, tr=False)
    return nat


@pytest.mark.parametrize("form", ["f32", "f32_tb", "bf16_convert", "nt", "tn", "nn"])
@pytest.mark.parametrize("M,N,K,P", [(512, 512, 128, 256), (700, 648, 192, 324), (256, 256, 64, 256), (1300, 1024, 256, 512)])
def test_gemm_epilogue_is_the_separate_passes(ops, oracle, form, M, N, K, P):
    """lc_gemm_next_epilogue: the product that carries the DropoutWrapper mask and the bf16 shadow in its epilogue gives
    BIT-identical fp32 and bf16 results to product -> lc_dropout_scale per column window -> lc_cast_bf16, for every
    product form, with beta / bias (applied before the mask), ragged shapes (right / bottom strips carry their origin)
    and an output that is a column window of a wider buffer."""
    if form in ("tn", "nn") and (M % 256 or N % 256):
        pytest.skip("K-major kernels take whole 256-tiles only")
    rng = np.random.default_rng(M + 3 * N + 5 * K)
    A = rng.normal(size=(M, K)).astype(np.float32)
    B = rng.normal(size=(K, N)).astype(np.float32)
    bias = dev(rng.normal(size=N).astype(np.float32))
    C0 = rng.normal(size=(M, N + 8)).astype(np.float32)
    keep, seed, stream0 = 0.8, 1234567, 6

    def product(out, epilogue):
        kw = dict(out=out, alpha=0.5, beta=2.0, bias=bias, epilogue=epilogue)
        if form == "f32":
            ops.gemm(dev(A), dev(B), **kw)
        elif form == "f32_tb":
            ops.gemm(dev(A), dev(np.ascontiguousarray(B.T)), tb=True, **kw)
        elif form == "bf16_convert":
            ops.gemm(dev(A), dev(B), bf16=True, **kw)
        elif form == "nt":
            a, _ = ops.cast_bf16(dev(A), nat=True, tr=False)
            _, bt = ops.cast_bf16(dev(B), nat=False, tr=True)
            ops.gemm_bf16_nt(a, bt, K=K, **kw)
        elif form == "tn":
            a, _ = ops.cast_bf16(dev(np.ascontiguousarray(A.T)), nat=True, tr=False)
            b, _ = ops.cast_bf16(dev(B), nat=True, tr=False)
            ops.gemm_bf16_tn(a, b, **kw)
        else:
            a, _ = ops.cast_bf16(dev(A), nat=True, tr=False)
            b, _ = ops.cast_bf16(dev(B), nat=True, tr=False)
            ops.gemm_bf16_nn(a, b, **kw)

    plain = dev(C0)[:, 8:]
    product(plain, None)
    want16 = _unfused(ops, plain, keep, seed, stream0, P)
    fused = dev(C0)[:, 8:]
    sh = torch.zeros((M, N + 16), dtype=torch.bfloat16, device="cuda")[:, 16:]
    product(fused, ops.Epilogue(keep, seed, stream0, P, sh))
    assert torch.equal(fused, plain)
    assert torch.equal(sh.view(torch.int16), want16.view(torch.int16))
    zeros = float((fused == 0).float().mean())
    assert abs(zeros - (1 - keep)) < 0.02, zeros
    # one-shot: the next product is plain again
    again = dev(C0)[:, 8:]
    product(again, None)
    ref = dev(C0)[:, 8:]
    product(ref, ops.Epilogue(1.0, 0, 0, 1, None))          # keep = 1, no shadow: a no-op epilogue
    assert torch.equal(again, ref)


def test_gemm_epilogue_shadow_only_and_disarm(ops):
    """keep = 1 with a shadow: the bf16 copy alone (no mask); e = NULL disarms a pending epilogue; a refused product still
    consumes it (the next product of the thread is never armed by accident)."""
    import ctypes
    from lstm_ctc_amd import _lib
    lib = _lib.load()
    g = torch.Generator(device="cuda").manual_seed(3)
    A = torch.randn(300, 96, device="cuda", generator=g)
    B = torch.randn(96, 200, device="cuda", generator=g)
    sh = torch.zeros(300, 200, dtype=torch.bfloat16, device="cuda")
    out = ops.gemm(A, B, epilogue=ops.Epilogue(shadow=sh))
    assert torch.equal(sh, out.to(torch.bfloat16))
    e = _lib.GemmEpilogue(0.5, 1, 0, 100, None, 0)
    assert lib.lc_gemm_next_epilogue(ctypes.byref(e)) == 0
    assert lib.lc_gemm_next_epilogue(None) == 0              # disarm
    assert torch.equal(ops.gemm(A, B), out)
    assert lib.lc_gemm_next_epilogue(ctypes.byref(e)) == 0
    assert lib.lc_gemm_f32(0, 0, 300, 200, 96, 1.0, None, 96, None, 200, 0.0, None, 200, None, None, 0, None) != 0
    assert torch.equal(ops.gemm(A, B), out)                  # the refused call consumed the arm
    bad = _lib.GemmEpilogue(0.0, 1, 0, 100, None, 0)
    assert lib.lc_gemm_next_epilogue(ctypes.byref(bad)) != 0  # keep = 0 refused
    assert torch.equal(ops.gemm(A, B), out)


# ------------------------------------------------------------------------------------------ fp32 products as bf16 x 3
def test_split_bf16x3_is_exact(ops):
    """lc_split_bf16x3: hi + mid + lo == x EXACTLY for every finite fp32 (3 x 8 significand bits), each term is the RNE bf16
    of what the previous terms left, the k-tiled layout is [row][k / 16][term][16], and the pad columns are zero."""
    rng = np.random.default_rng(3)
    rows, K = 37, 100
    x = (rng.normal(size=(rows, K)) * np.exp(rng.uniform(-20, 20, size=(rows, K)))).astype(np.float32)
    x[0, :4] = [0.0, -0.0, 1.0, np.float32(1e-30)]
    wide = dev(np.concatenate([np.zeros((rows, 4), np.float32), x], axis=1))
    out = ops.split_bf16x3(wide[:, 4:])                                   # a column window of a wider buffer
    Kp = (K + 15) // 16 * 16
    assert out.shape == (rows, 3 * Kp)
    t = out.view(rows, Kp // 16, 3, 16).float().cpu().numpy()
    hi, mid, lo = (t[:, :, i, :].reshape(rows, Kp) for i in range(3))
    assert (hi[:, K:] == 0).all() and (mid[:, K:] == 0).all() and (lo[:, K:] == 0).all()
    s = hi[:, :K].astype(np.float64) + mid[:, :K].astype(np.float64) + lo[:, :K].astype(np.float64)
    assert np.array_equal(s, x.astype(np.float64))
    xt = torch.from_numpy(x)
    hi_ref = xt.to(torch.bfloat16).float()
    mid_ref = (xt - hi_ref).to(torch.bfloat16).float()
    assert np.array_equal(hi[:, :K], hi_ref.numpy()) and np.array_equal(mid[:, :K], mid_ref.numpy())
    # the edges of "exactly": below ~1e-30 the low terms' residuals are fp32 denormals and flush (absolute error below
    # 2^-126, i.e. nothing next to any normal-range term of a sum); non-finite and > bf16-max inputs give NaN, not Inf
    tiny = (rng.normal(size=(8, 64)) * np.exp(rng.uniform(-87, -66, size=(8, 64)))).astype(np.float32)
    tt = ops.split_bf16x3(dev(tiny)).view(8, 4, 3, 16).float().cpu().numpy()
    ssum = sum(tt[:, :, i, :].reshape(8, 64).astype(np.float64) for i in range(3))
    assert np.abs(ssum - tiny).max() < 2.0 ** -125 and np.isfinite(tt).all()
    edge = torch.tensor([[float("inf"), float("nan"), 3.4e38, 1.0] + [0.0] * 12], dtype=torch.float32, device="cuda")
    e3 = ops.split_bf16x3(edge).view(1, 1, 3, 16).float().sum(dim=2).cpu().numpy()[0, 0]
    assert np.isnan(e3[:3]).all() and e3[3] == 1.0


@pytest.mark.parametrize("M,N,K", [(256, 256, 16), (512, 768, 1024), (700, 520, 40), (1000, 300, 100), (3, 5, 7),
                                   (2048, 1024, 4096)])
def test_gemm_bf16x3_is_fp32_grade(ops, M, N, K):
    """lc_gemm_bf16x3_nt (six bf16 term products per fp32 product, fp32 accumulate) against float64 on the SAME fp32
    operands: its error must not exceed the fp32 MFMA kernel's by more than a rounding's worth - and is measured next to
    it - for ragged M / N (zero-filled by the buffer descriptors), K not a multiple of 16 (zero pad), alpha / beta / bias,
    and operands of mixed magnitude (a bf16-only product would be off by 2^-9)."""
    rng = np.random.default_rng(M + 3 * N + 5 * K)
    A = (rng.normal(size=(M, K)) * np.exp(rng.uniform(-3, 3, size=(M, K)))).astype(np.float32)
    B = (rng.normal(size=(N, K)) * np.exp(rng.uniform(-3, 3, size=(N, K)))).astype(np.float32)
    bias = rng.normal(size=N).astype(np.float32)
    C0 = rng.normal(size=(M, N)).astype(np.float32)
    ref = 0.5 * (A.astype(np.float64) @ B.astype(np.float64).T) + 2.0 * C0 + bias
    mag = 0.5 * (np.abs(A).astype(np.float64) @ np.abs(B).astype(np.float64).T) + 2.0 * np.abs(C0) + np.abs(bias)
    a3, b3 = ops.split_bf16x3(dev(A)), ops.split_bf16x3(dev(B))
    out = dev(C0)
    ops.gemm_bf16x3_nt(a3, b3, K, out=out, alpha=0.5, beta=2.0, bias=dev(bias))
    f32 = dev(C0)
    ops.gemm(dev(A), dev(B), tb=True, out=f32, alpha=0.5, beta=2.0, bias=dev(bias))
    e3 = np.abs(out.cpu().numpy() - ref) / mag                            # error relative to the sum of magnitudes
    e32 = np.abs(f32.cpu().numpy() - ref) / mag
    assert e3.max() < 3e-7 * max(1.0, np.sqrt(K) / 4), (e3.max(), e32.max())
    assert e3.max() <= 2.0 * e32.max() + 6e-8 and np.sqrt((e3 ** 2).mean()) <= 1.5 * np.sqrt((e32 ** 2).mean()) + 1e-8, \
        (e3.max(), e32.max())
    # with the fused epilogue: bit-identical to the separate passes
    plain = out.clone()
    if N % 2 == 0:
        P = N // 2
        want = plain.clone()
        for d in range(2):
            ops.dropout_scale(want[:, d * P:(d + 1) * P], 0.75, 99, 4 + d)
        fused = dev(C0)
        ops.gemm_bf16x3_nt(a3, b3, K, out=fused, alpha=0.5, beta=2.0, bias=dev(bias), epilogue=ops.Epilogue(0.75, 99, 4, P))
        assert torch.equal(fused, want)


@pytest.mark.parametrize("M,N,K", [(256, 256, 16), (300, 520, 1000), (40, 512, 5000), (1024, 768, 8192), (33, 17, 100),
                                   (512, 256, 20000)])
def test_gemm_bf16x3_tn_is_fp32_grade(ops, M, N, K):
    """lc_gemm_bf16x3_tn: C = alpha A^T B + beta C + bias on K-MAJOR x3 shadows (transposing LDS reads), against float64 and
    next to the fp32 kernel: ragged M / N (the columns past them are never zeroed - they only meet output rows / columns
    that are not stored), K not a multiple of 16 (rows past K read as zeros), K splits with the reduction pass, row windows
    of larger shadows one step apart (dR = hs_prev^T dZ)."""
    rng = np.random.default_rng(M + 3 * N + 5 * K)
    shift = 24
    A = (rng.normal(size=(K + shift, M)) * np.exp(rng.uniform(-3, 3, size=(K + shift, M)))).astype(np.float32)
    B = (rng.normal(size=(K + shift, N)) * np.exp(rng.uniform(-3, 3, size=(K + shift, N)))).astype(np.float32)
    bias = rng.normal(size=N).astype(np.float32)
    C0 = rng.normal(size=(M, N)).astype(np.float32)
    Aw, Bw = A[:K], B[shift:shift + K]
    ref = 0.5 * (Aw.astype(np.float64).T @ Bw.astype(np.float64)) + 2.0 * C0 + bias
    mag = 0.5 * (np.abs(Aw).astype(np.float64).T @ np.abs(Bw).astype(np.float64)) + 2.0 * np.abs(C0) + np.abs(bias)
    a3, b3 = ops.split_bf16x3(dev(A)), ops.split_bf16x3(dev(B))
    out = dev(C0)
    ops.gemm_bf16x3_tn(a3[:K], b3[shift:shift + K], M, N, out=out, alpha=0.5, beta=2.0, bias=dev(bias))
    f32 = dev(C0)
    ops.gemm(dev(np.ascontiguousarray(Aw)), dev(np.ascontiguousarray(Bw)), ta=True, out=f32, alpha=0.5, beta=2.0, bias=dev(bias))
    e3 = np.abs(out.cpu().numpy() - ref) / mag
    e32 = np.abs(f32.cpu().numpy() - ref) / mag
    assert e3.max() < 3e-7 * max(1.0, np.sqrt(K) / 4), (e3.max(), e32.max())
    assert e3.max() <= 2.0 * e32.max() + 6e-8 and np.sqrt((e3 ** 2).mean()) <= 1.5 * np.sqrt((e32 ** 2).mean()) + 1e-8, \
        (e3.max(), e32.max())


def _adversarial_operands(kind, rng, R, C, K):
    """Two fp32 operand matrices, A [R, K] and B [C, K] (the product is A B^T), built to break a split-operand product."""
    if kind == "cancellation":          # every dot product cancels to ~1e-6 of its sum of magnitudes
        h = K // 2
        a, b = rng.normal(size=(R, h)), rng.normal(size=(C, h))
        A = np.concatenate([a, a], axis=1).astype(np.float32)
        B = np.concatenate([b, -b * (1.0 + 2.0 ** -21)], axis=1).astype(np.float32)
    elif kind == "spread":              # magnitudes spread over 2^24 inside every row of either operand
        A = (rng.normal(size=(R, K)) * 2.0 ** rng.uniform(-12, 12, size=(R, K))).astype(np.float32)
        B = (rng.normal(size=(C, K)) * 2.0 ** rng.uniform(-12, 12, size=(C, K))).astype(np.float32)
    elif kind == "tiny":                # operands in [1e-38, 1e-30] (their low terms flush) against huge partners
        A = (rng.choice([-1.0, 1.0], size=(R, K)) * 10.0 ** rng.uniform(-38, -30, size=(R, K))).astype(np.float32)
        B = (rng.normal(size=(C, K)) * 1e25).astype(np.float32)
    elif kind == "zeros_denormals":     # +-0, fp32 denormals and normal values mixed
        A = rng.normal(size=(R, K)).astype(np.float32)
        u = rng.uniform(size=(R, K))
        A[u < 0.15] = 0.0
        A[(u >= 0.15) & (u < 0.3)] = -0.0
        den = (u >= 0.3) & (u < 0.45)
        A[den] = (rng.choice([-1.0, 1.0], size=int(den.sum())) * 10.0 ** rng.uniform(-44.5, -38.2, size=int(den.sum()))
                  ).astype(np.float32)
        B = rng.normal(size=(C, K)).astype(np.float32)
        B[rng.uniform(size=(C, K)) < 0.1] = -0.0
    else:
        raise KeyError(kind)
    return A, B


@pytest.mark.parametrize("form", ["nt", "tn"])
@pytest.mark.parametrize("kind", ["cancellation", "spread", "tiny", "zeros_denormals"])
def test_gemm_bf16x3_adversarial_operands(ops, kind, form):
    """The bound include/lstm_ctc_hip.h states for the split-operand products, on operands chosen against them:
        |C - sum_k a_k b_k|  <=  3e-7 max(1, sqrt(K) / 4) sum_k |a_k| |b_k|  +  2^-125 sum_k (|a_k| + |b_k|)
    - the first term is fp32 accumulation (the fp32 MFMA kernel is held to the same, and measured next to it), the second
    the flush of split terms below 2^-126 (it only shows when an operand below ~1e-30 meets one above ~1e+8).  Cases: rows
    whose products cancel to 1e-6 of their magnitude sum, per-row magnitude spread 2^24, operands in [1e-38, 1e-30], +-0 and
    fp32 denormals.  NT = row operands (forward / dX products), TN = K-major operands (weight gradients)."""
    rng = np.random.default_rng(len(kind) * 7 + len(form))
    R, C, K = (300, 260, 1024) if form == "nt" else (272, 300, 2000)
    A, B = _adversarial_operands(kind, rng, R, C, K)
    A64, B64 = A.astype(np.float64), B.astype(np.float64)
    ref = A64 @ B64.T
    mag = np.abs(A64) @ np.abs(B64).T
    flush = 2.0 ** -125 * (np.abs(A64).sum(axis=1)[:, None] + np.abs(B64).sum(axis=1)[None, :])
    if kind == "cancellation":
        assert np.median(mag / np.maximum(np.abs(ref), 1e-300)) >= 1e6
    if form == "nt":
        out = ops.gemm_bf16x3_nt(ops.split_bf16x3(dev(A)), ops.split_bf16x3(dev(B)), K)
        f32 = ops.gemm(dev(A), dev(B), tb=True)
    else:
        At, Bt = np.ascontiguousarray(A.T), np.ascontiguousarray(B.T)             # K-major [K, R], [K, C]
        out = ops.gemm_bf16x3_tn(ops.split_bf16x3(dev(At)), ops.split_bf16x3(dev(Bt)), R, C)
        f32 = ops.gemm(dev(At), dev(Bt), ta=True)
    got, got32 = out.cpu().numpy().astype(np.float64), f32.cpu().numpy().astype(np.float64)
    assert np.isfinite(got).all()
    bound = 3e-7 * max(1.0, np.sqrt(K) / 4) * mag + flush
    assert (np.abs(got - ref) <= bound + 1e-300).all(), float((np.abs(got - ref) / np.maximum(bound, 1e-300)).max())
    if kind != "tiny":          # next to the fp32 MFMA kernel on the same operands: not a narrower product
        e3, e32 = np.abs(got - ref) / np.maximum(mag, 1e-300), np.abs(got32 - ref) / np.maximum(mag, 1e-300)
        assert e3.max() <= 2.0 * e32.max() + 6e-8 and np.sqrt((e3 ** 2).mean()) <= 1.5 * np.sqrt((e32 ** 2).mean()) + 1e-8, \
            (e3.max(), e32.max())


def test_gemm_bf16x3_tn_slices_k_for_the_descriptor_reach(ops):
    """A K slice of a K-major operand is addressed through one buffer descriptor (2 GB): K * ld * 2 bytes beyond that must
    split K further, whatever the tile count (T = 1000 at --batch-size 256 made every bf16x3 train step
    fail).  Here: 100 000 rows of a 12288-wide shadow (2.4 GB per operand), M = N = 256."""
    from lstm_ctc_amd import _lib
    lib = _lib.load()
    K, M, N, ld = 100000, 256, 256, 12288
    # (one tile: the CU-fill rule already slices K 32 ways here; a 4096 x 4096 gradient - 256 tiles, one slice by that rule -
    # is where the descriptor reach alone forces slices)
    assert lib.lc_gemm_bf16x3_tn_workspace_bytes_ld(M, N, K, ld, ld) >= lib.lc_gemm_bf16x3_tn_workspace_bytes(M, N, K) > 0
    reach = (0x70000000 // (3 * 4096 * 2)) // 16 * 16          # rows of a 4096-wide x3 shadow one descriptor covers
    assert lib.lc_gemm_bf16x3_tn_workspace_bytes(4096, 4096, 256000) == -(-256000 // reach) * 4096 * 4096 * 4
    assert lib.lc_gemm_bf16x3_tn_workspace_bytes_ld(4096, 4096, 8192, 0, 0) == 0
    rng = np.random.default_rng(3)
    A = rng.normal(size=(K, M)).astype(np.float32)
    B = rng.normal(size=(K, N)).astype(np.float32)
    wideA = torch.zeros((K, ld), dtype=torch.bfloat16, device="cuda")
    wideB = torch.zeros((K, ld), dtype=torch.bfloat16, device="cuda")
    a3, b3 = wideA[:, :3 * M], wideB[:, 3 * N:6 * N]
    a3.copy_(ops.split_bf16x3(dev(A)))
    b3.copy_(ops.split_bf16x3(dev(B)))
    out = ops.gemm_bf16x3_tn(a3, b3, M, N)
    ref = A.astype(np.float64).T @ B.astype(np.float64)
    mag = np.abs(A).astype(np.float64).T @ np.abs(B).astype(np.float64)
    assert (np.abs(out.cpu().numpy() - ref) / mag).max() < 3e-7 * np.sqrt(K) / 4
    del wideA, wideB, a3, b3
    torch.cuda.empty_cache()


# ---------------------------------------------------------------------------------------------- split-operand recurrences
@pytest.mark.parametrize("N,B,ndir", [(1024, 64, 2), (768, 40, 2), (1024, 100, 1), (512, 32, 2), (384, 20, 2), (256, 64, 1),
                                      (128, 9, 2), (320, 32, 2), (448, 17, 2), (192, 5, 1), (64, 16, 2), (512, 128, 1)])
def test_split_operand_recurrence_is_fp32_grade(ops, N, B, ndir):
    """lc_lstm_fwd_x3 / lc_lstm_bwd_x3 at every width a split-operand kernel exists for, teacher-forced against float64: each
    step's output from the KERNEL'S OWN previous state / dz, so nothing cascades and a wrong term pair, k order or piece
    layout in the step product shows at 1e-3 (a dropped low term at 1e-5) - next to the fp32 kernels on the same inputs:
    the split-operand step must be as close to float64 as the fp32 step (measured: 2.0e-7 both forward, 5e-8 / 6e-8 BPTT)."""
    T = 9
    g = torch.Generator().manual_seed(N + B)
    rows = T * B
    seq = torch.full((B,), T, dtype=torch.int32)
    seq[-2:] = torch.tensor([T - 3, 1])[: min(2, B)]
    n_ = np.arange(N)
    cols = [torch.from_numpy((n_ // 8) * 32 + k * 8 + (n_ % 8)).cuda() for k in range(4)]

    def make():
        gg = torch.Generator().manual_seed(N + B)
        fd = [dict(zx=(torch.randn(rows, 4 * N, generator=gg) * 0.5).cuda(),
                   R=(torch.randn(N, 4 * N, generator=gg) * (0.5 / N ** 0.5)).cuda(),
                   w_f=(torch.randn(N, generator=gg) * 0.2).cuda(), w_i=(torch.randn(N, generator=gg) * 0.2).cuda(),
                   w_o=(torch.randn(N, generator=gg) * 0.2).cuda(),
                   cs=torch.zeros(rows, N, device="cuda"), hs=torch.zeros(rows, N, device="cuda"), reverse=d)
              for d in range(ndir)]
        dh = [(torch.randn(rows, N, generator=gg) * 0.1).cuda() for _ in range(ndir)]
        return fd, dh

    seqd = seq.cuda()
    act_t = lambda t: (t < seqd)[:, None]
    z64 = lambda *s: torch.zeros(*s, dtype=torch.float64, device="cuda")
    err = {}
    for x3 in (False, True):
        fd, dh = make()
        zx0 = [d["zx"].clone() for d in fd]
        ops.lstm_fwd(fd, seqd, T, B, N, 1.0, x3=x3)
        kind_f = ops.last_lstm_schedule()["kind"]
        worst = 0.0
        for d in range(ndir):
            R = fd[d]["R"].double()
            wf, wi, wo = (fd[d][k].double() for k in ("w_f", "w_i", "w_o"))
            hs, cs = fd[d]["hs"].view(T, B, N).double(), fd[d]["cs"].view(T, B, N).double()
            zx = zx0[d].view(T, B, 4 * N).double()
            for t in range(T):
                tp = t + 1 if fd[d]["reverse"] else t - 1
                hp, cp = (hs[tp], cs[tp]) if 0 <= tp < T else (z64(B, N), z64(B, N))
                z = zx[t] + hp @ R
                ia, fa = torch.sigmoid(z[:, cols[0]] + wi * cp), torch.sigmoid(z[:, cols[2]] + 1.0 + wf * cp)
                cn = fa * cp + ia * torch.tanh(z[:, cols[1]])
                want = torch.where(act_t(t), torch.sigmoid(z[:, cols[3]] + wo * cn) * torch.tanh(cn), z64(B, N))
                worst = max(worst, float((hs[t] - want).abs().max()))
        bd = [dict(gates=fd[d]["zx"].clone(), RT=fd[d]["R"].t().contiguous(), w_f=fd[d]["w_f"], w_i=fd[d]["w_i"],
                   w_o=fd[d]["w_o"], cs=fd[d]["cs"], dh=dh[d], dpeep=torch.zeros(3, N, device="cuda"),
                   dbias=torch.zeros(4 * N, device="cuda"), reverse=d) for d in range(ndir)]
        ops.lstm_bwd(bd, seqd, T, B, N, x3=x3)
        kind_b = ops.last_lstm_schedule()["kind"]
        worst_b, scale_b = 0.0, 0.0
        for d in range(ndir):
            RT = bd[d]["RT"].double()
            wf, wi, wo = (fd[d][k].double() for k in ("w_f", "w_i", "w_o"))
            gts, cs = fd[d]["zx"].view(T, B, 4 * N).double(), fd[d]["cs"].view(T, B, N).double()
            dz = bd[d]["gates"].view(T, B, 4 * N).double()
            dc = z64(B, N)
            order = list(range(T)) if d else list(range(T - 1, -1, -1))
            for s_, t in enumerate(order):
                tprev = t + 1 if d else t - 1
                cp = cs[tprev] if 0 <= tprev < T else z64(B, N)
                dhh = dh[d].view(T, B, N)[t].double() + (dz[order[s_ - 1]] @ RT if s_ else z64(B, N))
                ia, ja, fa, oa = (gts[t][:, c] for c in cols)
                tc = torch.tanh(cs[t])
                do_pre = dhh * tc * oa * (1 - oa)
                dcn = dc + dhh * oa * (1 - tc * tc) + do_pre * wo
                di_pre, dj_pre, df_pre = dcn * ja * ia * (1 - ia), dcn * ia * (1 - ja * ja), dcn * cp * fa * (1 - fa)
                a_ = act_t(t)
                dc = torch.where(a_, dcn * fa + di_pre * wi + df_pre * wf, dc)
                want = z64(B, 4 * N)
                for c, v in zip(cols, (di_pre, dj_pre, df_pre, do_pre)):
                    want[:, c] = torch.where(a_, v, torch.zeros_like(v))
                worst_b = max(worst_b, float((dz[t] - want).abs().max()))
                scale_b = max(scale_b, float(want.abs().max()))
        err[x3] = (worst, worst_b / scale_b, kind_f, kind_b)
    pair = N in (768, 1024)
    assert err[True][2] == ("persistent_x3_xcd_pair" if pair else "persistent_x3"), err
    assert err[True][3] == ("persistent_x3_xcd_pair" if pair else "persistent_x3"), err
    assert err[False][2] in ("persistent_f32_xcd_pair", "persistent_f32")
    assert err[True][0] < 1e-6 and err[True][0] <= 2.0 * err[False][0] + 1e-7, err        # forward: |h - f64 step|
    assert err[True][1] < 1e-6 and err[True][1] <= 2.0 * err[False][1] + 1e-7, err        # BPTT: |dz - f64 step| / max |dz|


@pytest.mark.parametrize("N,B,ndir", [(1024, 64, 2), (768, 40, 2), (1024, 70, 1), (512, 20, 2), (320, 32, 2), (128, 7, 1), (640, 16, 2)])
def test_bwd_x3_writes_the_exact_shadow_of_dz(ops, N, B, ndir):
    """lc_lstm_bwd_x3 with dz_bf16 set: the x3 shadow of dz ([T * B, 12 N], the operand of the dX / dKx / dR products) must be
    the EXACT three-term split of the saved dz, bit for bit what lc_split_bf16x3 makes of it - whether a split-operand
    kernel's producers wrote it (XCD pairs at N = 768 / 1024, one XCD at N = 64 .. 512: their exchange terms carry a
    generation tag in gate i, the shadow must not) or the split pass behind another schedule did (N = 640)."""
    T = 7
    g = torch.Generator().manual_seed(N + 7 * B)
    rows = T * B
    seq = torch.full((B,), T, dtype=torch.int32)
    seq[-1] = 2
    seqd = seq.cuda()
    fd = [dict(zx=(torch.randn(rows, 4 * N, generator=g) * 0.5).cuda(), R=(torch.randn(N, 4 * N, generator=g) * (0.5 / N ** 0.5)).cuda(),
               w_f=(torch.randn(N, generator=g) * 0.2).cuda(), w_i=(torch.randn(N, generator=g) * 0.2).cuda(),
               w_o=(torch.randn(N, generator=g) * 0.2).cuda(), cs=torch.zeros(rows, N, device="cuda"),
               hs=torch.zeros(rows, N, device="cuda"), reverse=d) for d in range(ndir)]
    ops.lstm_fwd(fd, seqd, T, B, N, 1.0, x3=True)
    bd = [dict(gates=fd[d]["zx"], RT=fd[d]["R"].t().contiguous(), w_f=fd[d]["w_f"], w_i=fd[d]["w_i"], w_o=fd[d]["w_o"],
               cs=fd[d]["cs"], dh=(torch.randn(rows, N, generator=g) * 0.1).cuda(), dpeep=torch.zeros(3, N, device="cuda"),
               dbias=torch.zeros(4 * N, device="cuda"), reverse=d,
               dz_x3=torch.full((rows, 12 * N), float("nan"), dtype=torch.bfloat16, device="cuda")) for d in range(ndir)]
    ops.lstm_bwd(bd, seqd, T, B, N, x3=True)
    sch = ops.last_lstm_schedule()
    assert sch["dz_shadow_in_kernel"] == (N in (768, 1024) or (N <= 512 and N % 64 == 0)) and sch["backward"], sch
    for d in range(ndir):
        want = ops.split_bf16x3(bd[d]["gates"])
        assert float(bd[d]["gates"].abs().max()) > 0
        assert torch.equal(bd[d]["dz_x3"].view(torch.int16), want.view(torch.int16)), d
