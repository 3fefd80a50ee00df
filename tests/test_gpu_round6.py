"""Round-6 GPU tests.

* Eight ranks on this box's one GPU (gloo): the exact `python bench.py --gpus 8` child-launch path and
  `bin/nnet-train.py` under torch.distributed.run - every per-layer gradient bucket, finish(), the rank-health
  all-gather, the per-rank dropout streams, the every-8th-batch sharding - with the updated parameters compared
  against ONE process on the 8 x B batch (all-reduce before clip / update).
* The host-arithmetic rows (clip, Adam, edit distance) through the C ABI against the vectors TensorFlow's own unit
  tests hold (tests/golden/tf_host_arith_known_answers.json).
* The bf16 persistent recurrences at a width with a partial last workgroup (N = 832) on exact-size hipMalloc buffers
  (8-byte pair loads clamped per unit used to read past the end of the tensors).
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HERE = os.path.dirname(os.path.abspath(__file__))


def _clean_env(**extra):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **extra)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "TORCHELASTIC_RUN_ID", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


# ------------------------------------------------------------------------------------- 8 ranks: bench.py --gpus 8
def _one_process_on_the_global_batch(name, world, n_steps):
    """The same model on the concatenation of every rank's batch, for n_steps steps: (flat parameters after, before)."""
    import bench
    from lstm_ctc_amd import ops
    from lstm_ctc_amd.nnet.graph import create_graph_for_training_ctc
    w = bench.WORKLOADS[name]
    graph = create_graph_for_training_ctc(None, w["cfg"], learn_rate=4e-4, clip_norm=5.0, optimizer="adam",
                                          device=torch.device("cuda", 0), seed=123)
    parts = [bench.synth_batch(w, r, "cuda:0") for r in range(world)]
    x = torch.cat([p[0] for p in parts], dim=1).contiguous()              # [T, 8 B, D]
    seq = torch.cat([p[1] for p in parts]).contiguous()
    labels = torch.cat([p[2] for p in parts]).contiguous()
    offs = (torch.arange(world * w["B"] + 1, dtype=torch.int64) * w["L"]).to(torch.int32).cuda()
    init = graph.model.ps.flat.detach().cpu().clone()
    with ops.force_launch_train():                                        # the schedule the co-tenant ranks ran
        for _ in range(n_steps):
            graph.step_device(x, seq, labels, offs, w["L"], int(labels.numel()), fetch_eval=False)
    torch.cuda.synchronize()
    return graph.model.ps.flat.detach().cpu(), init


@pytest.mark.timeout(1800)
@pytest.mark.parametrize("name,buckets,overlap", [("rehearsal_keep1", "1", "0"), ("rehearsal_keep1", "1", "1"),
                                                  ("rehearsal_keep1", "0", "0"), ("rehearsal", "1", "0")])
def test_bench_eight_ranks_rehearsal(tmp_path, name, buckets, overlap):
    """`python bench.py --gpus 8` (NO torchrun in front: bench.py starts its ranks itself, as the driver's N > 1 runs may)
    with eight gloo ranks on GPU 0, c4's layer structure at toy width.  Without dropout the replicas after 3 steps must
    equal one process that saw the 8 x B batch; with dropout (per-rank streams) the eight replicas must still be
    bit-identical to each other and differ from the no-dropout result.  overlap = "0": the wide models' backward (c4 / c5:
    weight gradients on the main stream, a layer's bucket issued behind the BPTT of the layer below - a toy width would
    otherwise take c2 / c3's side-stream schedule, where the whole gradient goes out in finish())."""
    world, steps, warmup = 8, 2, 1
    dump = tmp_path / "dump"
    dump.mkdir()
    env = _clean_env(LC_BENCH_SHARED_GPU="1", LC_LSTM_PERSISTENT="0", LC_DP_BUCKETS=buckets, LC_OVERLAP_WGRAD=overlap,
                     LC_BENCH_DUMP_DIR=str(dump))
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--backend", "gloo", "--workload", name,
           "--steps", str(steps), "--warmup", str(warmup), "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, timeout=1500, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr.decode()[-4000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1                                                # rank 0 alone prints
    line = json.loads(lines[0])
    cfg = line["config"]
    assert line["n_gpus"] == world and cfg["parallelism"] == "dp8" and cfg["rccl_ranks"] == world
    assert cfg["global_batch"] == world * 2 and cfg["launched_by"].startswith("bench.py self-launch")
    assert cfg["dp_buckets"] == (buckets == "1" and overlap == "0") and cfg["collective_backend"] == "gloo"
    # 5 layers: the ranges of layers 4, 3, 2, 1 go out behind the BPTTs of layers 3, 2, 1, 0; layer 0, the head and the biases in finish()
    assert cfg["dp_bucket_ranges_per_step"] == (4 if cfg["dp_buckets"] else 0)
    assert cfg["ranks"]["n"] == world and cfg["ranks"]["lstm_schedule"] == ["launch_train"]
    assert cfg["per_rank_ms_per_step"]["max"] == line["ms_per_step"]
    assert line["value"] > 0 and line["scaling"] == "weak"
    assert line["allreduce"]["whole_gradient_alone"]["bytes"] > 0
    reps = [torch.load(str(dump / ("%s_rank%d.pt" % (name, k)))) for k in range(world)]
    assert [d["rank"] for d in reps] == list(range(world)) and all(d["world"] == world for d in reps)
    assert all(d["global_step"] == steps + warmup for d in reps)
    for d in reps[1:]:
        assert torch.equal(d["flat"], reps[0]["flat"])                    # the replicas never drift apart
    # every rank draws its own dropout stream (rank r's utterance b must not share rank 0's noise)
    assert len({d["drop_seed"] for d in reps}) == world
    ref, init = _one_process_on_the_global_batch("rehearsal_keep1", world, steps + warmup)
    moved = (ref - init).abs()
    diff = (reps[0]["flat"] - ref).abs()
    if name == "rehearsal_keep1":
        # 3 Adam steps move every weight by ~3 x lr = 1.2e-3; 8 ranks x B = the 8 B batch up to fp32 summation order
        assert float(moved.max()) > 5e-4
        assert float(diff.max()) < 2e-5 and float(diff.pow(2).mean().sqrt()) < 1e-6, (float(diff.max()),
                                                                                     float(diff.pow(2).mean().sqrt()))
    else:
        assert float(diff.max()) > 1e-4                                   # dropout really was on (and per rank)


# ------------------------------------------------------------------------------------- 8 ranks: bin/nnet-train.py
def _write_sorted_corpus(tmp_path, rng, n, D, V):
    from lstm_ctc_amd.nnet import write_tfrecord
    rows = []
    for i in range(n):
        T = int(rng.integers(12, 31))
        path = str(tmp_path / ("utt%03d.tfrecords" % i))
        write_tfrecord(path, rng.normal(size=(T, D)).astype(np.float32), rng.integers(0, V - 1, size=int(rng.integers(1, 5))))
        rows.append((T, "utt%03d %d %d 1 %s" % (i, T, D, path)))
    rows.sort(key=lambda r_: r_[0])                                      # recipes sort by length
    scp = tmp_path / "tfrecords.scp"
    scp.write_text("\n".join(r_[1] for r_ in rows) + "\n")
    return str(scp)


@pytest.mark.timeout(1800)
def test_nnet_train_eight_ranks_equal_one_process_on_the_8x_batch(tmp_path):
    """bin/nnet-train.py under `torch.distributed.run --nproc-per-node 8` (gloo, ranks sharing GPU 0): rank r trains on
    every 8th batch of 2 utterances (nnet/pipeline.py sharding; ragged lengths, so every rank has its own T_max), the
    gradient is summed before clip / Adam.  One process with --batch-size 16 sees the same 16 utterances per step: same
    tr_loss line, same checkpoint (fp32 summation order apart)."""
    from safetensors.numpy import load_file
    rng = np.random.default_rng(11)
    D, V, world, B = 6, 9, 8, 2
    scp = _write_sorted_corpus(tmp_path, rng, world * B * 3 + 5, D, V)   # 3 global steps + a ragged tail that is dropped
    config = tmp_path / "nnet.config"
    config.write_text("nnet_type = blstm\ninput_dim = %d\nleft_context = 0\nright_context = 0\nnum_layers = 5\n"
                      "num_neurons = 32\nnum_projects = 32\nnum_targets = %d\nuse_peepholes = true\ndropout_rate = 1.0\n" % (D, V))
    d = str(tmp_path)
    py, bindir = sys.executable, os.path.join(ROOT, "bin")
    env1 = _clean_env(LC_LSTM_PERSISTENT="0")
    r = subprocess.run([py, os.path.join(bindir, "nnet-init.py"), "--objective=ctc", "--batch-size", "4", scp, str(config),
                        d + "/nnet.0"], capture_output=True, timeout=600, env=env1)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    common = ["--objective=ctc", "--learn-rate=0.001", "--optimizer=adam", "--seed=1", "--shuffle=false", "--evaluate=true",
              "--report-interval=1"]
    # one process: 16 utterances per step; the trailing 5 utterances form a 4th, smaller batch - drop them from this run's list
    lines = open(scp).read().splitlines()
    scp48 = tmp_path / "first48.scp"
    scp48.write_text("\n".join(lines[:world * B * 3]) + "\n")
    r = subprocess.run([py, os.path.join(bindir, "nnet-train.py")] + common + ["--batch-size", str(world * B), str(scp48),
                       str(config), d + "/nnet.0", d + "/nnet.single"], capture_output=True, timeout=600, env=env1)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    tr1 = [l for l in r.stderr.decode().splitlines() if l.startswith("INFO:tensorflow:tr_loss")]
    env8 = _clean_env(LC_LSTM_PERSISTENT="0", LC_DP_TEST_SHARED_GPU="1")
    cmd = [py, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(bindir, "nnet-train.py")] + common + \
          ["--batch-size", str(B), scp, str(config), d + "/nnet.0", d + "/nnet.dp8"]
    r = subprocess.run(cmd, capture_output=True, timeout=1500, env=env8)
    err = r.stderr.decode()
    assert r.returncode == 0, err[-4000:]
    tr8 = [l for l in err.splitlines() if l.startswith("INFO:tensorflow:tr_loss")]
    assert len(tr1) == 1 and len(tr8) == 1                                # rank 0 alone logs the machine-parsed line
    steps8 = [l for l in err.splitlines() if l.startswith("INFO:tensorflow:step = ")]
    assert len(steps8) == 3                                               # 3 global steps; the ragged tail is dropped
    assert abs(float(tr1[0].split()[-1]) - float(tr8[0].split()[-1])) < 2e-5 * max(1.0, abs(float(tr1[0].split()[-1])))
    a, b = load_file(d + "/nnet.single"), load_file(d + "/nnet.dp8")
    p0 = load_file(d + "/nnet.0")
    assert sorted(a) == sorted(b)
    moved = max(float(np.abs(a[k] - p0[k]).max()) for k in a)
    worst = max(float(np.abs(a[k] - b[k]).max()) for k in a)
    assert moved > 1e-3 and worst < 5e-5, (moved, worst)


# ------------------------------------------------------------------------------------- TF's own test vectors
@pytest.fixture(scope="module")
def tfk():
    with open(os.path.join(HERE, "golden", "tf_host_arith_known_answers.json")) as f:
        return json.load(f)


def test_clip_by_global_norm_tf_vectors(tfk):
    """lc_optimizer_step's clip stage: sgd with lr = 1 from zero parameters leaves -clip(g) (nnet/graph.py:190-192)."""
    from lstm_ctc_amd import ops
    for case in tfk["clip_by_global_norm"]:
        g = np.concatenate([np.asarray(t, np.float32) for t in case["tensors"]])
        want = np.concatenate([np.asarray(t, np.float64) for t in case["expected"]])
        P = torch.zeros(g.size, device="cuda")
        G = torch.from_numpy(g).cuda()
        state, norm = torch.zeros(1, device="cuda"), torch.zeros(2, device="cuda")
        ops.optimizer_step(P, G, g.size, 0.0, case["clip_norm"], "sgd", 1.0, 1, state, norm)
        assert abs(norm[0].item() - case["global_norm"]) < 1e-6, case["name"]
        np.testing.assert_allclose(-P.cpu().numpy().astype(np.float64), want, rtol=2e-7, atol=0, err_msg=case["name"])


def test_adam_tf_vectors(tfk):
    """lc_optimizer_step(adam) over adam_test.py testBasic's three steps (nnet/graph.py:41-42)."""
    from lstm_ctc_amd import ops
    a = tfk["adam"]
    P = torch.tensor(a["var0"] + a["var1"], dtype=torch.float32, device="cuda")
    g = torch.tensor(a["grads0"] + a["grads1"], dtype=torch.float32, device="cuda")
    state, norm = torch.zeros(2 * 4, device="cuda"), torch.zeros(2, device="cuda")
    start = np.asarray(a["var0"] + a["var1"])
    for step in a["steps"]:
        ops.optimizer_step(P, g.clone(), 4, 0.0, 5.0, "adam", a["lr"], step["t"], state, norm)      # norm 0.142 < 5: no clip
        want = np.asarray(step["var0"] + step["var1"])
        got = P.cpu().numpy().astype(np.float64)
        # the movement itself (t x 1e-3), not only the value next to 1..4: float32 spacing at 4.0 is 4.8e-7
        np.testing.assert_allclose(got - start, want - start, atol=6e-7)
    assert abs((P.cpu().numpy()[0] - 1.0) / -3e-3 - 1.0) < 1e-3


@pytest.mark.parametrize("name", ["momentum", "sgd"])
def test_momentum_and_sgd_tf_vectors(tfk, name):
    """lc_optimizer_step(momentum 0.9 / sgd) over momentum_test.py / gradient_descent_test.py testBasic (nnet/graph.py:37-48)."""
    from lstm_ctc_amd import ops
    a = tfk[name]
    P = torch.tensor(a["var0"] + a["var1"], dtype=torch.float32, device="cuda")
    g = torch.tensor(a["grads0"] + a["grads1"], dtype=torch.float32, device="cuda")
    state, norm = torch.zeros(2 * 4, device="cuda"), torch.zeros(2, device="cuda")
    for step in a["steps"]:
        ops.optimizer_step(P, g.clone(), 4, 0.0, 5.0, name, a["lr"], step["t"], state, norm)      # norm 0.142 < 5: no clip
        np.testing.assert_allclose(P.cpu().numpy().astype(np.float64), np.asarray(step["var0"] + step["var1"]), rtol=3e-7)


def test_edit_distance_tf_vectors(tfk):
    from lstm_ctc_amd import ops
    for case in tfk["edit_distance"]:
        hyp, truth = case["hyp"], case["truth"]
        W = max(1, max(len(h) for h in hyp))
        tok = np.zeros((len(hyp), W), np.int32)
        for b, h in enumerate(hyp):
            tok[b, :len(h)] = h
        n = np.asarray([len(h) for h in hyp], np.int32)
        flat = np.asarray([v for t in truth for v in t], np.int32)
        offs = np.concatenate([[0], np.cumsum([len(t) for t in truth])]).astype(np.int32)
        assert ops.edit_distance_host(tok, n, flat, offs).tolist() == case["expected"], case["name"]


# ------------------------------------------------------------------------------------- bf16 recurrences, exact-size buffers
_EXACT_SCRIPT = r'''
import ctypes, sys
import numpy as np
import torch
sys.path.insert(0, %(root)r)
from lstm_ctc_amd import ops
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]

class Exact:
    """A hipMalloc of exactly the tensor's bytes (no caching-allocator slack behind it), seen by torch through
    __cuda_array_interface__."""
    def __init__(self, n):
        p = ctypes.c_void_p()
        assert hip.hipMalloc(ctypes.byref(p), n * 4) == 0
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f4", "data": (p.value, False), "version": 2}

def exact_like(t):
    holder = Exact(t.numel())
    e = torch.as_tensor(holder, device="cuda").view(t.shape)
    e._keep = holder
    e.copy_(t)
    return e

T, B, N = %(T)d, 32, 832                       # T * B * N * 4 bytes = a whole number of 2 MB pages at T = 256
torch.manual_seed(0)
seq = torch.full((B,), T, dtype=torch.int32, device="cuda"); seq[-2:] = T - 3
mk = lambda *s: torch.randn(*s, device="cuda")
res = {}
for exact in (False, True):
    wrap = exact_like if exact else (lambda t: t.clone())
    torch.manual_seed(1)
    fdirs, bdirs = [], []
    for d in range(2):
        zx = mk(T * B, 4 * N) * 0.5
        fdirs.append(dict(zx=wrap(zx), R=mk(N, 4 * N) * 0.03, w_f=mk(N) * 0.1, w_i=mk(N) * 0.1, w_o=mk(N) * 0.1,
                          cs=wrap(torch.zeros(T * B, N, device="cuda")), hs=wrap(torch.zeros(T * B, N, device="cuda")),
                          reverse=d))
    ops.lstm_fwd(fdirs, seq, T, B, N, 1.0, bf16=True)
    kf = ops.last_lstm_schedule()["kind"]
    for d in range(2):
        f = fdirs[d]
        bdirs.append(dict(gates=f["zx"], RT=mk(4 * N, N) * 0.03, w_f=f["w_f"], w_i=f["w_i"], w_o=f["w_o"], cs=f["cs"],
                          dh=wrap(mk(T * B, N) * 0.1), dpeep=torch.zeros(3, N, device="cuda"),
                          dbias=torch.zeros(4 * N, device="cuda"), reverse=d))
    ops.lstm_bwd(bdirs, seq, T, B, N, bf16=True)
    kb = ops.last_lstm_schedule()["kind"]
    torch.cuda.synchronize()
    assert int(ops.lstm_status("cuda").item()) == 0
    res[exact] = [t.clone() for d in range(2) for t in (fdirs[d]["hs"], fdirs[d]["cs"], bdirs[d]["gates"], bdirs[d]["dbias"])]
    assert kf == kb == "persistent_bf16", (kf, kb)
for a, b in zip(res[False], res[True]):
    assert torch.isfinite(a).all() and torch.equal(a, b)
print("EXACT-OK")
'''


@pytest.mark.parametrize("T", [8, 256])
def test_bf16_recurrences_partial_workgroup_on_exact_size_buffers(T):
    """N = 832 (26 units left for the last workgroup of 32: a partial pair block) on hipMalloc'd tensors of exactly
    T * B * N * 4 bytes - at T = 256 a whole number of 2 MB pages, so a read 4 bytes past the last row leaves the mapping.
    In a child process: a memory fault ends the process, not the test session.  Results must be bit-identical to the same
    run on ordinary (caching-allocator) tensors."""
    r = subprocess.run([sys.executable, "-c", _EXACT_SCRIPT % {"root": ROOT, "T": T}], capture_output=True, timeout=900,
                       env=_clean_env(), cwd=ROOT)
    assert r.returncode == 0 and b"EXACT-OK" in r.stdout, (r.stdout.decode()[-1000:], r.stderr.decode()[-3000:])


# ------------------------------------------------------------------------------------- lc_gemm_bf16_nt2
@pytest.mark.parametrize("M,N,K1,K2", [(512, 512, 256, 128), (768, 256, 64, 192), (600, 300, 128, 128), (100, 64, 64, 32),
                                       (2048, 1024, 512, 512)])
def test_gemm_bf16_nt2_is_the_sum_of_the_two_products(M, N, K1, K2):
    """C = alpha (A1 B1^T + A2 B2^T) + beta C + bias in one pass over C (whole 256-tiles: one kernel walking both operand
    pairs; ragged edges and small shapes: strips / the two products in sequence) against float64 on the SAME bf16 operands,
    and against the two separate lc_gemm_bf16_nt calls it replaces (same products, another fp32 summation order); with the
    fused epilogue (DropoutWrapper mask + bf16 shadow) bit-identical to the separate passes on ITS OWN result."""
    from lstm_ctc_amd import ops
    g = torch.Generator(device="cuda").manual_seed(M + N + K1)
    Kw = max(K1, K2)                                       # both pairs share lda / ldb: column windows of equally wide buffers
    mk = lambda r, c: torch.randn(r, Kw, device="cuda", generator=g).to(torch.bfloat16)[:, :c]
    A1, B1, A2, B2 = mk(M, K1), mk(N, K1), mk(M, K2), mk(N, K2)
    bias = torch.randn(N, device="cuda", generator=g)
    C0 = torch.randn(M, N, device="cuda", generator=g)
    ref = 0.5 * (A1.double() @ B1.double().t() + A2.double() @ B2.double().t()) + 2.0 * C0.double() + bias.double()
    got = C0.clone()
    ops.gemm_bf16_nt2(A1, B1, A2, B2, out=got, alpha=0.5, beta=2.0, bias=bias)
    two = C0.clone()
    ops.gemm_bf16_nt(A1, B1, out=two, alpha=0.5, beta=2.0, bias=bias, K=K1)
    ops.gemm_bf16_nt(A2, B2, out=two, alpha=0.5, beta=1.0, K=K2)
    scale = float(ref.abs().max())
    assert float((got.double() - ref).abs().max()) < 2e-6 * scale * max(1.0, ((K1 + K2) / 256) ** 0.5)
    assert float((got - two).abs().max()) < 4e-6 * scale
    # fused epilogue on the final value
    keep, seed, stream0, P = 0.8, 4321, 3, N // 2
    sh = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
    fused = C0.clone()
    ops.gemm_bf16_nt2(A1, B1, A2, B2, out=fused, alpha=0.5, beta=2.0, bias=bias, epilogue=ops.Epilogue(keep, seed, stream0, P, sh))
    want = got.clone()
    for d in range(2):
        ops.dropout_scale(want[:, d * P:(d + 1) * P], keep, seed, stream0 + d)
    assert torch.equal(fused, want)
    assert torch.equal(sh, want.to(torch.bfloat16))
    # one-shot: the next product is plain again
    again = C0.clone()
    ops.gemm_bf16_nt2(A1, B1, A2, B2, out=again, alpha=0.5, beta=2.0, bias=bias)
    assert torch.equal(again, got)


def test_c5_backward_with_and_without_the_fused_dx(monkeypatch):
    """Model.backward in bf16 mode takes ONE dX product per bidirectional layer (lc_gemm_bf16_nt2); LC_FUSE_DX=0 is the two
    products it replaces.  Same operands, another summation order (and bf16 re-rounding of what is built on it): gradients within 5e-4 of the largest entry."""
    from lstm_ctc_amd import ops
    from lstm_ctc_amd.nnet.model import Model
    cfg = dict(nnet_type="blstm", input_dim=40, left_context=0, right_context=0, num_layers=3, num_neurons=256,
               num_projects=256, num_targets=20, use_peepholes=True, dropout_rate=0.9, compute_dtype="bf16")
    T, B = 16, 32                                          # T * B = 512 rows: whole 256-tiles, the one-kernel route
    g = torch.Generator().manual_seed(5)
    x = torch.randn(T, B, 40, generator=g).cuda()
    seq = torch.full((B,), T, dtype=torch.int32).cuda()
    labels = torch.randint(0, 19, (B * 4,), generator=g, dtype=torch.int32).cuda()
    offs = (torch.arange(B + 1) * 4).to(torch.int32).cuda()
    grads = {}
    for fuse in ("1", "0"):
        monkeypatch.setenv("LC_FUSE_DX", fuse)
        model = Model(cfg, "cuda", seed=3)
        assert model.fuse_dx == (fuse == "1")
        ops.PROFILE = []
        try:
            logits = model.forward(x, seq, drop_seed=11)
            _, grad = ops.ctc_loss(logits, labels, offs, seq, 4)
            model.backward(grad)
            torch.cuda.synchronize()
            n_products = sum(1 for k, _, _, _ in ops.PROFILE if k.startswith("gemm"))
        finally:
            ops.PROFILE = None
        grads[fuse] = (model.ps.grad.clone(), n_products)
    assert grads["1"][1] == grads["0"][1] - 2              # layers 2 and 1: one dX product instead of two
    a, b = grads["1"][0], grads["0"][0]
    # (measured 8e-5: the fp32 dinp differs in its last bits, and where that flips a bf16 rounding of the next layer's operands
    # the difference is one bf16 ulp of an operand, not one fp32 ulp)
    assert torch.isfinite(a).all() and float((a - b).abs().max()) < 5e-4 * float(b.abs().max())


@pytest.mark.parametrize("N,want", [(256, False), (320, False), (384, False), (448, False), (512, True)])
def test_x3_forward_width_rule_by_behaviour(monkeypatch, N, want):
    """bf16x3 mode: the x3 flag Model.forward hands to ops.lstm_fwd (the rule used to be tested by grepping
    model.py's source), and the schedule the library then reports."""
    from lstm_ctc_amd import ops
    from lstm_ctc_amd.nnet import model as model_mod
    seen = []
    real = ops.lstm_fwd

    def spy(dirs, seq_len, T, B, N_, fb, bf16=False, x3=False):
        seen.append((N_, bool(x3)))
        return real(dirs, seq_len, T, B, N_, fb, bf16=bf16, x3=x3)

    monkeypatch.setattr(ops, "lstm_fwd", spy)
    cfg = dict(nnet_type="blstm", input_dim=40, left_context=0, right_context=0, num_layers=1, num_neurons=N,
               num_projects=N, num_targets=12, use_peepholes=True, dropout_rate=1.0, compute_dtype="bf16x3")
    m = model_mod.Model(cfg, "cuda", seed=2)
    T, B = 8, 16
    x = torch.randn(T, B, 40, device="cuda")
    m.forward(x, torch.full((B,), T, dtype=torch.int32, device="cuda"))
    torch.cuda.synchronize()
    assert seen == [(N, want)]
    assert ops.last_lstm_schedule()["kind"] == ("persistent_x3" if want else "persistent_f32")


@pytest.mark.parametrize("keep", [0.9, 1.0])
def test_c5_shadow_only_recurrences_are_bit_identical(monkeypatch, keep):
    """c5 (bf16 operands, N = 1024, rows >= 4096): the recurrences store hs / dz only as the bf16 shadows that every product of
    the step reads (`shadow_only`; the fp32 copies stay unwritten).  Nothing may change: logits, encoder states, every gradient
    bit for bit against LC_C5_SHADOW_ONLY=0; and the fp32 buffers really are untouched (a sentinel survives)."""
    from lstm_ctc_amd import ops
    from lstm_ctc_amd.nnet.model import Model
    cfg = dict(nnet_type="blstm", input_dim=40, left_context=0, right_context=0, num_layers=2, num_neurons=1024,
               num_projects=1024, num_targets=44, use_peepholes=True, dropout_rate=keep, compute_dtype="bf16")
    T, B = 64, 64                                          # (keep = 1: inference / parity runs - the projection's epilogue still carries the shadow)
    g = torch.Generator().manual_seed(9)
    x = torch.randn(T, B, 40, generator=g).cuda()
    seq = torch.full((B,), T, dtype=torch.int32)
    seq[:5] = torch.tensor([T - 9, T - 5, T - 3, T - 1, T - 1], dtype=torch.int32)
    seq = seq.cuda()
    labels = torch.randint(0, 43, (B * 6,), generator=g, dtype=torch.int32).cuda()
    offs = (torch.arange(B + 1) * 6).to(torch.int32).cuda()
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("LC_C5_SHADOW_ONLY", mode)
        model = Model(cfg, "cuda", seed=3)
        assert model.shadow_only == (mode == "1")
        logits = model.forward(x, seq, drop_seed=7)
        assert ops.last_lstm_schedule()["kind"] == "persistent_bf16"
        flags = [bool(dd.get("shadow_only")) for L in model.saved["layers"] for dd in L["dirs"]]
        assert flags == [mode == "1"] * 4
        enc = model.encoder().clone()
        _, grad = ops.ctc_loss(logits, labels, offs, seq, 6)
        model.backward(grad)
        assert ops.last_lstm_schedule()["kind"] == "persistent_bf16"
        torch.cuda.synchronize()
        assert int(ops.lstm_status("cuda").item()) == 0
        res[mode] = (logits.clone(), enc, model.ps.grad.clone())
    for a, b in zip(res["1"], res["0"]):
        assert torch.isfinite(a).all() and torch.equal(a, b)
    # the kernel-level contract: with shadow_only the fp32 hs / dz keep whatever they held
    N = 1024
    rows = T * B
    mk = lambda *s: torch.randn(*s, device="cuda")
    dirs = [dict(zx=mk(rows, 4 * N) * 0.3, R=mk(N, 4 * N) * 0.02, w_f=mk(N) * 0.1, w_i=mk(N) * 0.1, w_o=mk(N) * 0.1,
                 cs=torch.zeros(rows, N, device="cuda"), hs=torch.full((rows, N), 123.0, device="cuda"),
                 hs_bf16=torch.zeros(rows, N, dtype=torch.bfloat16, device="cuda"), shadow_only=True, reverse=d) for d in range(2)]
    ops.lstm_fwd(dirs, seq, T, B, N, 1.0, bf16=True)
    torch.cuda.synchronize()
    for dd in dirs:
        assert bool((dd["hs"] == 123.0).all()) and float(dd["hs_bf16"].float().abs().max()) > 0


@pytest.mark.parametrize("form", ["f32", "nt", "nn", "tn", "nt2"])
@pytest.mark.parametrize("M,N,K", [(512, 512, 128), (700, 392, 192)])
def test_gemm_epilogue_shadow_only_leaves_c_untouched(form, M, N, K):
    """lc_gemm_epilogue_t.shadow_only: the product writes ONLY the bf16 shadow of its (masked) result - bit-identical to the
    shadow of the ordinary epilogue - and the float32 C keeps what it held (whole tiles and ragged strips, every product form;
    lc_gemm_bf16_nt2's ragged edges, which run two products in sequence, keep a partial sum there)."""
    from lstm_ctc_amd import ops
    if form in ("nn", "tn") and (M % 256 or N % 256):
        pytest.skip("K-major kernels take whole 256-tiles only")
    g = torch.Generator(device="cuda").manual_seed(M + N)
    A = torch.randn(M, K, device="cuda", generator=g)
    B = torch.randn(K, N, device="cuda", generator=g)
    a16, b16t, b16 = A.to(torch.bfloat16), B.t().contiguous().to(torch.bfloat16), B.to(torch.bfloat16)
    at16 = A.t().contiguous().to(torch.bfloat16)
    bias = torch.randn(N, device="cuda", generator=g)

    def product(out, ep):
        if form == "f32":
            ops.gemm(A, B, out=out, bias=bias, epilogue=ep)
        elif form == "nt":
            ops.gemm_bf16_nt(a16, b16t, out=out, bias=bias, epilogue=ep)
        elif form == "nn":
            ops.gemm_bf16_nn(a16, b16, out=out, bias=bias, epilogue=ep)
        elif form == "tn":
            ops.gemm_bf16_tn(at16, b16, out=out, bias=bias, epilogue=ep)
        else:
            ops.gemm_bf16_nt2(a16, b16t, a16, b16t, out=out, bias=bias, epilogue=ep)

    keep, seed, P = 0.75, 99, N // 2
    full = torch.zeros(M, N, device="cuda")
    sh_full = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
    product(full, ops.Epilogue(keep, seed, 4, P, sh_full))
    only = torch.full((M, N), 7.0, device="cuda")
    sh_only = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
    product(only, ops.Epilogue(keep, seed, 4, P, sh_only, shadow_only=True))
    if form == "nt2" and (M % 256 or N % 256):
        # ragged edges run the two products in sequence: the strips of C hold the first one's partial sum (C is "unspecified")
        assert bool((only[:M // 256 * 256, :N // 256 * 256] == 7.0).all())
    else:
        assert bool((only == 7.0).all())
    assert torch.equal(sh_only, sh_full) and torch.equal(sh_full, full.to(torch.bfloat16))
    assert 0.2 < float((sh_only == 0).float().mean()) < 0.3
    again = torch.zeros(M, N, device="cuda")                 # one-shot: the next product writes C again
    product(again, None)
    assert float(again.abs().max()) > 0


@pytest.mark.parametrize("kind", ["f32_nn", "f32_nt", "bf16_nt"])
def test_gemm_whole_round_tail_split(kind):
    """An unsplit product whose 256 x 256 tiles leave the last round of 256 CUs mostly empty (here 1056 tiles = 4.125 rounds)
    runs whole rounds on the big kernel and the rows behind them on the 128 x 128 kernel (LC_GEMM_TAIL / option gemm_tail):
    same result as the one-kernel route up to summation order, against float64; with the fused epilogue bit-identical masks."""
    from lstm_ctc_amd import ops
    M, N, K = 256 * 66, 4096, 128
    g = torch.Generator(device="cuda").manual_seed(1)
    A = torch.randn(M, K, device="cuda", generator=g)
    B = torch.randn(K, N, device="cuda", generator=g)
    Bt = B.t().contiguous()
    bias = torch.randn(N, device="cuda", generator=g)
    a16, bt16 = A.to(torch.bfloat16), Bt.to(torch.bfloat16)

    def product(ep=None):
        if kind == "f32_nn":
            return ops.gemm(A, B, bias=bias, epilogue=ep)
        if kind == "f32_nt":
            return ops.gemm(A, Bt, tb=True, bias=bias, epilogue=ep)
        return ops.gemm_bf16_nt(a16, bt16, bias=bias, epilogue=ep)

    ref = ((a16.double() @ bt16.double().t()) if kind == "bf16_nt" else (A.double() @ B.double())) + bias.double()
    out = {}
    for mode in (0, 1):
        ops.set_option("gemm_tail", mode)
        try:
            out[mode] = product()
            sh = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
            masked = product(ops.Epilogue(0.8, 77, 2, N // 2, sh))
        finally:
            ops.set_option("gemm_tail", None)
        assert float((out[mode].double() - ref).abs().max()) < 3e-6 * float(ref.abs().max())
        want = out[mode].clone()
        for d in range(2):
            ops.dropout_scale(want[:, d * (N // 2):(d + 1) * (N // 2)], 0.8, 77, 2 + d)
        assert torch.equal(masked, want) and torch.equal(sh, want.to(torch.bfloat16))
    if kind == "bf16_nt":                 # the rows of the whole rounds: the same kernel, same bits (fp32: 4.125 rounds do not
        assert torch.equal(out[0][:256 * 64], out[1][:256 * 64])         # "fill" the big kernel unsplit - it is all 128-tiles there)
    assert float((out[0] - out[1]).abs().max()) < 3e-6 * float(ref.abs().max())


@pytest.mark.parametrize("M,N,K1,K2", [(512, 512, 256, 128), (768, 256, 64, 192), (600, 300, 128, 128), (100, 64, 64, 32)])
@pytest.mark.parametrize("big", ["2", "1"])
def test_gemm_f32_nt2_is_the_sum_of_the_two_products(monkeypatch, M, N, K1, K2, big):
    """lc_gemm_f32_nt2 (float32 twin of lc_gemm_bf16_nt2): one kernel walking both operand pairs on whole 256-tiles
    (LC_GEMM_F32_BIG=2 takes that route at test sizes), strips / sequence elsewhere - against float64 and against the two
    lc_gemm_f32 calls it replaces; the fused epilogue bit-identical to the separate passes on its own result."""
    from lstm_ctc_amd import ops
    monkeypatch.setenv("LC_GEMM_F32_BIG", big)
    g = torch.Generator(device="cuda").manual_seed(M + N + K1)
    Kw = max(K1, K2)
    mk = lambda r, c: torch.randn(r, Kw, device="cuda", generator=g)[:, :c]
    A1, B1, A2, B2 = mk(M, K1), mk(N, K1), mk(M, K2), mk(N, K2)
    bias = torch.randn(N, device="cuda", generator=g)
    C0 = torch.randn(M, N, device="cuda", generator=g)
    ref = 0.5 * (A1.double() @ B1.double().t() + A2.double() @ B2.double().t()) + 2.0 * C0.double() + bias.double()
    got = C0.clone()
    ops.gemm_nt2(A1, B1, A2, B2, out=got, alpha=0.5, beta=2.0, bias=bias)
    two = C0.clone()
    ops.gemm(A1, B1, tb=True, out=two, alpha=0.5, beta=2.0, bias=bias)
    ops.gemm(A2, B2, tb=True, out=two, alpha=0.5, beta=1.0)
    scale = float(ref.abs().max())
    assert float((got.double() - ref).abs().max()) < 2e-6 * scale
    assert float((got - two).abs().max()) < 4e-6 * scale
    keep, seed, stream0, P = 0.8, 4321, 3, N // 2
    sh = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
    fused = C0.clone()
    ops.gemm_nt2(A1, B1, A2, B2, out=fused, alpha=0.5, beta=2.0, bias=bias, epilogue=ops.Epilogue(keep, seed, stream0, P, sh))
    want = got.clone()
    for d in range(2):
        ops.dropout_scale(want[:, d * P:(d + 1) * P], keep, seed, stream0 + d)
    assert torch.equal(fused, want) and torch.equal(sh, want.to(torch.bfloat16))
    again = C0.clone()
    ops.gemm_nt2(A1, B1, A2, B2, out=again, alpha=0.5, beta=2.0, bias=bias)
    assert torch.equal(again, got)


def test_c4_backward_with_and_without_the_fused_dx(monkeypatch):
    """fp32 Model.backward takes ONE dX product per bidirectional layer (lc_gemm_f32_nt2); LC_FUSE_DX=0 is the two products
    it replaces: same operands, another fp32 summation order - every gradient within 1e-5 of the largest entry."""
    from lstm_ctc_amd import ops
    from lstm_ctc_amd.nnet.model import Model
    monkeypatch.setenv("LC_GEMM_F32_BIG", "2")
    cfg = dict(nnet_type="blstm", input_dim=40, left_context=0, right_context=0, num_layers=3, num_neurons=1024,
               num_projects=256, num_targets=20, use_peepholes=True, dropout_rate=0.9)
    T, B = 16, 32
    g = torch.Generator().manual_seed(5)
    x = torch.randn(T, B, 40, generator=g).cuda()
    seq = torch.full((B,), T, dtype=torch.int32).cuda()
    labels = torch.randint(0, 19, (B * 4,), generator=g, dtype=torch.int32).cuda()
    offs = (torch.arange(B + 1) * 4).to(torch.int32).cuda()
    grads = {}
    for fuse in ("1", "0"):
        monkeypatch.setenv("LC_FUSE_DX", fuse)
        model = Model(cfg, "cuda", seed=3)
        assert not model.overlap_wgrad                       # N = 1024: the wide models' backward (no side stream)
        ops.PROFILE = []
        try:
            logits = model.forward(x, seq, drop_seed=11)
            _, grad = ops.ctc_loss(logits, labels, offs, seq, 4)
            model.backward(grad)
            torch.cuda.synchronize()
            n_products = sum(1 for k, _, _, _ in ops.PROFILE if k.startswith("gemm"))
        finally:
            ops.PROFILE = None
        grads[fuse] = (model.ps.grad.clone(), n_products)
    assert grads["1"][1] == grads["0"][1] - 2
    a, b = grads["1"][0], grads["0"][0]
    assert torch.isfinite(a).all() and float((a - b).abs().max()) < 1e-5 * float(b.abs().max())
