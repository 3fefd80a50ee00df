"""Data-parallel host logic on CPU: world_size-2 gloo processes (no GPU).

The product's exchange helpers (lstm_ctc_amd.nnet.dp) are exercised with the CPU oracle standing in for the
HIP kernels: each rank computes the gradient of its utterance shard, the flat gradient is all-reduced, then
L2 + clip + Adam run on the sum.  The result must equal one process on the full batch — the property the
north star's weak-scaling recipe relies on — and the logged triple must sum across ranks."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cfg():
    return dict(nnet_type="blstm", input_dim=6, left_context=0, right_context=0, num_layers=2, num_neurons=16,
                num_projects=8, num_targets=7, use_peepholes=True, dropout_rate=1.0)


def _data():
    rng = np.random.default_rng(0)
    B, T = 6, 9
    seq = np.array([9, 8, 8, 7, 6, 5], np.int32)
    x = rng.normal(size=(B, T, 6)).astype(np.float64)
    for b in range(B):
        x[b, seq[b]:] = 0
    labels = np.full((B, 3), -1, np.int64)
    for b in range(B):
        n = rng.integers(1, 4)
        labels[b, :n] = rng.integers(0, 6, size=n)
    return x, seq, labels


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lstm_ctc_amd.nnet import dp
    from oracle import oracle as orc
    cfg = _cfg()
    params = orc.init_params(cfg, seed=5, dtype=np.float64)
    x, seq, labels = _data()
    sl = slice(rank, None, world)                                   # rank r takes every world-th utterance
    out = orc.validation_graph(params, cfg, x[sl], seq[sl], labels[sl], want_grad=True)
    grads, _ = orc.backward(params, cfg, out["saved"], np.ascontiguousarray(out["dlogits"]))
    names = sorted(grads)
    flat = torch.from_numpy(np.concatenate([grads[k].reshape(-1) for k in names]))
    dp.allreduce_sum_(flat, dist.group.WORLD)
    size, eloss, ev = dp.reduce_triple(out["size"], out["eval_loss"], out["eval"], dist.group.WORLD, "cpu")
    # replicated parameters: broadcast must be a no-op on identical buffers and fix a diverged one
    p0 = torch.from_numpy(params[names[0]].reshape(-1).copy())
    if rank == 1:
        p0 += 1.0
    dp.broadcast_(p0, dist.group.WORLD, src=0)
    if rank == 0:
        np.savez(out_path, flat=flat.numpy(), size=size, eloss=eloss, ev=ev, p0=p0.numpy())
    else:
        np.savez(out_path + ".r1.npz", p0=p0.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_sum_equals_full_batch(tmp_path, oracle):
    port = 29500 + (os.getpid() % 2000)
    out_path = str(tmp_path / "dp.npz")
    mp.spawn(_worker, args=(2, port, out_path), nprocs=2, join=True)
    got = np.load(out_path)
    cfg = _cfg()
    params = oracle.init_params(cfg, seed=5, dtype=np.float64)
    x, seq, labels = _data()
    full = oracle.validation_graph(params, cfg, x, seq, labels, want_grad=True)
    grads, _ = oracle.backward(params, cfg, full["saved"], np.ascontiguousarray(full["dlogits"]))
    names = sorted(grads)
    ref = np.concatenate([grads[k].reshape(-1) for k in names])
    np.testing.assert_allclose(got["flat"], ref, rtol=1e-10, atol=1e-12)
    assert int(got["size"]) == full["size"]
    assert abs(float(got["eloss"]) - full["eval_loss"]) < 1e-9
    assert float(got["ev"]) == full["eval"]
    # clip + update on the summed gradient == single-process update (clip acts on the TOTAL gradient)
    g_sum = {}
    off = 0
    for k in names:
        n = grads[k].size
        g_sum[k] = got["flat"][off:off + n].reshape(grads[k].shape)
        off += n
    a, na = oracle.l2_and_clip(params, g_sum, 5.0, 1e-5)
    b, nb = oracle.l2_and_clip(params, grads, 5.0, 1e-5)
    assert abs(na - nb) < 1e-9
    r1 = np.load(out_path + ".r1.npz")
    np.testing.assert_array_equal(got["p0"], r1["p0"])


def test_pipeline_sharding_is_disjoint_and_even(tmp_path):
    from lstm_ctc_amd.nnet import write_tfrecord, dataset_from_tfrecords, create_pipeline_sequence_batch
    rng = np.random.default_rng(1)
    lines = []
    for i in range(11):
        T = 4 + i
        path = str(tmp_path / ("u%02d.tfrecords" % i))
        write_tfrecord(path, rng.normal(size=(T, 3)).astype(np.float32), rng.integers(0, 4, size=2))
        lines.append("u%02d %d 3 1 %s" % (i, T, path))
    scp = tmp_path / "t.scp"
    scp.write_text("\n".join(lines) + "\n")
    _, ds, dim = dataset_from_tfrecords(str(scp))
    seen = []
    steps = []
    for r in range(2):
        _, pipe = create_pipeline_sequence_batch(ds, dim, batch_size=2, rank=r, world_size=2)
        batches = list(pipe)
        steps.append(len(batches))
        seen.append(sorted(int(t) for b in batches for t in b["sequence_length"]))
    assert steps[0] == steps[1] == 3                # 6 batches (the last one short) -> 3 per rank
    assert not set(seen[0]) & set(seen[1])
    assert sorted(seen[0] + seen[1]) == list(range(4, 15))
    _, single = create_pipeline_sequence_batch(ds, dim, batch_size=2)
    assert sum(len(b["sequence_length"]) for b in single) == 11     # one process sees everything, last batch smaller


def _bucket_worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lstm_ctc_amd.nnet import dp
    rng = np.random.default_rng(100 + rank)
    flat = torch.from_numpy(rng.normal(size=1000))
    whole = flat.clone()
    dp.allreduce_sum_(whole, dist.group.WORLD)
    b = dp.GradientBuckets(flat, dist.group.WORLD)
    b.issue(600, 800)                     # the order the backward hands the layers over: top layer first
    b.wait()
    b.issue(300, 600)
    b.issue(0, 0)                         # an empty range is ignored
    b.wait()
    b.finish()                            # [0, 300) and [800, 1000): everything not reduced yet, exactly once
    overlap = False
    try:
        b2 = dp.GradientBuckets(flat.clone(), dist.group.WORLD)
        b2.issue(10, 20)
        b2.issue(15, 30)
    except AssertionError:
        overlap = True
    # every rank must hold the SAME bits (the replicas apply the same update): compare with rank 0's buffer
    mine = flat.clone()
    dist.broadcast(mine, src=0)
    same = torch.tensor([int(torch.equal(mine, flat))])
    dist.all_reduce(same, op=dist.ReduceOp.MIN)
    if rank == 0:
        np.savez(out_path, flat=flat.numpy(), whole=whole.numpy(), overlap=overlap, same=int(same.item()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_gradient_buckets_reduce_every_element_exactly_once(tmp_path, world):
    """dp.GradientBuckets (per-layer all-reduces issued during the backward + the remainder at the end) gives the same
    buffer as one all-reduce of the whole flat gradient, and refuses overlapping ranges.  world = 8: the node's rank count."""
    port = 31500 + (os.getpid() % 2000) + world
    out_path = str(tmp_path / "buckets.npz")
    mp.spawn(_bucket_worker, args=(world, port, out_path), nprocs=world, join=True)
    got = np.load(out_path)
    if world == 2:
        np.testing.assert_array_equal(got["flat"], got["whole"])
    else:       # a ring's summation order depends on where an element falls in the chunking of ITS all-reduce: last-ulp
        np.testing.assert_allclose(got["flat"], got["whole"], rtol=1e-12, atol=1e-14)
    assert bool(got["overlap"]) and int(got["same"]) == 1


def test_pipeline_sharding_eight_ranks(tmp_path):
    """Rank r of 8 takes every 8th batch: equal step counts on every rank, disjoint utterances,
    global step k = batches 8k .. 8k+7 = the 8 B consecutive utterances one process with batch 8 B would see; the ragged
    tail (fewer than 8 batches) is dropped on every rank alike."""
    from lstm_ctc_amd.nnet import write_tfrecord, dataset_from_tfrecords, create_pipeline_sequence_batch
    rng = np.random.default_rng(2)
    lines = []
    n, B, world = 8 * 2 * 2 + 7, 2, 8                    # 2 global steps + 7 utterances (3.5 batches) of tail
    for i in range(n):
        T = 3 + i
        path = str(tmp_path / ("u%02d.tfrecords" % i))
        write_tfrecord(path, rng.normal(size=(T, 3)).astype(np.float32), rng.integers(0, 4, size=2))
        lines.append("u%02d %d 3 1 %s" % (i, T, path))
    scp = tmp_path / "t.scp"
    scp.write_text("\n".join(lines) + "\n")
    _, ds, dim = dataset_from_tfrecords(str(scp))
    per_rank = []
    for r in range(world):
        _, pipe = create_pipeline_sequence_batch(ds, dim, batch_size=B, rank=r, world_size=world)
        per_rank.append([sorted(int(t) for t in b["sequence_length"]) for b in pipe])
    assert all(len(p) == 2 for p in per_rank)
    for k in range(2):                                   # global step k: utterances [16 k, 16 k + 16), lengths 3 + index
        seen = sorted(t for p in per_rank for t in p[k])
        assert seen == [3 + i for i in range(16 * k, 16 * k + 16)]
        for r in range(world):
            assert per_rank[r][k] == [3 + 16 * k + 2 * r, 3 + 16 * k + 2 * r + 1]
