#!/usr/bin/env python3
"""bench.py — the headline benchmark: acoustic frames/sec of full BiLSTM-CTC train steps on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload (BASELINE.json metric, config c4): 5 x BiLSTM-1024 (P = N, peepholes) + affine head, V = 44,
synthetic 40-dim fbank, T = 1000, B = 64 utterances per GPU, L = 100 labels, fp32, adam lr 4e-4,
clip 5, L2 1e-5, output dropout keep 0.9.  One "step" = forward + CTC loss/gradient + BPTT backward +
(N > 1) RCCL all-reduce of the flat fp32 gradient + L2/clip/Adam update, inputs resident in HBM.
frames = sum_b sequence_length_b; value = frames of all ranks / max-over-ranks time (weak scaling).

Extra objects on the JSON line:
  roofline      dominant kernel (the f32 MFMA GEMM): algorithmic FLOPs / HIP-event time, measured live on the
                launch stream over the timed steps, against the 157.3 TFLOP/s fp32 matrix peak.
  roofline_ctc  the CTC op (row stats + alpha/beta scan + gradient): algorithmic bytes T*B*(8V+8S) / time,
                against the 8 TB/s HBM peak (north-star target: >= 40 %).
  cpu_baseline  the CPU oracle (restatement of the TF-1.8 graph; TF itself is not installable) timed on this
                box's host cores on a bounded sample of the same model, rank 0 at N = 1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# multi-process RCCL on this pool needs dmabuf IPC (the image exports this already; keep it if launched bare)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402

WORKLOADS = {
    "c4": dict(desc="c4: 5xBiLSTM-1024 (P=N, peepholes) CTC, V=44, synthetic 40-d fbank T=1000 B=64/GPU L=100, fp32",
               cfg=dict(nnet_type="blstm", input_dim=40, left_context=0, right_context=0, num_layers=5,
                        num_neurons=1024, num_projects=1024, num_targets=44, use_peepholes=True, dropout_rate=0.9),
               B=64, T=1000, L=100),
    "c2": dict(desc="c2: 3xBiLSTM-320 (P=N, peepholes) CTC, V=72, synthetic 40-d fbank T=1000 B=32/GPU L=100, fp32",
               cfg=dict(nnet_type="blstm", input_dim=40, left_context=0, right_context=0, num_layers=3,
                        num_neurons=320, num_projects=320, num_targets=72, use_peepholes=True, dropout_rate=0.9),
               B=32, T=1000, L=100),
    "c3": dict(desc="c3: 5xBiLSTM-512 + high-rank MoE head (E=V=72, tau=10), T=1000 B=32/GPU L=100, fp32",
               cfg=dict(nnet_type="blstm", input_dim=40, left_context=0, right_context=0, num_layers=5,
                        num_neurons=512, num_projects=512, num_targets=72, use_peepholes=True, dropout_rate=0.9,
                        num_experts=72, moe_temp=10.0),
               B=32, T=1000, L=100),
    "c1": dict(desc="c1: 1 x uniLSTM-256 (P=N, peepholes, forget_bias 1) phn-CTC, V=72, synthetic 40-d fbank T=1000 B=32/GPU "
                    "L=100, fp32 (BASELINE configs[0] is the reference's CPU plumbing case; here on the GPU)",
               cfg=dict(nnet_type="lstm", input_dim=40, left_context=0, right_context=0, num_layers=1,
                        num_neurons=256, num_projects=256, num_targets=72, dropout_rate=0.9),
               B=32, T=1000, L=100),
    "c5": dict(desc="c5: c4 with bf16 GEMM operands (bf16 MFMA gate GEMMs + recurrence, fp32 accumulate/state/CTC/"
                    "optimizer): 5xBiLSTM-1024, V=44, T=1000 B=64/GPU L=100",
               cfg=dict(nnet_type="blstm", input_dim=40, left_context=0, right_context=0, num_layers=5,
                        num_neurons=1024, num_projects=1024, num_targets=44, use_peepholes=True, dropout_rate=0.9,
                        compute_dtype="bf16"),
               B=64, T=1000, L=100),
}
PEAK_F32_MFMA_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md chip table
PEAK_BF16_MFMA_TFLOPS = 2500.0   # dense bf16 (same table)
PEAK_HBM_GBS = 8000.0


def measured_traffic(workload, kernel):
    """HBM-side bytes per launch of `kernel` ("gemm" / "ctc") from the rocprofv3 PMC passes of THIS round
    (tools/pmc_traffic.py writes profiles/r2_pmc_traffic.json: FETCH_SIZE corrected x2 for wide reads + WRITE_SIZE, as
    /opt/skills/guides/MI355X_MICROARCH.md prescribes).  PMC counters cannot be collected inside a timed run, so the
    figure belongs to the profiled run of the same command; None when no such file covers this workload."""
    try:
        with open(os.path.join(ROOT, "profiles", "r2_pmc_traffic.json")) as f:
            return json.load(f).get(workload, {}).get(kernel)
    except (OSError, ValueError):
        return None


def synth_batch(w, rank, device):
    """SURVEY.md §8d throughput set: x ~ N(0,1), all T_b = T, L_b = L, labels uniform on {0..V-2}."""
    g = torch.Generator().manual_seed(777 + rank)
    B, T, L, V, D = w["B"], w["T"], w["L"], w["cfg"]["num_targets"], w["cfg"]["input_dim"]
    x = torch.randn((T, B, D), generator=g, dtype=torch.float32)
    labels = torch.randint(0, V - 1, (B * L,), generator=g, dtype=torch.int32)
    offs = (torch.arange(B + 1, dtype=torch.int64) * L).to(torch.int32)
    seq = torch.full((B,), T, dtype=torch.int32)
    return x.to(device), seq.to(device), labels.to(device), offs.to(device)


def ctc_large_batch(w, device, B=512):
    """The CTC op alone on B utterances of the workload's shape (all local micro-batches scanned in one launch): where
    the scan is no longer bound by the T-long chain of one utterance but by issue / HBM.  Outside the timed region."""
    from lstm_ctc_amd import ops
    T, L, V = w["T"], w["L"], w["cfg"]["num_targets"]
    g = torch.Generator().manual_seed(5)
    logits = torch.randn((T, B, V), generator=g).to(device)
    labels = torch.randint(0, V - 1, (B * L,), generator=g, dtype=torch.int32).to(device)
    offs = (torch.arange(B + 1, dtype=torch.int64) * L).to(torch.int32).to(device)
    seq = torch.full((B,), T, dtype=torch.int32).to(device)
    for _ in range(2):
        ops.ctc_loss(logits, labels, offs, seq, L)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        ops.ctc_loss(logits, labels, offs, seq, L)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    gbs = T * B * (8 * V + 8 * (2 * L + 1)) / (ms * 1e-3) / 1e9
    return {"B": B, "avg_call_ms": round(ms, 4), "achieved": round(gbs, 1), "unit": "GB/s",
            "frac": round(gbs / PEAK_HBM_GBS, 4)}


def cpu_baseline_full(name="c2"):
    """ONE full train step of the CPU oracle on the whole of a small workload (c2: 3 x BiLSTM-320, T = 1000, B = 32), as
    SURVEY.md section 8(d) asks: not a sample."""
    from oracle import oracle as orc
    orc.build()
    w = WORKLOADS[name]
    cfg, B, T, L = dict(w["cfg"]), w["B"], w["T"], w["L"]
    params = orc.init_params(cfg, seed=1)
    rng = np.random.default_rng(777)
    x = rng.normal(size=(B, T, cfg["input_dim"])).astype(np.float32)
    seq = np.full(B, T, np.int32)
    labels = rng.integers(0, cfg["num_targets"] - 1, size=(B, L)).astype(np.int64)
    state = {}
    orc.train_step(params, cfg, x[:, :8], np.full(B, 8, np.int32), labels[:, :2], state, optimizer="adam", lr=4e-4,
                   drop_seed=1)                   # untimed: first touch
    t0 = time.time()
    orc.train_step(params, cfg, x, seq, labels, state, optimizer="adam", lr=4e-4, drop_seed=1)
    dt = time.time() - t0
    return {"value": round(B * T / dt, 2), "unit": "frames/s", "cores": orc.num_threads(), "kind": "port",
            "sample": "1 FULL train step of the CPU oracle on %s (B=%d T=%d L=%d, %d frames), %.1f s" %
                      (name, B, T, L, B * T, dt)}


def cpu_baseline(w, budget_s=25.0, probe_T=8, max_T=256):
    """Times ONE train step of the CPU oracle on a bounded sample of the same workload: the same model and batch
    size, T' frames per utterance.  An untimed step first-touches the parameter-sized buffers, a short probe step
    gives the rate, T' is chosen so that the timed step takes about `budget_s`, and `value` is that step's
    frames / its wall time (the part of a step that does not scale with T - L2 / clip / Adam over all parameters -
    stays inside, so the figure understates a full T = 1000 step somewhat)."""
    from oracle import oracle as orc
    orc.build()
    cfg = dict(w["cfg"])
    B = w["B"]
    params = orc.init_params(cfg, seed=1)
    state = {}                                     # Adam slots live across steps, as in a real run

    def timed_step(Tp):
        rng = np.random.default_rng(777)
        Lp = max(1, Tp // 4)
        x = rng.normal(size=(B, Tp, cfg["input_dim"])).astype(np.float32)
        seq = np.full(B, Tp, np.int32)
        labels = rng.integers(0, cfg["num_targets"] - 1, size=(B, Lp)).astype(np.int64)
        t0 = time.time()
        orc.train_step(params, cfg, x, seq, labels, state, optimizer="adam", lr=4e-4, drop_seed=1)
        return time.time() - t0, Lp

    timed_step(min(4, probe_T))                    # untimed: first touch
    t_probe, _ = timed_step(probe_T)
    Tp = int(min(max_T, max(probe_T, budget_s / t_probe * probe_T)))
    dt, Lp = timed_step(Tp)
    return {"value": round(B * Tp / dt, 2), "unit": "frames/s", "cores": orc.num_threads(), "kind": "port",
            "sample": "1 train step of the CPU oracle (restatement of the TF-1.8 graph; TF not installable), "
                      "same model, B=%d T=%d L=%d (%d frames), %.1f s" % (B, Tp, Lp, B * Tp, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c4", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="skip the per-kernel HIP-event bracketing")
    ap.add_argument("--host-batch", action="store_true",
                    help="hand every step the batch as HOST numpy arrays in the loader's contract (PCIe-inclusive "
                         "rate, for DESIGN.md; never the headline value)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU path)"
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    pg = None
    if world > 1 or "TORCHELASTIC_RUN_ID" in os.environ:     # launched by torch.distributed.run (also at N = 1)
        if os.environ.get("NCCL_DEBUG", "").upper() == "VERSION":      # keep stdout to the one JSON line
            os.environ["NCCL_DEBUG"] = "WARN"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        torch.distributed.init_process_group("nccl", device_id=device)   # nccl == RCCL on ROCm
        pg = torch.distributed.group.WORLD

    from lstm_ctc_amd import ops
    from lstm_ctc_amd.nnet.graph import create_graph_for_training_ctc

    w = WORKLOADS[args.workload]
    bf16 = w["cfg"].get("compute_dtype") == "bf16"
    graph = create_graph_for_training_ctc(None, w["cfg"], learn_rate=4e-4, clip_norm=5.0, optimizer="adam",
                                          device=device, seed=123, process_group=pg)   # same init on every rank
    x, seq, labels, offs = synth_batch(w, rank, device)
    size = int(labels.numel())
    frames_per_step = int(seq.sum().item())

    host_batch = None
    if args.host_batch:            # nnet/pipeline.py:35-61: feats [B,Tmax,D] f32, labels [B,Lmax] int64 (-1 pad), lengths [B]
        host_batch = {"nnet_input": x.permute(1, 0, 2).contiguous().cpu().numpy(),
                      "sequence_length": seq.cpu().numpy(),
                      "nnet_target": labels.view(w["B"], w["L"]).cpu().numpy().astype(np.int64)}

    def one_step():
        if host_batch is not None:
            return graph.step(host_batch, fetch_eval=False)
        return graph.step_device(x, seq, labels, offs, w["L"], size, fetch_eval=False)

    for _ in range(args.warmup):
        out = one_step()

    def barrier():
        if pg is not None:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    ops.PROFILE = None if args.no_profile else []
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = one_step()
    barrier()
    dt = time.perf_counter() - t0
    prof, ops.PROFILE = ops.PROFILE, None
    if pg is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        total_frames = frames_per_step * world * args.steps
        line = {
            "metric": "acoustic frames/sec (whole node), 5xBiLSTM-1024 CTC" if args.workload == "c4"
                      else "acoustic frames/sec (whole node)",
            "value": round(total_frames / dt, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16" if bf16 else "f32",
            "data": "synthetic" + (" (host batch each step: PCIe-inclusive)" if args.host_batch else ""),
            "config": {"workload": w["desc"], "global_batch": w["B"] * world, "seq_len": w["T"],
                       "parallelism": "dp%d" % world, "optimizer": "adam lr 4e-4, clip 5, L2 1e-5",
                       # how many ranks the collective library itself saw (None: launched bare, no process group)
                       "rccl_ranks": torch.distributed.get_world_size(pg) if pg is not None else None,
                       "collective_backend": torch.distributed.get_backend(pg) if pg is not None else None,
                       # per-layer gradient buckets all-reduced during the backward (dp.GradientBuckets); None without a group
                       "dp_buckets": (bool(graph.dp_buckets) and not graph.model.overlap_wgrad) if pg is not None else None,
                       "persist_fallbacks": graph.persist_fallbacks,
                       "lstm_schedule": ops.last_lstm_schedule()["kind"],
                       "last_loss_per_label": round(out["eval_loss"] / max(size, 1), 4)},
        }
        if prof:
            agg = {}
            for kind, work, s, e in prof:
                ms = s.elapsed_time(e)
                a = agg.setdefault(kind, [0.0, 0.0, 0])
                if kind == "ctc":
                    T_, B_, V_ = work
                    work = float(T_ * B_ * (8 * V_ + 8 * (2 * w["L"] + 1)))
                a[0] += work
                a[1] += ms
                a[2] += 1
            g = agg.get("gemm_bf16" if bf16 else "gemm")
            if g:
                tf = g[0] / (g[1] * 1e-3) / 1e12
                peak = PEAK_BF16_MFMA_TFLOPS if bf16 else PEAK_F32_MFMA_TFLOPS
                line["roofline"] = {"kernel": "gemm_bf16g_kernel (v_mfma_f32_32x32x16_bf16, 256 x 256 x 64 tiles, bf16 shadow "
                                              "operands DMA'd into LDS; gemm_bf16s_kernel / gemm_bf16_kernel on ragged "
                                              "shapes and K % 8 != 0)" if bf16
                                    else "gemm_f32g_kernel (v_mfma_f32_32x32x2_f32, 256 x 256 x 32 tiles, LDS-DMA operands) + "
                                         "gemm_f32_kernel (128 x 128 tiles) on shapes that do not fill whole rounds",
                                    "bound": "mfma",
                                    "achieved": round(tf, 2), "peak": peak, "unit": "TFLOP/s",
                                    "frac": round(tf / peak, 4), "traffic": measured_traffic(args.workload, "gemm"),
                                    "launches": g[2], "avg_launch_ms": round(g[1] / g[2], 4),
                                    "share_of_step": round(g[1] / (dt * 1e3), 3)}
            c = agg.get("ctc")
            if c:
                gbs = c[0] / (c[1] * 1e-3) / 1e9
                line["roofline_ctc"] = {"kernel": "ctc_mm_kernel phase 1 + phase 2 (alpha / beta meet in the middle, "
                                                  "gradient inside the scan)", "bound": "hbm",
                                        "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                        "frac": round(gbs / PEAK_HBM_GBS, 4),
                                        "traffic": measured_traffic(args.workload, "ctc"),
                                        "avg_call_ms": round(c[1] / c[2], 4)}
                try:        # the same op with all local micro-batches in one launch (B = 512): issue-bound regime
                    line["roofline_ctc"]["large_batch"] = ctc_large_batch(w, device)
                except Exception as exc:
                    line["roofline_ctc"]["large_batch"] = {"error": repr(exc)}
            for kind in ("lstm_fwd", "lstm_bwd"):
                r = agg.get(kind)
                if r:
                    line.setdefault("breakdown_ms_per_step", {})[kind] = round(r[1] / args.steps, 3)
                    line.setdefault("recurrence_tflops", {})[kind] = round(r[0] / (r[1] * 1e-3) / 1e12, 2)
            if g:
                line.setdefault("breakdown_ms_per_step", {})["gemm"] = round(g[1] / args.steps, 3)
            if bf16 and agg.get("gemm"):         # the weight-only folds (R = proj.Kh and its gradient) stay fp32
                line["breakdown_ms_per_step"]["gemm_f32_weight_folds"] = round(agg["gemm"][1] / args.steps, 3)
            if agg.get("cast_bf16"):             # fp32 -> bf16 shadow copies (natural / transposed) of the operands
                cb = agg["cast_bf16"]
                line["breakdown_ms_per_step"]["cast_bf16"] = round(cb[1] / args.steps, 3)
                line["cast_bf16_gbs"] = round(cb[0] / (cb[1] * 1e-3) / 1e9, 1)
            if c:
                line.setdefault("breakdown_ms_per_step", {})["ctc"] = round(c[1] / args.steps, 3)
        if world == 1 and not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline(w)
                if args.workload == "c4":                 # plus one small config in full (no sampling)
                    line["cpu_baseline_c2_full"] = cpu_baseline_full("c2")
            except Exception as exc:                      # the oracle is a reported baseline, never the product
                line["cpu_baseline"] = {"error": repr(exc)}
        print(json.dumps(line), flush=True)
    if pg is not None:
        torch.distributed.barrier()            # rank 0 is still measuring / printing: tear the group down together
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
