#!/usr/bin/env python3
"""bench.py — the headline benchmark: acoustic frames/sec of full BiLSTM-CTC train steps on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload (BASELINE.json metric, config c4): 5 x BiLSTM-1024 (P = N, peepholes) + affine head, V = 44,
synthetic 40-dim fbank, T = 1000, B = 64 utterances per GPU, L = 100 labels, fp32, adam lr 4e-4,
clip 5, L2 1e-5, output dropout keep 0.9.  One "step" = forward + CTC loss/gradient + BPTT backward +
(N > 1) RCCL all-reduce of the flat fp32 gradient + L2/clip/Adam update, inputs resident in HBM.
frames = sum_b sequence_length_b; value = frames of all ranks / max-over-ranks time (weak scaling).

Launching: `python bench.py --gpus N` with N > 1 and no torchrun environment starts the N ranks itself (a child
`python -m torch.distributed.run`; the parent never touches the GPU) and relays rank 0's line; a rank count that differs
from --gpus, or fewer visible GPUs than ranks, is an error (non-zero exit, no JSON line) - never a smaller job under an
N-GPU label.

Extra objects on the JSON line:
  roofline      dominant kernel (the f32 MFMA GEMM): algorithmic FLOPs / HIP-event time, measured live on the
                launch stream over the timed steps, against the 157.3 TFLOP/s fp32 matrix peak.
  roofline_ctc  the CTC op (row stats + alpha/beta scan + gradient): algorithmic bytes T*B*(8V+8S) / time,
                against the 8 TB/s HBM peak (north-star target: >= 40 %).
  secondary     (c4, N = 1) the other BASELINE configs - c5, c2, c3 - and c4x3 / c2x3 / c3x3 (the fp32 configurations with
                their products AND recurrences as fp32-on-bf16x3 split operands, DESIGN.md sections 3f, 3g) timed in the same
                process after the headline region (5 warm-up + 10 timed steps each): ms_per_step, frames/s, their GEMM / CTC
                rooflines (c1 - c3: GEMM rates from three extra steps without the weight-gradient overlap).  Printed
                compacted against the headline (`compact_secondary`; constants in `secondary_protocol`) so that the whole
                line stays inside the driver's 8 KB window.
  inference     (c4, N = 1) the forward pass alone on the workload's batch (is_training false): frames/s for c4, c4x3, c3
                (the high-rank head), c3x3 and c5.
  cli_corpus    (c4, N = 1) bin/nnet-train.py as a child process on a synthetic TFRecord corpus (c4 and c2) next to the
                resident-input rate of the same model: what the loader + upload + run loop cost end to end.
  allreduce     (N > 1) time the compute stream waited for gradient collectives per step, and the whole 480 MB
                gradient all-reduce timed alone (algorithmic and bus bandwidth).
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

# torch is imported inside the functions that need it: the parent of a self-launched N-rank run (--gpus N without
# torchrun) must not initialise the GPU before it has started its ranks as child processes.

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
    "c4x3": dict(desc="c4x3: c4 with the activation products AND the recurrent step products as fp32-on-bf16x3 (each fp32 "
                      "operand split exactly into 3 bf16 terms, 6 bf16 MFMA term products per fp32 product, fp32 accumulate: "
                      "fp32-grade results; gate math, state, CTC, optimizer as in c4): 5xBiLSTM-1024, V=44, T=1000 B=64/GPU L=100",
                 cfg=dict(nnet_type="blstm", input_dim=40, left_context=0, right_context=0, num_layers=5,
                          num_neurons=1024, num_projects=1024, num_targets=44, use_peepholes=True, dropout_rate=0.9,
                          compute_dtype="bf16x3"),
                 B=64, T=1000, L=100),
}
# tests only (tests/test_gpu_round6.py): c4's layer structure - 5 bidirectional layers + affine head, so the same gradient
# bucket ranges, finish() leftovers and per-rank dropout streams as the headline - at toy width, so that EIGHT ranks can share the
# one GPU of a test box over gloo and rehearse the exact `--gpus 8` launch path.  Never a performance figure.
WORKLOADS["rehearsal"] = dict(desc="rehearsal: 5xBiLSTM-64 (P=N, peepholes) CTC, V=12, T=24 B=2/rank L=4, fp32 (tests only)",
                              cfg=dict(nnet_type="blstm", input_dim=40, left_context=0, right_context=0, num_layers=5,
                                       num_neurons=64, num_projects=64, num_targets=12, use_peepholes=True, dropout_rate=0.9),
                              B=2, T=24, L=4)
WORKLOADS["rehearsal_keep1"] = dict(WORKLOADS["rehearsal"], desc=WORKLOADS["rehearsal"]["desc"].replace("fp32", "no dropout, fp32"),
                                    cfg=dict(WORKLOADS["rehearsal"]["cfg"], dropout_rate=1.0))
for _n in ("c2", "c3"):          # the same split-operand mode on the smaller configurations (not in the default secondary set)
    WORKLOADS[_n + "x3"] = dict(WORKLOADS[_n], desc=WORKLOADS[_n]["desc"].replace(", fp32", "") + ", fp32 products as bf16x3",
                                cfg=dict(WORKLOADS[_n]["cfg"], compute_dtype="bf16x3"))
PEAK_F32_MFMA_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md chip table
PEAK_BF16_MFMA_TFLOPS = 2500.0   # dense bf16 (same table)
PEAK_HBM_GBS = 8000.0


def measured_traffic(workload, kernel):
    """HBM-side bytes per launch of `kernel` ("gemm" / "ctc") from the rocprofv3 PMC passes of THIS round
    (tools/pmc_traffic.py writes profiles/r3_pmc_traffic.json: FETCH_SIZE corrected x2 for wide reads + WRITE_SIZE, as
    /opt/skills/guides/MI355X_MICROARCH.md prescribes).  PMC counters cannot be collected inside a timed run, so the
    figure belongs to the profiled run of the same command; None when no such file covers this workload."""
    return _traffic(workload, kernel)[0]


def _traffic(workload, kernel):
    """(bytes per launch or None, the committed file the figure was read from or None)."""
    try:
        for name in ("r6_pmc_traffic.json", "r5_pmc_traffic.json", "r4_pmc_traffic.json", "r3_pmc_traffic.json", "r2_pmc_traffic.json"):   # this round's passes, else older
            path = os.path.join(ROOT, "profiles", name)
            if os.path.exists(path):
                with open(path) as f:
                    v = json.load(f).get(workload, {}).get(kernel)
                if v is not None:
                    return v, "profiles/" + name
    except (OSError, ValueError):
        pass
    return None, None


def traffic_fields(workload, kernel):
    """`traffic` + `traffic_source` of a roofline object: the figure is READ from the committed PMC summary of a profiled
    run of the same command (counters cannot be collected inside a timed run), never measured in this run."""
    v, src = _traffic(workload, kernel)
    return {"traffic": v, "traffic_source": (src + " (PMC passes of a profiled run; not measured in this run)") if src else None}


def synth_batch(w, rank, device):
    """SURVEY.md §8d throughput set: x ~ N(0,1), all T_b = T, L_b = L, labels uniform on {0..V-2}."""
    import torch
    g = torch.Generator().manual_seed(777 + rank)
    c = w["cfg"]
    B, T, L, V = w["B"], w["T"], w["L"], c["num_targets"]
    D = c["input_dim"] * (1 + (c.get("left_context") or 0) + (c.get("right_context") or 0))   # spliced width
    x = torch.randn((T, B, D), generator=g, dtype=torch.float32)
    labels = torch.randint(0, V - 1, (B * L,), generator=g, dtype=torch.int32)
    offs = (torch.arange(B + 1, dtype=torch.int64) * L).to(torch.int32)
    seq = torch.full((B,), T, dtype=torch.int32)
    return x.to(device), seq.to(device), labels.to(device), offs.to(device)


def synth_batch_ragged(w, rank, device):
    """SURVEY.md section 8d PARITY set as a throughput workload (real recipe batches are ragged and length-sorted,
    egs/wsj/run_wsj_phn.sh:143-147): T_b ~ U{0.6 T .. T} sorted ascending, L_b ~ U{0.2 L .. 1.2 L}, frames past T_b zero,
    utterance 0 holds an adjacent repeated label.  Returns (x [T,B,D], seq, labels flat, offsets, max label length)."""
    import torch
    g = torch.Generator().manual_seed(777 + rank)
    c = w["cfg"]
    B, T, L, V = w["B"], w["T"], w["L"], c["num_targets"]
    D = c["input_dim"] * (1 + (c.get("left_context") or 0) + (c.get("right_context") or 0))
    seq = torch.sort(torch.randint(int(0.6 * T), T + 1, (B,), generator=g, dtype=torch.int32)).values
    lens = torch.randint(max(1, int(0.2 * L)), int(1.2 * L) + 1, (B,), generator=g, dtype=torch.int32)
    x = torch.randn((T, B, D), generator=g, dtype=torch.float32)
    x *= (torch.arange(T)[:, None] < seq[None, :]).to(torch.float32)[:, :, None]
    offs = torch.cat([torch.zeros(1, dtype=torch.int64), torch.cumsum(lens.to(torch.int64), 0)]).to(torch.int32)
    labels = torch.randint(0, V - 1, (int(offs[-1]),), generator=g, dtype=torch.int32)
    if int(lens[0]) >= 2:
        labels[1] = labels[0]
    return x.to(device), seq.to(device), labels.to(device), offs.to(device), int(lens.max())


def ctc_large_batch(w, device, B=512):
    """The CTC op alone on B utterances of the workload's shape (all local micro-batches scanned in one launch): where
    the scan is no longer bound by the T-long chain of one utterance but by issue / HBM.  Outside the timed region."""
    import torch
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


def host_memory_available_gb():
    """What this process may still allocate: the smaller of the machine's MemAvailable and the cgroup's limit - usage."""
    avail = None
    try:
        with open("/proc/meminfo") as f:
            for l in f:
                if l.startswith("MemAvailable:"):
                    avail = int(l.split()[1]) * 1024
    except OSError:
        pass
    for lim, cur in (("/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory.current"),
                     ("/sys/fs/cgroup/memory/memory.limit_in_bytes", "/sys/fs/cgroup/memory/memory.usage_in_bytes")):
        try:
            a, b = open(lim).read().strip(), open(cur).read().strip()
            if a != "max":
                room = int(a) - int(b)
                avail = room if avail is None else min(avail, room)
        except (OSError, ValueError):
            pass
    return None if avail is None else avail / 1e9


# peak resident memory of one oracle train step, measured in the build container (c4, f32): 4.2 GB + 25 MB per frame index
CPU_FULL_NEED_GB = {"c4": 30.0, "c2": 2.0}


def cpu_baseline_full(name="c2"):
    """ONE full train step of the CPU oracle on the WHOLE workload (c4: 5 x BiLSTM-1024, B = 64, T = 1000, L = 100 - the
    benched batch itself; c2: 3 x BiLSTM-320, B = 32), as SURVEY.md section 8(d) asks: not a sample.  Skipped - with the
    reason in place of a value - when the host cannot hold the step's saved activations (c4: ~30 GB)."""
    from oracle import oracle as orc
    orc.build()
    w = WORKLOADS[name]
    need, have = CPU_FULL_NEED_GB.get(name, 0.0), host_memory_available_gb()
    if have is not None and have < 1.5 * need:
        return {"skipped": "the full %s step needs ~%.0f GB of host memory for the oracle's saved activations; %.1f GB "
                           "available to this process" % (name, need, have)}
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
            "sample": "1 train step of the CPU oracle (TF-1.8 restated; TF not installable), same model, "
                      "B=%d T=%d L=%d (%d frames), %.1f s" % (B, Tp, Lp, B * Tp, dt)}


# the other BASELINE configs timed after the headline (c2x3 / c3x3 - the split-operand mode on the small configurations - left
# the default set in round 6: `--workload c2x3` still runs them)
SECONDARY_DEFAULT = ("c5", "c4x3", "c2", "c3")
SUMMARY_KEYS = ("c4_ms", "c4x3_ms", "c5_ms", "c2_ms", "c3_ms", "gemm_frac", "ctc_frac_b64", "ctc_frac_b512",
                "ctc_traffic_ratio_b512_profiled", "c5_gemm_frac", "c5_rec_ms", "c4x3_rec_ms", "c4_rec_ms", "c4x3_gemm_frac",
                "c4_frames_s", "c4x3_frames_s", "c5_frames_s", "c4_ragged_frames_s", "c4_ragged_padded_share",
                "c2_ragged_frames_s", "cpu_frames_s", "cpu_cores", "cpu_sample")


def build_summary(line, workload="c4"):
    """The figures a reader needs, as ONE small object (< 1 KB) that main() appends as the LAST key of the line: the driver keeps
    only the last ~8 KB of stdout, and round 4's line had outgrown that (c5 / c4x3 / the B = 512 CTC figure fell off the front).
    Every value is copied from the line itself (nothing is measured here); a leg that did not run leaves None."""
    sec = dict(line.get("secondary") or {})
    sec[workload] = line

    def get(d, *path):
        for k in path:
            d = d.get(k) if isinstance(d, dict) else None
        return d

    def rec_ms(name):
        bd = get(sec.get(name), "breakdown_ms_per_step")
        if not bd or bd.get("lstm_fwd") is None or bd.get("lstm_bwd") is None:
            return None
        return round(bd["lstm_fwd"] + bd["lstm_bwd"], 3)

    out = {}
    for name in ("c4", "c4x3", "c5", "c2", "c3"):
        out[name + "_ms"] = get(sec.get(name), "ms_per_step")
    out["gemm_frac"] = get(sec.get("c4"), "roofline", "frac")
    out["ctc_frac_b64"] = get(sec.get("c4"), "roofline_ctc", "frac")
    out["ctc_frac_b512"] = get(sec.get("c4"), "roofline_ctc", "large_batch", "frac")
    w = WORKLOADS["c4"]
    alg = w["T"] * 512 * (8 * w["cfg"]["num_targets"] + 8 * (2 * w["L"] + 1))
    tr = _traffic("ctc_b512", "ctc")[0]
    # (PMC passes of a PROFILED run of the same command, read from the committed profiles/ file: not measured in this run)
    out["ctc_traffic_ratio_b512_profiled"] = round(tr / alg, 3) if tr else None
    out["c5_gemm_frac"] = get(sec.get("c5"), "roofline", "frac")
    out["c5_rec_ms"], out["c4x3_rec_ms"], out["c4_rec_ms"] = rec_ms("c5"), rec_ms("c4x3"), rec_ms("c4")
    out["c4x3_gemm_frac"] = get(sec.get("c4x3"), "roofline", "frac")
    for name in ("c4", "c4x3", "c5"):
        out[name + "_frames_s"] = get(sec.get(name), "value")
    out["c4_ragged_frames_s"] = get(sec.get("c4_ragged"), "value")
    out["c4_ragged_padded_share"] = get(sec.get("c4_ragged"), "ragged", "padded_frame_share")
    out["c2_ragged_frames_s"] = get(sec.get("c2_ragged"), "value")
    out["cpu_frames_s"], out["cpu_cores"] = get(line, "cpu_baseline", "value"), get(line, "cpu_baseline", "cores")
    smp = get(line, "cpu_baseline", "sample")
    out["cpu_sample"] = ("full step" if "FULL" in smp else "bounded sample") if isinstance(smp, str) else None
    assert tuple(out) == SUMMARY_KEYS
    return out


def strip_notes(obj):
    """Drops the explanatory `note` strings from every object of the line (they are DESIGN.md section 4's text, repeated
    per roofline they cost ~2 KB of the driver's 8 KB window)."""
    if isinstance(obj, dict):
        return {k: strip_notes(v) for k, v in obj.items() if k != "note"}
    if isinstance(obj, list):
        return [strip_notes(v) for v in obj]
    return obj


SECONDARY_CONFIG_KEEP = ("workload", "persist_fallbacks", "lstm_schedule", "last_loss_per_label", "product_kernels")


SECONDARY_PROTOCOL = ("10 + 5 warm-up steps each, after the headline, same process; frames/s; rooflines: TFLOP/s against the "
                      "f32 (157.3) / bf16 (2500; x3: term products) MFMA peak; product_kernels: FLOP shares")


def compact_secondary(name, entry, head, head_name="c4", ctc_seen=None):
    """One `secondary` entry without what the headline (or `secondary_protocol`) already says: config keys equal to the
    headline's, the workload text after its first clause, the rooflines' constant fields (bound / peak / unit /
    traffic_source / launch counts), the CTC roofline of a workload whose CTC launches are another entry's (c5 / c4x3: the
    headline's; c2x3: c2's), float noise in `traffic`.  Nothing is re-measured or renamed; the whole default line fits the
    driver's 8 KB window again (it was 12.8 KB in round 5's first runs)."""
    if not isinstance(entry, dict) or "config" not in entry:
        return entry
    e = {k: v for k, v in entry.items() if k not in ("unit", "steps", "warmup")}
    hc, cfg = head.get("config") or {}, dict(e["config"])
    for k in list(cfg):
        if k not in SECONDARY_CONFIG_KEEP and cfg[k] == hc.get(k):
            del cfg[k]
    wl = str(cfg.get("workload", "")).split(" (")[0].split(", ")[0][:64]
    cfg["workload"] = wl if wl.startswith(name + ":") else "%s: %s" % (name, wl.split(": ", 1)[-1])
    if isinstance(cfg.get("product_kernels"), dict):
        cfg["product_kernels"] = {k: v.get("flop_share") if isinstance(v, dict) else v for k, v in cfg["product_kernels"].items()}
    e["config"] = cfg
    for key in ("roofline", "roofline_ctc", "roofline_f32_leftovers", "roofline_x3"):
        r = e.get(key)
        if isinstance(r, dict):
            r = {k: v for k, v in r.items() if v is not None and k not in ("bound", "peak", "unit", "traffic_source", "launches")}
            if isinstance(r.get("traffic"), float):
                r["traffic"] = int(round(r["traffic"]))
            e[key] = r
    if isinstance(e.get("roofline_f32_leftovers"), dict):
        e["roofline_f32_leftovers"] = {k: v for k, v in e["roofline_f32_leftovers"].items() if k in ("frac", "share_of_step")}
    e.pop("cast_bf16_gbs", None)
    if name.endswith("_ragged"):                  # same kernels as the all-T entry of that name: its step time and frame counts matter
        e.pop("recurrence_tflops", None)
        e.pop("dtype", None)
        e["config"].pop("product_kernels", None)
        e["config"].pop("last_loss_per_label", None)
        for key in ("roofline", "roofline_ctc"):
            if isinstance(e.get(key), dict):
                e[key] = {"frac": e[key].get("frac")}
    a = WORKLOADS.get(name)
    if a and isinstance(e.get("roofline_ctc"), dict) and ctc_seen is not None:
        shape = (a["B"], a["T"], a["L"], a["cfg"]["num_targets"])
        if shape in ctc_seen:
            e["roofline_ctc"] = {"frac": e["roofline_ctc"].get("frac"), "same_launches_as": ctc_seen[shape]}
        else:
            ctc_seen[shape] = name
    return e


def finalize_line(line, workload="c4"):
    """What main() prints: the line without `note` strings, `secondary` entries compacted against the headline, and with
    `summary` as its LAST key."""
    line = strip_notes({k: v for k, v in line.items() if k != "summary"})
    if isinstance(line.get("secondary"), dict):
        h = WORKLOADS.get(workload)
        seen = {(h["B"], h["T"], h["L"], h["cfg"]["num_targets"]): workload} if h and isinstance(line.get("roofline_ctc"), dict) else {}
        sec = {k: compact_secondary(k, v, line, workload, seen) for k, v in line["secondary"].items()}
        line = {k: v for k, v in line.items() if k != "secondary_protocol"}
        out = {}
        for k, v in line.items():                 # `secondary_protocol` right in front of `secondary`
            if k == "secondary":
                out["secondary_protocol"] = SECONDARY_PROTOCOL
                v = sec
            out[k] = v
        line = out
    if isinstance(line.get("inference"), dict):
        line["inference"] = {k: ({kk: vv for kk, vv in v.items() if kk not in ("batch", "seq_len", "lstm_schedule")} if isinstance(v, dict) else v)
                             for k, v in line["inference"].items()}
    cc = line.get("cli_corpus")
    if isinstance(cc, dict):                      # the corpus description once (entries differ by --batch-size only)
        first = True
        for k, v in cc.items():
            if isinstance(v, dict) and "ratio" in v:
                cc[k] = {kk: vv for kk, vv in v.items() if kk not in ("steps_counted", "process_wall_s") and (first or kk != "corpus")}
                first = False
    for r in ("roofline", "roofline_ctc"):
        if isinstance(line.get(r), dict) and isinstance(line[r].get("traffic"), float):
            line[r]["traffic"] = int(round(line[r]["traffic"]))
    line["summary"] = build_summary(line, workload)
    return line


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="c4", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="skip the per-kernel HIP-event bracketing")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the c5 / c2 / c3 lines that a default c4 run at N = 1 appends as `secondary`")
    ap.add_argument("--no-cli-corpus", action="store_true",
                    help="skip the end-to-end bin/nnet-train.py leg (c4 and c2 on a synthetic TFRecord corpus) that a "
                         "default c4 run at N = 1 appends as `cli_corpus`")
    ap.add_argument("--launch", choices=("auto", "torchrun", "none"), default="auto",
                    help="auto: --gpus N > 1 without torchrun's environment starts N ranks as a child "
                         "`python -m torch.distributed.run`; torchrun: do that at N = 1 too; none: never")
    ap.add_argument("--backend", choices=("nccl", "gloo"), default="nccl",
                    help="collective backend under torchrun: nccl (= RCCL, the product's) or gloo (tests only: with "
                         "LC_BENCH_SHARED_GPU=1 all ranks share GPU 0, which RCCL refuses, so that the N > 1 code path "
                         "can be executed on a one-GPU box; never a performance figure)")
    ap.add_argument("--host-batch", action="store_true",
                    help="hand every step the batch as HOST numpy arrays in the loader's contract (PCIe-inclusive "
                         "rate, for DESIGN.md; never the headline value)")
    return ap.parse_args(argv)


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(args, argv):
    """`python bench.py --gpus N` without torchrun's environment: start the N ranks as a CHILD
    `python -m torch.distributed.run` (never exec: a process that has touched the GPU must not be replaced, and this
    parent does not touch it - torch.cuda.device_count() does not initialise HIP), relay the ranks' output, exit with
    the child's status.  Refuses - non-zero, no JSON line - when the box shows fewer than N GPUs: an N-GPU number must
    never come from fewer devices."""
    import subprocess
    import torch
    have = torch.cuda.device_count()
    shared = args.backend == "gloo" and os.environ.get("LC_BENCH_SHARED_GPU") == "1"      # tests only: all ranks on GPU 0
    if have < args.gpus and not (shared and have >= 1):
        sys.stderr.write("bench.py: --gpus %d asked, %d GPU(s) visible: refusing to run (an N-GPU figure measured on "
                         "fewer devices would be wrong)\n" % (args.gpus, have))
        return 2
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % args.gpus,
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ, LC_BENCH_SELF_LAUNCHED="1")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    for line in proc.stdout:                       # rank 0 prints the one JSON line; pass everything through
        sys.stdout.write(line)
        sys.stdout.flush()
    return proc.wait()


def lc_overrides():
    """Every LC_* development switch present in the environment: a stray one changes which kernel runs, so the bench
    line shows them (empty dict = library defaults)."""
    ours = ("LC_LSTM_", "LC_GEMM_", "LC_CTC_", "LC_DP_", "LC_OVERLAP_", "LC_X3_", "LC_FUSE_")          # (LC_CTYPE, LC_ALL, ... are the locale's)
    return {k: v for k, v in sorted(os.environ.items()) if k.startswith(ours)}


def allreduce_alone(graph, pg, device, iters=5):
    """The gradient exchange by itself, outside the timed region: one all-reduce(sum) of the whole flat fp32 gradient
    buffer on RCCL, HIP-event timed.  busbw = 2 (N-1)/N x bytes / time, the per-link figure ring collectives are
    priced by."""
    import torch
    flat = graph.model.ps.grad
    world = torch.distributed.get_world_size(pg)
    if torch.distributed.get_backend(pg) == "gloo":          # tests: staged through the host, not a bandwidth figure
        from lstm_ctc_amd.nnet import dp
        t0 = time.perf_counter()
        dp.allreduce_sum_(flat, pg)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3
        return {"bytes": flat.numel() * 4, "ms": round(ms, 3), "algbw_GBs": None, "busbw_GBs": None, "backend": "gloo"}
    for _ in range(2):
        torch.distributed.all_reduce(flat, group=pg)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    torch.distributed.barrier()
    e0.record()
    for _ in range(iters):
        torch.distributed.all_reduce(flat, group=pg)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    nbytes = flat.numel() * 4
    return {"bytes": nbytes, "ms": round(ms, 3), "algbw_GBs": round(nbytes / (ms * 1e-3) / 1e9, 1),
            "busbw_GBs": round(2.0 * (world - 1) / world * nbytes / (ms * 1e-3) / 1e9, 1)}


class UnhealthyRun(RuntimeError):
    """A rank spent timed steps off the schedule the figure is quoted for (see run_workload): no value is printed."""


def run_workload(name, steps, warmup, device, pg, rank, world, profile=True, host_batch=False, full=True,
                 workload=None, ragged=False):
    """Warm-up + EXACTLY `steps` timed train steps of one workload between barrier + synchronize; returns the fields of
    its JSON object (rank 0; other ranks get None).  full=False: the compact form used for `secondary` entries."""
    import torch
    from lstm_ctc_amd import ops
    from lstm_ctc_amd.nnet.graph import create_graph_for_training_ctc

    w = workload or WORKLOADS[name]
    bf16 = w["cfg"].get("compute_dtype") == "bf16"
    x3 = w["cfg"].get("compute_dtype") == "bf16x3"
    graph = create_graph_for_training_ctc(None, w["cfg"], learn_rate=4e-4, clip_norm=5.0, optimizer="adam",
                                          device=device, seed=123, process_group=pg)   # same init on every rank
    if ragged:
        x, seq, labels, offs, maxlen = synth_batch_ragged(w, rank, device)
    else:
        x, seq, labels, offs = synth_batch(w, rank, device)
        maxlen = w["L"]
    size = int(labels.numel())
    frames_per_step = int(seq.sum().item())
    # algorithmic bytes of one CTC call: sum_b T_b (8 V + 8 S_b), S_b = 2 L_b + 1
    _lens = (offs[1:] - offs[:-1]).to(torch.int64)
    ctc_bytes_per_call = float((seq.to(torch.int64) * (8 * w["cfg"]["num_targets"] + 8 * (2 * _lens + 1))).sum().item())

    hb = None
    if host_batch:            # nnet/pipeline.py:35-61: feats [B,Tmax,D] f32, labels [B,Lmax] int64 (-1 pad), lengths [B]
        hb = {"nnet_input": x.permute(1, 0, 2).contiguous().cpu().numpy(),
              "sequence_length": seq.cpu().numpy(),
              "nnet_target": labels.view(w["B"], w["L"]).cpu().numpy().astype(np.int64)}

    def one_step():
        if hb is not None:
            return graph.step(hb, fetch_eval=False)
        return graph.step_device(x, seq, labels, offs, maxlen, size, fetch_eval=False)

    for _ in range(warmup):
        out = one_step()

    def barrier():
        if pg is not None:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    ops.PROFILE = [] if profile else None
    fallbacks_before = graph.persist_fallbacks
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = one_step()
    barrier()
    dt_local = time.perf_counter() - t0
    prof, ops.PROFILE = ops.PROFILE, None
    dump = os.environ.get("LC_BENCH_DUMP_DIR")          # tests only: every rank's replica after the timed steps
    if dump:
        torch.save({"flat": graph.model.ps.flat.detach().cpu(), "global_step": graph.global_step,
                    "drop_seed": graph.drop_seed, "rank": rank, "world": world},
                   os.path.join(dump, "%s_rank%d.pt" % (name, rank)))
    # Health of EVERY rank, not rank 0's alone: `value` is all ranks' frames over the MAX of their times, so one rank that
    # re-ran steps on the launch train (a persistent launch that could not complete) - or latched onto it in
    # the warm-up - would set the whole node's number and the line would not say why.  Such a run prints no value.
    health = {"rank": rank, "persist_fallbacks": int(graph.persist_fallbacks),
              "fallbacks_in_timed_steps": int(graph.persist_fallbacks - fallbacks_before),
              "latched": bool(graph._fallback.latched), "lstm_schedule": ops.last_lstm_schedule()["kind"],
              "last_loss_per_label": float(out["eval_loss"]) / max(size, 1)}
    ranks = [health]
    if pg is not None:
        ranks = [None] * world
        torch.distributed.all_gather_object(ranks, health, group=pg)
    sick = [h for h in ranks if h["latched"] or h["fallbacks_in_timed_steps"] > 0]
    if sick:
        raise UnhealthyRun("workload %s: rank(s) %s left the persistent LSTM schedule inside the measurement (%s): a figure "
                           "with a rank on the launch train is not this configuration's rate" % (
                               name, [h["rank"] for h in sick],
                               "; ".join("rank %d: %d fall-back(s) in the timed steps, %d in all%s" % (
                                   h["rank"], h["fallbacks_in_timed_steps"], h["persist_fallbacks"],
                                   ", latched" if h["latched"] else "") for h in sick)))
    dt, rank_ms = dt_local, None
    if pg is not None:
        gloo = torch.distributed.get_backend(pg) == "gloo"
        t = torch.tensor([dt_local], dtype=torch.float64, device="cpu" if gloo else device)
        every = [torch.zeros_like(t) for _ in range(world)]
        torch.distributed.all_gather(every, t)
        per_rank = [float(e.item()) for e in every]
        dt = max(per_rank)                               # the contract: MAX over ranks
        rank_ms = {"min": round(min(per_rank) / steps * 1e3, 3), "max": round(dt / steps * 1e3, 3)}
    alone = None
    if pg is not None and world > 1 and full:
        alone = allreduce_alone(graph, pg, device)
    # c1 - c3 run their weight-gradient GEMMs on a side stream UNDER the next layer's BPTT (Model.overlap_wgrad), so a
    # GEMM's event bracket in the timed steps also holds its wait for CUs: not the kernel's rate.  Three more steps with
    # the overlap off, OUTSIDE the timed region, bracket the same launches alone: the rate the roofline object quotes.
    prof_alone = None
    if profile and graph.model.overlap_wgrad and world == 1:
        graph.model.overlap_wgrad = False
        one_step()
        ops.PROFILE = []
        for _ in range(3):
            one_step()
        torch.cuda.synchronize()
        prof_alone, ops.PROFILE = ops.PROFILE, None
        graph.model.overlap_wgrad = True
    if rank != 0:
        return None

    total_frames = frames_per_step * world * steps
    line = {"value": round(total_frames / dt, 1), "unit": "frames/s", "ms_per_step": round(dt / steps * 1e3, 3),
            "dtype": "bf16" if bf16 else "f32 (bf16x3 split operands)" if x3 else "f32", "steps": steps, "warmup": warmup}
    cfg = {"workload": w["desc"], "global_batch": w["B"] * world, "seq_len": w["T"],
           "parallelism": "dp%d" % world, "optimizer": "adam lr 4e-4, clip 5, L2 1e-5",
           # how many ranks the collective library itself saw (None: launched bare, no process group)
           "rccl_ranks": torch.distributed.get_world_size(pg) if pg is not None else None,
           "collective_backend": torch.distributed.get_backend(pg) if pg is not None else None,
           # per-layer gradient buckets all-reduced during the backward (dp.GradientBuckets); None without a group
           "dp_buckets": (bool(graph.dp_buckets) and not graph.model.overlap_wgrad) if pg is not None else None,
           "dp_bucket_ranges_per_step": getattr(graph, "last_bucket_ranges", 0) if pg is not None else None,
           # over ALL ranks (all-gathered): fall-backs to the launch train in warm-up + timed steps (any inside the timed
           # steps, or a latched rank, refuses the run: see above), the recurrence schedule(s) taken, the last loss
           "persist_fallbacks": max(h["persist_fallbacks"] for h in ranks),
           "lstm_schedule": ops.last_lstm_schedule()["kind"],
           "lc_overrides": lc_overrides(),
           "last_loss_per_label": round(out["eval_loss"] / max(size, 1), 4)}
    if rank_ms is not None:
        cfg["per_rank_ms_per_step"] = rank_ms
    if pg is not None:
        losses = [h["last_loss_per_label"] for h in ranks]
        cfg["ranks"] = {"n": len(ranks),
                        "persist_fallbacks": {"min": min(h["persist_fallbacks"] for h in ranks),
                                              "max": max(h["persist_fallbacks"] for h in ranks)},
                        "fallbacks_in_timed_steps": 0, "latched": 0,
                        "lstm_schedule": sorted({h["lstm_schedule"] for h in ranks}),
                        "last_loss_per_label": {"min": round(min(losses), 4), "max": round(max(losses), 4)}}
    if ragged:
        padded = w["B"] * w["T"]
        cfg["workload"] = w["desc"] + " [RAGGED: T_b ~ U{%d..%d} sorted, L_b ~ U{%d..%d}; frames = sum of T_b]" % (
            int(0.6 * w["T"]), w["T"], max(1, int(0.2 * w["L"])), int(1.2 * w["L"]))
        line["ragged"] = {"frames_per_step": frames_per_step, "padded_frames_per_step": padded,
                          "padded_frame_share": round(1.0 - frames_per_step / padded, 4),
                          "padded_frames_s": round(padded * world * steps / dt, 1)}
    line["config"] = cfg
    if prof:
        agg = {}
        for kind, work, s, e in prof:
            ms = s.elapsed_time(e)
            a = agg.setdefault(kind, [0.0, 0.0, 0])
            if kind == "ctc":
                work = ctc_bytes_per_call
            a[0] += work
            a[1] += ms
            a[2] += 1
        # which product kernels the step's GEMM work went to (host-level kinds: "gemm" = the fp32 MFMA kernels, "gemm_bf16" =
        # bf16 operands, "gemm_x3" = fp32 as three-term bf16 split operands): launches per step and share of the product flops
        gk = {k: v for k, v in agg.items() if k.startswith("gemm")}
        gtot = sum(v[0] for v in gk.values()) or 1.0
        cfg["product_kernels"] = {k: {"launches_per_step": round(v[2] / steps, 1), "flop_share": round(v[0] / gtot, 4)}
                                  for k, v in sorted(gk.items())}
        g = agg.get("gemm_bf16" if bf16 else "gemm")
        gx = agg.get("gemm_x3")
        if gx:      # fp32 products issued as 6 bf16 MFMA term products each: priced in bf16 MFMA work against the bf16 peak
            tf = gx[0] / (gx[1] * 1e-3) / 1e12
            line["roofline_x3"] = {"bound": "mfma", "achieved": round(6 * tf, 1), "peak": PEAK_BF16_MFMA_TFLOPS,
                                   "unit": "TFLOP/s", "frac": round(6 * tf / PEAK_BF16_MFMA_TFLOPS, 4),
                                   "fp32_equivalent_tflops": round(tf, 2), "launches": gx[2],
                                   "avg_launch_ms": round(gx[1] / gx[2], 4), "share_of_step": round(gx[1] / (dt * 1e3), 3),
                                   "note": "gemm_x3_kernel: 6 v_mfma_f32_32x32x16_bf16 term products per 16 k of an fp32 "
                                           "product; `achieved` counts those bf16 flops"}
        # With the weight-gradient GEMMs on a side stream UNDER the next layer's BPTT (Model.overlap_wgrad: c1-c3) a
        # GEMM's event bracket also holds its wait for CUs the recurrence occupies: not the kernel's rate, so no frac.
        contaminated = bool(graph.model.overlap_wgrad)
        alone_agg = {}
        for kind, work, s_, e_ in (prof_alone or []):
            if kind in ("gemm", "gemm_bf16", "gemm_x3"):
                a = alone_agg.setdefault(kind, [0.0, 0.0, 0])
                a[0] += work
                a[1] += s_.elapsed_time(e_)
                a[2] += 1
        if g:
            tf = g[0] / (g[1] * 1e-3) / 1e12
            peak = PEAK_BF16_MFMA_TFLOPS if bf16 else PEAK_F32_MFMA_TFLOPS
            roof = {"kernel": "gemm_bf16g_kernel (256 x 256 x 64 tiles, LDS-DMA) + gemm_bf16s / gemm_bf16_kernel on ragged shapes" if bf16
                    else "gemm_f32g_kernel (v_mfma_f32_32x32x2_f32, 256 x 256 x 32 tiles, LDS-DMA) + gemm_f32_kernel (128 x 128) on partial rounds",
                    "bound": "mfma", "achieved": round(tf, 2), "peak": peak, "unit": "TFLOP/s",
                    "frac": round(tf / peak, 4), **traffic_fields(name + ("_ragged" if ragged else ""), "gemm"),
                    "launches": g[2], "avg_launch_ms": round(g[1] / g[2], 4),
                    "share_of_step": round(g[1] / (dt * 1e3), 3)}
            ga = alone_agg.get("gemm_bf16" if bf16 else "gemm")
            if contaminated and ga:
                tfa = ga[0] / (ga[1] * 1e-3) / 1e12
                roof.update(achieved=round(tfa, 2), frac=round(tfa / peak, 4), launches=ga[2] // 3,
                            avg_launch_ms=round(ga[1] / ga[2], 4), share_of_step=None, measured_without_overlap=True,
                            note="in the timed steps this workload's weight-gradient GEMMs run on a side stream under the "
                                 "next layer's BPTT (their brackets there include waiting for CUs); this rate is from 3 "
                                 "extra steps with that overlap off, outside the timed region: the same launches, alone")
            elif contaminated:
                roof.update(achieved=None, frac=None, avg_launch_ms=None, share_of_step=None,
                            note="GEMMs overlap the next layer's BPTT on a side stream in this workload: their event "
                                 "brackets include waiting for CUs, so no per-kernel rate is quoted")
            if not full:
                roof.pop("kernel")
            gxa = alone_agg.get("gemm_x3")
            if gx and contaminated and gxa:
                tfx = gxa[0] / (gxa[1] * 1e-3) / 1e12
                line["roofline_x3"].update(achieved=round(6 * tfx, 1), frac=round(6 * tfx / PEAK_BF16_MFMA_TFLOPS, 4),
                                           fp32_equivalent_tflops=round(tfx, 2), launches=gxa[2] // 3,
                                           avg_launch_ms=round(gxa[1] / gxa[2], 4), share_of_step=None,
                                           measured_without_overlap=True)
            elif gx and contaminated:      # (the same for the split-operand weight gradients on the side stream)
                line["roofline_x3"].update(achieved=None, frac=None, fp32_equivalent_tflops=None, avg_launch_ms=None,
                                           share_of_step=None)
            if gx and gx[1] > g[1]:      # the split-operand kernels carry the step: they are the line's `roofline`
                line["roofline_f32_leftovers"] = roof
                line["roofline"] = dict(line.pop("roofline_x3"), **traffic_fields(name, "gemm"))
            else:
                line["roofline"] = roof
        c = agg.get("ctc")
        if c:
            gbs = c[0] / (c[1] * 1e-3) / 1e9
            line["roofline_ctc"] = {"kernel": "ctc_mm_kernel phase 1 + 2 (alpha / beta meet in the middle)", "bound": "hbm",
                                    "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                    "frac": round(gbs / PEAK_HBM_GBS, 4),
                                    **traffic_fields(name + ("_ragged" if ragged else ""), "ctc"),
                                    "avg_call_ms": round(c[1] / c[2], 4)}
            if not full:
                line["roofline_ctc"].pop("kernel")
            if full:
                try:        # the same op with all local micro-batches in one launch (B = 512): issue-bound regime
                    line["roofline_ctc"]["large_batch"] = ctc_large_batch(w, device)
                except Exception as exc:
                    line["roofline_ctc"]["large_batch"] = {"error": repr(exc)}
        bd = {}
        for kind in ("lstm_fwd", "lstm_bwd"):
            r = agg.get(kind)
            if r:
                bd[kind] = round(r[1] / steps, 3)
                line.setdefault("recurrence_tflops", {})[kind] = round(r[0] / (r[1] * 1e-3) / 1e12, 2)
        if g and not contaminated:
            bd["gemm"] = round(g[1] / steps, 3)
        if bf16 and agg.get("gemm"):         # the weight-only folds (R = proj.Kh and its gradient) stay fp32
            bd["gemm_f32_weight_folds"] = round(agg["gemm"][1] / steps, 3)
        if agg.get("cast_bf16"):             # fp32 -> bf16 shadow copies (natural / transposed) of the operands
            cb = agg["cast_bf16"]
            bd["cast_bf16"] = round(cb[1] / steps, 3)
            line["cast_bf16_gbs"] = round(cb[0] / (cb[1] * 1e-3) / 1e9, 1)
        if c:
            bd["ctc"] = round(c[1] / steps, 3)
        ar = agg.get("allreduce")
        if ar:
            # time the COMPUTE stream spent waiting on gradient collectives (whole-buffer all-reduce: its full duration;
            # per-layer buckets: what was not hidden under the weight-gradient GEMMs), rank 0's view
            bd["allreduce_exposed"] = round(ar[1] / steps, 3)
            line["allreduce"] = {"exposed_ms_per_step": round(ar[1] / steps, 3),
                                 "bytes_per_step": int(ar[0] / steps), "waits_per_step": round(ar[2] / steps, 1)}
        if alone is not None:
            line.setdefault("allreduce", {})["whole_gradient_alone"] = alone
        line["breakdown_ms_per_step"] = bd
    del graph
    torch.cuda.empty_cache()
    return line


def forward_only(name, device, steps=10, warmup=3):
    """The inference path of nnet-forward (nnet/graph.py:212-241) as batched here (SURVEY.md section 8f NEXT-3): the model's
    forward alone - is_training false, no dropout - on the workload's resident batch; frames/s of `steps` passes."""
    import torch
    from lstm_ctc_amd import ops
    from lstm_ctc_amd.nnet.model import Model
    w = WORKLOADS[name]
    model = Model(dict(w["cfg"], is_training=False), device, seed=123)
    x, seq, _, _ = synth_batch(w, 0, device)
    status = ops.lstm_status(device)
    status.zero_()
    for _ in range(warmup):
        model.forward(x, seq)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        logits = model.forward(x, seq)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if int(status.item()) != 0 or not bool(torch.isfinite(logits).all()):
        raise RuntimeError("forward pass failed (recurrence status %d)" % int(status.item()))
    frames = int(seq.sum().item()) * steps
    return {"frames_s": round(frames / dt, 1), "ms_per_batch": round(dt / steps * 1e3, 3), "batch": w["B"], "seq_len": w["T"],
            "lstm_schedule": ops.last_lstm_schedule()["kind"]}


def cli_corpus(name, device, steps=15, timeout=600):
    """End-to-end throughput of the product's real entry point next to the resident-input rate:
    a synthetic corpus in the recipes' format - raw 40-dim utterances of 3 * T frames as one-SequenceExample TFRecord files,
    `left_context = right_context = 1`, `subsample = 3` (egs/wsj/run_wsj_phn.sh's front end), so the model sees T frames
    of 120 - is trained for `steps` steps by bin/nnet-train.py as a CHILD process (loader threads, CRC checks, splice /
    subsample, page-locked staging, upload, run loop, logging all inside); the frames/sec it logs (after 3 warm steps) is
    compared with `run_workload` on the SAME model with its batch resident in HBM.  B distinct files are reused
    `steps` times (page cache, as in any epoch after the first)."""
    import shutil
    import subprocess
    import tempfile
    from lstm_ctc_amd.nnet import tfrecord as tfr

    w0 = WORKLOADS[name]
    cfg = dict(w0["cfg"], left_context=1, right_context=1, subsample=3)
    w = dict(w0, cfg=cfg, desc=w0["desc"] + " [input 120 = 40 x splice(1,1), subsample 3]")
    B, T, L, V = w["B"], w["T"], w["L"], cfg["num_targets"]
    resident = run_workload(name, 10, 3, device, None, 0, 1, profile=False, full=False, workload=w)
    tmp = tempfile.mkdtemp(prefix="lc_corpus_")
    try:
        rng = np.random.default_rng(777)
        paths = []
        for b in range(B):
            path = os.path.join(tmp, "utt%03d.tfrecords" % b)
            tfr.write_tfrecord(path, rng.normal(size=(3 * T, 40)).astype(np.float32), rng.integers(0, V - 1, size=L))
            paths.append(path)
        scp, scp1 = os.path.join(tmp, "tfrecords.scp"), os.path.join(tmp, "init.scp")
        with open(scp, "w") as f:
            for s in range(steps):
                for b, path in enumerate(paths):
                    f.write("s%02du%03d %d 40 1 %s\n" % (s, b, 3 * T, path))
        with open(scp1, "w") as f:
            for b, path in enumerate(paths[:4]):
                f.write("u%03d %d 40 1 %s\n" % (b, 3 * T, path))
        conf = os.path.join(tmp, "nnet.config")
        with open(conf, "w") as f:
            for k, v in cfg.items():
                f.write("%s = %s\n" % (k, str(v).lower() if isinstance(v, bool) else v))
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "TORCHELASTIC_RUN_ID")}
        py, bindir = sys.executable, os.path.join(ROOT, "bin")
        nnet0, nnet1 = os.path.join(tmp, "nnet.0"), os.path.join(tmp, "nnet.1")
        r = subprocess.run([py, os.path.join(bindir, "nnet-init.py"), scp1, conf, nnet0, "--objective", "ctc",
                            "--batch-size", "4"], capture_output=True, text=True, timeout=timeout, env=env)
        if r.returncode != 0:
            return {"error": "nnet-init failed: " + r.stderr[-400:]}
        t0 = time.perf_counter()
        r = subprocess.run([py, os.path.join(bindir, "nnet-train.py"), scp, conf, nnet0, nnet1, "--objective", "ctc",
                            "--optimizer", "adam", "--learn-rate", "4e-4", "--batch-size", str(B), "--shuffle", "false",
                            "--report-interval", "0"], capture_output=True, text=True, timeout=timeout, env=env)
        wall = time.perf_counter() - t0
        if r.returncode != 0:
            return {"error": "nnet-train failed: " + r.stderr[-400:]}
        thr = [l for l in r.stderr.splitlines() if l.startswith("INFO:tensorflow:throughput:")]
        if not thr:
            return {"error": "no throughput line in nnet-train's log"}
        fps = float(thr[-1].split("frames/sec =")[1].split()[0])
        return {"cli_frames_s": round(fps, 1), "resident_frames_s": resident["value"],
                "ratio": round(fps / resident["value"], 4), "resident_ms_per_step": resident["ms_per_step"],
                "cli_ms_per_step": round(B * T / fps * 1e3, 3), "steps_counted": steps - 3,
                "process_wall_s": round(wall, 1),
                "corpus": "%d files of %d x 40 raw frames (%.1f MB each), reused %d x; splice(1,1) + subsample 3 -> "
                          "T=%d x 120; bin/nnet-train.py --batch-size %d --num-parallel-calls 32 --batch-threads 8"
                          % (B, 3 * T, os.path.getsize(paths[0]) / 1e6, steps, T, B)}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main(argv=None):
    argv = sys.argv[1:] if argv is None else list(argv)
    args = parse_args(argv)
    under_torchrun = "WORLD_SIZE" in os.environ and "RANK" in os.environ
    if not under_torchrun and args.launch != "none" and (args.gpus > 1 or args.launch == "torchrun"):
        sys.exit(self_launch(args, argv))

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if os.environ.get("LC_BENCH_FAIL_RANK") == str(rank):
        # tests only: THIS rank's persistent recurrences give up at once (the library reads the limit per call), so that
        # the refusal of a run with one rank on the launch train can be exercised
        os.environ["LC_LSTM_SPIN_LIMIT"] = "0"
        os.environ.pop("LC_LSTM_PERSISTENT", None)
    if world != args.gpus:
        # the figure is labelled with the number of ranks that RAN; a mismatch with what was asked for is an error,
        # never a silently smaller job
        sys.stderr.write("bench.py: --gpus %d but %d rank(s) were launched (WORLD_SIZE); refusing to run\n"
                         % (args.gpus, world))
        sys.exit(3)
    shared_gpu = args.backend == "gloo" and os.environ.get("LC_BENCH_SHARED_GPU") == "1"      # tests only
    if not shared_gpu and torch.cuda.device_count() < max(world, local_rank + 1):
        sys.stderr.write("bench.py: %d rank(s) but %d GPU(s) visible; one process per GPU is the contract\n"
                         % (world, torch.cuda.device_count()))
        sys.exit(2)
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU path)"
    if shared_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    pg = None
    if world > 1 or "TORCHELASTIC_RUN_ID" in os.environ:     # launched by torch.distributed.run (also at N = 1)
        if os.environ.get("NCCL_DEBUG", "").upper() == "VERSION":      # keep stdout to the one JSON line
            os.environ["NCCL_DEBUG"] = "WARN"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if args.backend == "gloo":
            torch.distributed.init_process_group("gloo")
        else:
            torch.distributed.init_process_group("nccl", device_id=device)   # nccl == RCCL on ROCm
        pg = torch.distributed.group.WORLD
        assert torch.distributed.get_world_size(pg) == args.gpus

    try:
        res = run_workload(args.workload, args.steps, args.warmup, device, pg, rank, world,
                           profile=not args.no_profile, host_batch=args.host_batch)
    except UnhealthyRun as exc:           # every rank raises it (the verdict is all-gathered): no JSON line, non-zero
        if rank == 0:
            sys.stderr.write("bench.py: %s; refusing to print a value\n" % exc)
        sys.stderr.flush()
        if pg is not None:
            torch.distributed.destroy_process_group()
        sys.exit(4)
    if rank == 0:
        w = WORKLOADS[args.workload]
        line = {"metric": "acoustic frames/sec (whole node), 5xBiLSTM-1024 CTC" if args.workload == "c4"
                          else "acoustic frames/sec (whole node)",
                "value": res.pop("value"), "unit": res.pop("unit"), "n_gpus": world, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": res.pop("ms_per_step"), "higher_is_better": True,
                "scaling": "weak", "vs_baseline": None, "dtype": res.pop("dtype"),
                "data": "synthetic" + (" (host batch each step: PCIe-inclusive)" if args.host_batch else "")}
        res.pop("steps"), res.pop("warmup")
        res["config"]["launched_by"] = ("bench.py self-launch -> torch.distributed.run"
                                        if os.environ.get("LC_BENCH_SELF_LAUNCHED") else
                                        "torch.distributed.run" if pg is not None else "bare python")
        line.update(res)
    # the other BASELINE configs in front of the same clock: after the headline's timed region, same process, N = 1
    if world == 1 and args.workload == "c4" and not args.no_secondary:
        sec = {}
        for name in SECONDARY_DEFAULT:
            try:
                sec[name] = run_workload(name, 10, 5, device, pg, rank, world, profile=not args.no_profile, full=False)
            except Exception as exc:
                sec[name] = {"error": repr(exc)}
        # real recipe batches are ragged and length-sorted: the section-8d parity set as a throughput line (frames = sum T_b)
        for name in ("c4", "c2"):
            try:
                sec[name + "_ragged"] = run_workload(name, 10, 5, device, pg, rank, world, profile=not args.no_profile,
                                                     full=False, ragged=True)
            except Exception as exc:
                sec[name + "_ragged"] = {"error": repr(exc)}
        if rank == 0:
            line["secondary"] = sec
    # the product's real entry point on a TFRecord corpus, next to the resident-input rate of the same model
    if world == 1 and args.workload == "c4" and not args.no_cli_corpus:
        line["cli_corpus"] = {}
        for name in ("c4", "c2"):
            try:
                line["cli_corpus"][name] = cli_corpus(name, device)
            except Exception as exc:
                line["cli_corpus"][name] = {"error": repr(exc)}
    # the forward pass alone (what nnet-forward runs per batch of utterances), fp32 and split-operand
    if world == 1 and args.workload == "c4" and not args.no_secondary:
        line["inference"] = {}
        for name in ("c4", "c4x3", "c3", "c5"):
            try:
                line["inference"][name] = forward_only(name, device)
            except Exception as exc:
                line["inference"][name] = {"error": repr(exc)}
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            try:
                if args.workload in CPU_FULL_NEED_GB and not args.host_batch:
                    # the WHOLE benched batch once (c4: ~75 s on 16 cores), after every GPU leg; the bounded sample of
                    # the earlier rounds stays beside it as `cpu_baseline_sample`
                    sample = cpu_baseline(w, budget_s=10.0)
                    full = cpu_baseline_full(args.workload)
                    if "value" in full:
                        line["cpu_baseline"], line["cpu_baseline_sample"] = full, sample
                    else:
                        line["cpu_baseline"] = dict(sample, full_step=full["skipped"])
                    if args.workload == "c4":             # plus the reference's real recipe scale in full
                        line["cpu_baseline_c2_full"] = cpu_baseline_full("c2")
                else:
                    line["cpu_baseline"] = cpu_baseline(w)
            except Exception as exc:                      # the oracle is a reported baseline, never the product
                line["cpu_baseline"] = {"error": repr(exc)}
        print(json.dumps(finalize_line(line, args.workload)), flush=True)
    if pg is not None:
        torch.distributed.barrier()            # rank 0 is still measuring / printing: tear the group down together
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
