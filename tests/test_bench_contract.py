"""CPU-side checks of bench.py's contract pieces (the timed path itself needs a GPU and runs on the GPU box):
the workloads it names are the BASELINE configs, and the cpu_baseline leg (the only place outside tests/ and
smoke() that may run the oracle) produces the object the driver expects."""
import importlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    sys.path.insert(0, ROOT)
    return importlib.import_module("bench")


def test_workloads_are_the_baseline_configs():
    b = _bench()
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert "5" in base["metric"] and "1024" in base["metric"]           # the metric is quoted on c4
    c4 = b.WORKLOADS["c4"]
    assert (c4["cfg"]["num_layers"], c4["cfg"]["num_neurons"], c4["B"], c4["T"]) == (5, 1024, 64, 1000)
    assert c4["cfg"].get("compute_dtype") is None                        # fp32, like the reference
    assert b.WORKLOADS["c2"]["cfg"]["num_neurons"] == 320 and b.WORKLOADS["c2"]["B"] == 32
    assert b.WORKLOADS["c3"]["cfg"]["num_experts"] == 72
    assert b.WORKLOADS["c5"]["cfg"]["compute_dtype"] == "bf16"
    assert b.WORKLOADS["c1"]["cfg"]["nnet_type"] == "lstm" and b.WORKLOADS["c1"]["cfg"]["num_neurons"] == 256
    assert b.PEAK_F32_MFMA_TFLOPS == 157.3 and b.PEAK_HBM_GBS == 8000.0


def test_cpu_baseline_object():
    b = _bench()
    w = dict(cfg=dict(nnet_type="blstm", input_dim=8, left_context=0, right_context=0, num_layers=1, num_neurons=16,
                      num_projects=16, num_targets=6, use_peepholes=True, dropout_rate=0.9), B=4, T=10, L=3)
    out = b.cpu_baseline(w, budget_s=5.0, max_T=16)
    assert set(out) == {"value", "unit", "cores", "kind", "sample"}
    assert out["unit"] == "frames/s" and out["kind"] == "port" and out["value"] > 0 and out["cores"] >= 1
    # (the tiny model is fast, so T' normally hits max_T = 16; on a loaded host the probe step may pick a shorter one)
    import re
    m = re.search(r"B=4 T=(\d+)", out["sample"])
    assert m and 1 <= int(m.group(1)) <= 16, out["sample"]


def test_more_gpus_than_visible_is_refused():
    """`python bench.py --gpus 2` on a box that shows fewer GPUs (this container: none) must exit non-zero with a clear
    message and print NO JSON line - never an N-GPU label on a smaller job."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, timeout=300, env=env, cwd=ROOT)
    import torch
    if torch.cuda.device_count() >= 2:
        return                       # a real multi-GPU box runs it; covered by the GPU suite
    assert r.returncode != 0
    assert b"--gpus 2 asked" in r.stderr and b"refusing to run" in r.stderr
    assert b"{" not in r.stdout


def test_rank_count_must_match_gpus_flag():
    """Launched with WORLD_SIZE = 1 but --gpus 4 (what a wrapper that forgets torchrun's flags would do): refused."""
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1", "--warmup", "0"],
                       capture_output=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 3 and b"rank(s) were launched" in r.stderr and b"{" not in r.stdout


def test_summary_is_last_and_inside_the_drivers_window():
    """The driver keeps the last ~8 KB of stdout (the headline CTC figure, c5 and c4x3 fell off the
    front of a 14 KB line).  `finalize_line` - what main() prints - ends the line with a compact `summary` that repeats them:
    on a real default line (round 4's, committed) every key is present, none is None, the object stays under 1 KB, sits wholly
    inside the last 8000 characters, and no `note` string survives anywhere."""
    b = _bench()
    raw = open(os.path.join(ROOT, "profiles", "r4_bench_default.json")).read().splitlines()
    line = json.loads([l for l in raw if l.startswith("{")][-1])
    out = b.finalize_line(line)
    assert list(out)[-1] == "summary" and tuple(out["summary"]) == b.SUMMARY_KEYS
    # (round 4's run had no ragged legs: those keys - added in round 6 - stay None, every other figure is there)
    assert {k for k, v in out["summary"].items() if v is None} == {"c4_ragged_frames_s", "c4_ragged_padded_share",
                                                                   "c2_ragged_frames_s"}, out["summary"]
    assert out["summary"]["c4_ms"] == line["ms_per_step"]
    assert out["summary"]["c5_ms"] == line["secondary"]["c5"]["ms_per_step"]
    assert out["summary"]["ctc_frac_b512"] == line["roofline_ctc"]["large_batch"]["frac"]
    text = json.dumps(out)
    sm = json.dumps(out["summary"])
    assert len(sm) < 1024 and text.endswith('"summary": ' + sm + "}")
    tail = text[-8000:]
    assert '"summary": ' + sm in tail
    assert '"note"' not in text
    # a leg that did not run leaves None, never a KeyError
    bare = b.finalize_line({"ms_per_step": 1.0, "value": 2.0})
    assert bare["summary"]["c4_ms"] == 1.0 and bare["summary"]["c5_ms"] is None


def test_the_committed_round5_line_ends_with_a_complete_summary():
    """The default line this round's evidence run produced on the GPU box (`profiles/r5_bench_default.json`, printed by main()
    through `finalize_line`): `summary` is its last key, every figure is there, and it sits inside the last 8000 characters
    of the line - what the driver keeps."""
    b = _bench()
    raw = open(os.path.join(ROOT, "profiles", "r5_bench_default.json")).read().splitlines()
    text = [l for l in raw if l.startswith("{")][-1]
    line = json.loads(text)
    assert list(line)[-1] == "summary" and {"c4_ms", "c5_ms", "gemm_frac", "ctc_frac_b512", "cpu_frames_s"} <= set(line["summary"])
    assert all(v is not None for v in line["summary"].values()), line["summary"]
    assert '"summary": ' + json.dumps(line["summary"]) in text[-8000:]
    assert line["summary"]["c4_ms"] == line["ms_per_step"] and line["dtype"] == "f32" and line["n_gpus"] == 1
    assert '"note"' not in text


def test_the_whole_default_line_fits_the_drivers_window():
    """Beyond the summary: `finalize_line` compacts the `secondary` entries against the headline (config keys the headline
    already states, constant roofline fields, the CTC roofline of workloads that launch the headline's CTC), so the WHOLE
    default line - 12.8 KB in round 5's first evidence run - is under 8000 characters and the driver's tail shows `metric` and
    `value` again.  Nothing measured is lost: every secondary keeps its value, step time, rooflines' achieved / frac,
    recurrence rates and breakdown; the compaction is idempotent (a line that went through it once is unchanged)."""
    b = _bench()
    raw = open(os.path.join(ROOT, "profiles", "r5_bench_default.json")).read().splitlines()
    line = json.loads([l for l in raw if l.startswith("{")][-1])
    out = b.finalize_line(line)
    text = json.dumps(out)
    assert len(text) < 8000, len(text)
    assert text.startswith('{"metric": ') and list(out)[-1] == "summary"
    assert b.finalize_line(out) == out
    assert list(out).index("secondary_protocol") + 1 == list(out).index("secondary")
    for name, e in out["secondary"].items():
        src = line["secondary"][name]
        assert e["value"] == src["value"] and e["ms_per_step"] == src["ms_per_step"] and e["dtype"] == src["dtype"]
        assert e["config"]["workload"].startswith(name + ":") and e["config"]["persist_fallbacks"] == 0
        assert e["config"]["lstm_schedule"] == src["config"]["lstm_schedule"]
        assert e["roofline"]["frac"] == src["roofline"]["frac"] and e["roofline"].get("achieved") == src["roofline"]["achieved"]
        assert e["roofline_ctc"]["frac"] == src["roofline_ctc"]["frac"]
        assert e["recurrence_tflops"] == src["recurrence_tflops"] and e["breakdown_ms_per_step"] == src["breakdown_ms_per_step"]
        assert set(e["config"]["product_kernels"]) == set(src["config"]["product_kernels"])
    assert out["secondary"]["c5"]["roofline_ctc"]["same_launches_as"] == "c4"          # B, T, L, V of the headline
    assert "same_launches_as" not in out["secondary"]["c2"]["roofline_ctc"]            # another shape: kept in full
    assert out["secondary"]["c2x3"]["roofline_ctc"]["same_launches_as"] == "c2"
    # the headline keeps the contract's fields untouched
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in out["roofline"], k
    assert out["summary"] == b.build_summary(out, "c4")


def test_the_committed_round6_line():
    """`profiles/r6_bench_default.json` (this round's default run on the GPU box, as main() printed it): under the driver's
    8 KB window, `summary` last with every key of SUMMARY_KEYS present and measured, the CPU baseline a FULL c4 step, the
    ragged lines there with their padded share, the traffic figures labelled as read from a profiled run."""
    b = _bench()
    raw = open(os.path.join(ROOT, "profiles", "r6_bench_default.json")).read().splitlines()
    text = [l for l in raw if l.startswith("{")][-1]
    line = json.loads(text)
    assert len(text) < 8000 and list(line)[-1] == "summary" and tuple(line["summary"]) == b.SUMMARY_KEYS
    assert all(v is not None for v in line["summary"].values()), line["summary"]
    assert line["metric"].startswith("acoustic frames/sec") and line["dtype"] == "f32" and line["n_gpus"] == 1
    assert line["config"]["workload"].startswith("c4:") and line["vs_baseline"] is None and line["scaling"] == "weak"
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in line["roofline"], k
    assert "not measured in this run" in line["roofline"]["traffic_source"]
    cb = line["cpu_baseline"]
    assert "FULL" in cb["sample"] and "B=64 T=1000 L=100" in cb["sample"] and cb["kind"] == "port" and cb["cores"] >= 1
    assert "cpu_baseline_sample" in line and line["summary"]["cpu_sample"] == "full step"
    rg = line["secondary"]["c4_ragged"]
    assert 0.15 < rg["ragged"]["padded_frame_share"] < 0.25 and rg["value"] < line["value"]
    assert abs(rg["ragged"]["padded_frames_s"] / line["value"] - 1.0) < 0.03      # a ragged step costs what the padded one costs
    assert b.finalize_line(line) == line                                        # what main() printed is already compact
