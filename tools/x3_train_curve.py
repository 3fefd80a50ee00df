"""Training-level comparison of the compute modes: the same model (c3's shape by default), initialisation, data (8 fixed
synthetic batches, cycled) and dropout seeds trained for X3_STEPS steps under fp32, bf16x3 and bf16; prints the loss per
label averaged over windows of 8 steps (one pass over the batches).  The first steps of such a run are chaotic (DESIGN.md
section 3f), so what to read is where the curves go, not whether they agree step by step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from lstm_ctc_amd.nnet.graph import create_graph_for_training_ctc

name = os.environ.get("X3_WORKLOAD", "c3")
steps = int(os.environ.get("X3_STEPS", 160))
w = dict(bench.WORKLOADS[name])
if os.environ.get("X3_T"):
    w["T"] = int(os.environ["X3_T"])
batches = [bench.synth_batch(w, r, "cuda") for r in range(8)]
curves = {}
for mode in ("fp32", "bf16x3", "bf16"):
    g = create_graph_for_training_ctc(None, dict(w["cfg"], compute_dtype=mode), learn_rate=4e-4, clip_norm=5.0,
                                      optimizer="adam", device="cuda", seed=123)
    losses = []
    for s in range(steps):
        x, seq, labels, offs = batches[s % 8]
        out = g.step_device(x, seq, labels, offs, w["L"], int(labels.numel()), fetch_eval=True)
        losses.append(out["eval_loss"] / labels.numel())
    curves[mode] = [sum(losses[i:i + 8]) / 8 for i in range(0, steps, 8)]
    del g
    torch.cuda.empty_cache()
print("%s, %d steps, loss per label, mean over passes of 8 batches" % (w["desc"][:60], steps))
print("pass   " + "  ".join("%8s" % m for m in curves))
for i in range(len(curves["fp32"])):
    print("%4d   " % i + "  ".join("%8.4f" % curves[m][i] for m in curves))
