"""Which bf16 shadow casts a c5 train step still makes (development): shape, orientation and call site of every
ops.cast_bf16 call of one step."""
import os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from lstm_ctc_amd import ops
from lstm_ctc_amd.nnet.model import Model

w = bench.WORKLOADS[os.environ.get("WL", "c5")]
model = Model(dict(w["cfg"]), "cuda", seed=1)
b = bench.synth_batch(w, 0, "cuda")
x, seq = b[0], b[1]
orig = ops.cast_bf16
log = []
def spy(t, nat=True, tr=False, out_nat=None):
    fr = traceback.extract_stack(limit=4)
    log.append((tuple(t.shape), "tr" if tr else "nat", "%s:%d" % (os.path.basename(fr[-3].filename), fr[-3].lineno)))
    return orig(t, nat=nat, tr=tr, out_nat=out_nat)
ops.cast_bf16 = spy
logits = model.forward(x, seq)
model.backward(torch.randn_like(logits) * 1e-3)
torch.cuda.synchronize()
for e in log:
    print(e)
print(len(log), "casts;", sum(s[0] * s[1] for s, _, _ in log) * 6 / 1e6, "MB moved")
