import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import test_gpu_configs as tc
from oracle import oracle as orc
orc.build()
from lstm_ctc_amd.nnet.model import Model
for case in ("c2_3x320_persistent", "c3_5x512_moe_persistent", "c4_5x1024_b64_t8", "c4_1024_b64_t40", "c4_1024_b100_t6"):
    cfg, B, T, wf, wb, env = tc.FP32_CASES[case]
    rng = np.random.default_rng(sum(map(ord, case)))
    x, seq, labels = tc._batch(rng, cfg, B, T)
    model = Model(cfg, "cuda", seed=17)
    params = tc._randomise_biases(model, rng)
    got = tc._run_model(model, cfg, x, seq, labels)
    ref, _ = tc._oracle_reference(orc, params, cfg, x, seq, labels)
    rl = ref["logits"]; scale = max(np.abs(rl).max(), 1.0)
    err = np.abs(got["logits"] - rl)
    for floor in (1e-1, 1e-2, 1e-3):
        rel = (err / np.maximum(np.abs(rl), floor * scale)).max()
        print(case, "scale %.2f max abs err %.2e | max rel err with floor %.0e*scale: %.2e" % (scale, err.max(), floor, rel))
