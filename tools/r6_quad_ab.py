"""Round 6: c5 step with the library as built (the shadow-only BPTT stores whole 64-byte lines after a quad transpose) or
with LC_DEV_LIB=<tag> (tools/lstm_dev_build.sh quad0 -DLC_P_SHADOW_QUAD=0: four dword stores per lane)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lstm_ctc_amd import _lib
if os.environ.get("LC_DEV_LIB"):
    _lib.LIB_PATH = _lib.LIB_PATH + "." + os.environ["LC_DEV_LIB"]
import torch
import bench
res = bench.run_workload("c5", 10, 5, torch.device("cuda", 0), None, 0, 1, profile=True, full=False)
print("lib", os.environ.get("LC_DEV_LIB", "default"), "c5", res["ms_per_step"], res["breakdown_ms_per_step"], res["config"]["last_loss_per_label"])
