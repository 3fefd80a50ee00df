"""The XCD-pair recurrence at its four widths (N = 640 / 768 / 896 / 1024, one BiLSTM layer, B = 64, T = 1000): us per
step of the forward / backward recurrence (bench.py's per-launch HIP-event brackets) against the launch train
(LC_LSTM_PERSISTENT=0), and the train-step time."""
import os, sys, json, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1:
    import torch
    import bench
    N = int(sys.argv[1])
    w = dict(desc="pair probe N=%d" % N, B=64, T=1000, L=100,
             cfg=dict(nnet_type="blstm", input_dim=40, left_context=0, right_context=0, num_layers=1, num_neurons=N,
                      num_projects=N, num_targets=44, use_peepholes=True, dropout_rate=0.9))
    r = bench.run_workload("probe", 6, 2, torch.device("cuda:0"), None, 0, 1, profile=True, full=False, workload=w)
    print(json.dumps({"N": N, "ms_per_step": r["ms_per_step"], "sched": r["config"]["lstm_schedule"],
                      "bd": r.get("breakdown_ms_per_step")}))
    sys.exit(0)
for N in (640, 768, 896, 1024):
    row = {}
    for persist in ("1", "0"):
        out = subprocess.run([sys.executable, __file__, str(N)], env=dict(os.environ, LC_LSTM_PERSISTENT=persist),
                             capture_output=True, text=True).stdout.strip().splitlines()
        row[persist] = json.loads(out[-1])
    p, t = row["1"], row["0"]
    print("N=%4d  %-26s fwd %5.2f us/step  bwd %5.2f us/step  step %6.2f ms   | %-18s fwd %5.2f  bwd %5.2f  step %6.2f ms"
          % (N, p["sched"], p["bd"]["lstm_fwd"], p["bd"]["lstm_bwd"], p["ms_per_step"], t["sched"], t["bd"]["lstm_fwd"],
             t["bd"]["lstm_bwd"], t["ms_per_step"]), flush=True)
