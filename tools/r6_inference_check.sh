timeout 900 python -m pytest tests/test_gpu_round6.py -q -k "shadow_only" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_model.py tests/test_gpu_train.py -q -k "bf16 or c5" 2>&1 | tail -3
python - <<'PY'
import sys; sys.path.insert(0, "/root/repo")
import torch, bench
for name in ("c5", "c5", "c4"):
    print(name, bench.forward_only(name, torch.device("cuda", 0)))
PY
