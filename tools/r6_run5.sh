mkdir -p gpurun_out/r6
timeout 900 python -m pytest tests/test_gpu_round6.py -q 2>&1 | tail -15 > gpurun_out/r6/pytest_round6.txt
cat gpurun_out/r6/pytest_round6.txt
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_configs.py -q -k "bf16 or epilogue" 2>&1 | tail -5
for m in 0 1 0 1; do LC_C5_SHADOW_ONLY=$m timeout 600 python bench.py --workload c5 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('shadow_only=$m', d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'], d['breakdown_ms_per_step'], d['config']['last_loss_per_label'])
"; done > gpurun_out/r6/c5_shadow_only_ab2.txt 2>&1
cat gpurun_out/r6/c5_shadow_only_ab2.txt
